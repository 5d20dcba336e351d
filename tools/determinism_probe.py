#!/usr/bin/env python3
"""Development aid: renders a Bezier band several times in scheduled and image order and compares the frames bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cgraytracing_amd as cg, scenes
sc = cg.Scene(scenes.scene_c5(scenes.stone_texture()))
kw = dict(rows=256, row_offset=3000)
ref = sc.trace_grid_host(8192, 8192, 16, scenes.cam_dof(), 5, 12345, reorder=False, **kw)
for k in range(4):
    a = sc.trace_grid_host(8192, 8192, 16, scenes.cam_dof(), 5, 12345, reorder=(k % 2 == 0), **kw)
    d = a["rgb"] != ref["rgb"]
    print("run", k, "sched" if k % 2 == 0 else "image", "pixels differing:", int(d.any(axis=-1).sum()), "rays", a["nrays"], ref["nrays"],
          "max abs diff", float(np.abs(a["rgb"] - ref["rgb"]).max()))

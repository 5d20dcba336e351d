#!/usr/bin/env python3
"""Development aid: renders a configuration several times through the scheduled launch (unit queue, tile queue, light tiles on
the second stream -- all of them dynamic) and once in image order, and compares the frames bit for bit.

  python tools/determinism_probe.py [c3 c4 c5band] [--runs N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cgraytracing_amd as cg, scenes

which = [a for a in sys.argv[1:] if a in ("c3", "c4", "c5band")] or ["c5band"]
runs = int(sys.argv[sys.argv.index("--runs") + 1]) if "--runs" in sys.argv else 4
for name in which:
    if name == "c3":
        objs, W, H, spp, kw = scenes.scene_c3(True), 2048, 2048, 16, {}
    elif name == "c4":
        objs, W, H, spp, kw = scenes.scene_dragon(), 4096, 4096, 8, {}
    else:
        objs, W, H, spp, kw = scenes.scene_c5(scenes.stone_texture()), 8192, 8192, 16, dict(rows=256, row_offset=3000)
    with cg.Scene(objs) as sc:
        ref = sc.trace_grid_host(W, H, spp, scenes.cam_dof(), 5, 12345, reorder=False, **kw)
        bad = 0
        for k in range(runs):
            a = sc.trace_grid_host(W, H, spp, scenes.cam_dof(), 5, 12345, **kw)
            d = int((a["rgb"] != ref["rgb"]).any(axis=-1).sum())
            same = d == 0 and a["nrays"] == ref["nrays"] and np.array_equal(a["nhit"], ref["nhit"])
            bad += 0 if same else 1
            print(name, "scheduled run", k, "pixels differing from the image-order frame:", d, "rays", a["nrays"], ref["nrays"], flush=True)
        print(name, "OK" if bad == 0 else "MISMATCH in %d runs" % bad, flush=True)

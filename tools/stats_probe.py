#!/usr/bin/env python3
"""Development aid: node / triangle tests per ray of the mesh configurations (STATS variant).  python tools/stats_probe.py [c3 c4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cgraytracing_amd as cg, scenes
for name in ([a for a in sys.argv[1:] if a in ("c3", "c4")] or ["c3", "c4"]):
    objs, W, H = (scenes.scene_c3(True), 2048, 2048) if name == "c3" else (scenes.scene_dragon(), 4096, 4096)
    with cg.Scene(objs) as sc:
        r = sc.trace_grid_host(W, H, 4, scenes.cam_dof(), 5, 12345, stats=True, reorder=False)
        c = r["counters"]
        print(name, "rays", int(c[0]), "hitpoints", int(c[1]), "wave iterations", int(c[2]), "node tests", int(c[3]), "triangle tests", int(c[4]),
              "| per ray: nodes %.2f triangles %.2f" % (c[3] / c[0], c[4] / c[0]), flush=True)

#!/usr/bin/env python3
"""Development aid: cells and triangles tested per ray on C5's bump floor (STATS variant), and the band's time.  python tools/floor_stats.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, cgraytracing_amd as cg, scenes
objs = scenes.scene_c5(scenes.stone_texture()) if "--vase" in sys.argv else scenes.planes(scenes.stone_texture())  # the STATS variants do not exist for Bezier scenes
W = H = 8192
with cg.Scene(objs) as sc:
    for ro in (200, 1500, 2800, 3600):
        r = sc.trace_grid_host(W, H, 4, scenes.cam_dof(), 5, 12345, rows=64, row_offset=ro, stats=True, reorder=False)
        c = r["counters"]
        R = 512
        out = torch.zeros((R, W, 3), dtype=torch.float32, device="cuda")
        sc.trace_grid(W, H, 16, scenes.cam_dof(), 5, 12345, rows=R, row_offset=ro, out=out, nhit=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sc.trace_grid(W, H, 16, scenes.cam_dof(), 5, 12345, rows=R, row_offset=ro, out=out, nhit=False)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print("rows %d..: 64 rows at spp 4: cells/ray %.2f, triangle tests/ray %.2f; %d rows at spp 16: %.3f ms = %.1f Grays/s = %.0f lane-instruction slots per ray" % (ro, c[3] / c[0], c[4] / c[0], R, ms, W * R * 16 / ms / 1e6, 39.3e12 * ms * 1e-3 / (W * R * 16)), flush=True)

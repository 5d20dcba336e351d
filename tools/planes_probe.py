#!/usr/bin/env python3
"""Development aid: what a plane-only ray costs, with and without the thin lens.  python tools/planes_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, cgraytracing_amd as cg, scenes
W = H = 4096
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for name, objs in (("planes", scenes.planes()), ("planes+chessboard", scenes.planes(scenes.chessboard_texture(False)))):
    for cname, cam in (("pinhole", scenes.cam_pinhole()), ("thin lens", scenes.cam_dof())):
        with cg.Scene(objs) as sc:
            out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
            for reorder in (True, False):
                sc.trace_grid(W, H, spp, cam, 5, 12345, out=out, nhit=False, reorder=reorder)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    sc.trace_grid(W, H, spp, cam, 5, 12345, out=out, nhit=False, reorder=reorder)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 3
                rays = W * H * spp
                # 1024 SIMDs x 2.4 GHz / 4 cycles = 614 G wave-instructions/s = 39.3 T lane-instruction slots/s
                print("%-18s %-9s %-11s %7.3f ms  %6.1f Grays/s  = %5.0f lane-instruction slots per ray at full issue" % (name, cname, "default" if reorder else "image order", ms, rays / ms / 1e6, 39.3e12 * ms * 1e-3 / rays))

#!/usr/bin/env python3
"""Development aid: one configuration, a few frames (for rocprofv3 --kernel-trace / --pmc).  python tools/one_frame.py tiny|c4|c3|c5band|c5floor [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, cgraytracing_amd as cg, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rows, ro = None, 0
if which == "tiny":
    objs, W, H = scenes.planes() + [scenes.TriangleMesh.from_triangles(scenes.pyramid_tris(0.001, (0.0, -19.0, 30.0)), (0.6, 0.7, 0.9), 0.0, 0.0)], 4096, 4096
elif which == "c4":
    objs, W, H = scenes.scene_dragon(), 4096, 4096
elif which == "c3":
    objs, W, H = scenes.scene_c3(True), 2048, 2048
elif which == "c5floor":  # a band of C5 that sees the bump floor and nothing else (light tiles only)
    objs, W, H, rows, ro = scenes.scene_c5(scenes.stone_texture()), 8192, 8192, 512, 200
else:
    objs, W, H, rows, ro = scenes.scene_c5(scenes.stone_texture()), 8192, 8192, 256, 3000
sc = cg.Scene(objs)
r = rows or H
out = torch.zeros((r, W, 3), dtype=torch.float32, device="cuda"); cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
for _ in range(3):
    sc.trace_grid(W, H, spp, scenes.cam_dof(), 5, 12345, rows=r, row_offset=ro, out=out, nhit=False, counters=cnt)
torch.cuda.synchronize()
print(which, cnt.cpu().numpy()[:3])

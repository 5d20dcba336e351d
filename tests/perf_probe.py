"""Ad-hoc kernel timing probe (not a test): python tests/perf_probe.py [scene ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import cgraytracing_amd as cg
import scenes


def run(name, objs, cam, W, H, spp, depth, reps=5, stats=False, rows=None, row_offset=0):
    sc = cg.Scene(objs)
    rows = H if rows is None else rows
    out = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
    nh = torch.zeros((rows, W), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    kw = dict(rows=rows, row_offset=row_offset)
    sc.trace_grid(W, H, spp, cam, depth, 12345, out=out, nhit=nh, counters=cnt, stats=stats, **kw)
    torch.cuda.synchronize()
    cnt.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sc.trace_grid(W, H, spp, cam, depth, 12345, out=out, nhit=nh, counters=cnt, stats=stats, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    c = cnt.cpu().numpy() / reps
    rays = c[0]
    print("%-28s %4dx%-4d spp%-3d d%d  %8.3f ms  %9.1f Mrays/s  rays/px/s %.3f  util %.3f  nodes/ray %.1f tris/ray %.1f"
          % (name, W, rows, spp, depth, ms, rays / ms / 1e3, rays / (W * rows * spp), rays / max(1, 64 * c[2]),
             c[3] / max(rays, 1), c[4] / max(rays, 1)), flush=True)
    H = rows
    sc.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2"]
    if "c2" in which:
        run("c2 dof d5", scenes.scene_c2(), scenes.cam_dof(), 1920, 1080, 64, 5)
        run("c2 pinhole d5", scenes.scene_c2(), scenes.cam_pinhole(), 1920, 1080, 64, 5)
        run("c2 dof d1", scenes.scene_c2(), scenes.cam_dof(), 1920, 1080, 64, 1)
        run("c2 pinhole d1", scenes.scene_c2(), scenes.cam_pinhole(), 1920, 1080, 64, 1)
        run("c1 pinhole d1", scenes.scene_c1(), scenes.cam_pinhole(), 1920, 1080, 64, 1)
        m = scenes.wall_spheres() + [scenes.Sphere((-15.0, -20.0, 60), 10, (0.3, 0.3, 0.3), 0.0, 0.0),
                                     scenes.Sphere((10.0, -13.0, 30), 7, (1.0, 1.0, 1.0), 0.8, 0.0),
                                     scenes.Sphere((-8.0, -13.0, 25), 7, (1.0, 1.0, 1.0), 0.8, 0.0)]
        run("c2 two mirrors pinhole d5", m, scenes.cam_pinhole(), 1920, 1080, 64, 5)
        g = scenes.wall_spheres() + [scenes.Sphere((-15.0, -20.0, 60), 10, (0.3, 0.3, 0.3), 0.0, 0.0),
                                     scenes.Sphere((10.0, -13.0, 30), 7, (1.0, 1.0, 1.0), 0.0, 0.0),
                                     scenes.Sphere((-8.0, -13.0, 25), 7, (1.0, 1.0, 1.0), 0.8, 0.5)]
        run("c2 glass only pinhole d5", g, scenes.cam_pinhole(), 1920, 1080, 64, 5)
        g2 = scenes.wall_spheres() + [scenes.Sphere((-15.0, -20.0, 60), 10, (0.3, 0.3, 0.3), 0.0, 0.0),
                                      scenes.Sphere((10.0, -13.0, 30), 7, (1.0, 1.0, 1.0), 0.0, 0.0),
                                      scenes.Sphere((-8.0, -13.0, 25), 0.01, (1.0, 1.0, 1.0), 0.8, 0.5)]
        run("c2 tiny glass pinhole d5", g2, scenes.cam_pinhole(), 1920, 1080, 64, 5)
    if "c3" in which:
        run("c3 bunny glass dof", scenes.scene_c3(True), scenes.cam_dof(), 2048, 2048, 16, 5, reps=2, stats=True)
        run("c3 bunny glass dof nostat", scenes.scene_c3(True), scenes.cam_dof(), 2048, 2048, 16, 5, reps=2)
    if "c4" in which:
        run("c4 dragon dof", scenes.scene_dragon(), scenes.cam_dof(), 2048, 2048, 4, 5, reps=2, stats=True)
        run("c4 dragon dof nostat", scenes.scene_dragon(), scenes.cam_dof(), 2048, 2048, 4, 5, reps=2)
        run("c4 dragon 1024 spp16", scenes.scene_dragon(), scenes.cam_dof(), 1024, 1024, 16, 5, reps=2)
        run("c4 dragon 512 spp64", scenes.scene_dragon(), scenes.cam_dof(), 512, 512, 64, 5, reps=2)
        run("c4 dragon 2048 spp16", scenes.scene_dragon(), scenes.cam_dof(), 2048, 2048, 16, 5, reps=1)
    if "c5" in which:
        tex = scenes.stone_texture()
        run("c5 bump only", scenes.planes(tex), scenes.cam_dof(), 1024, 1024, 2, 5, reps=1, stats=True)
        run("c5 bump+vase", scenes.scene_c5(tex), scenes.cam_dof(), 1536, 1536, 2, 5, reps=1)
        run("vase only", scenes.planes() + [scenes.vase_bezier()], scenes.cam_dof(), 1536, 1536, 2, 5, reps=1)
    if "c5band" in which:  # a slice of C5 at its real width: 256 rows through the vase, stone-sized bump floor
        tex = scenes.stone_texture()
        run("c5 band rows 3000..3255", scenes.scene_c5(tex), scenes.cam_dof(), 8192, 8192, 16, 5, reps=1, rows=256, row_offset=3000)
    if "c5full" in which:  # BASELINE.json configs[4], one GPU's share at the full sample count: 8192 x 1024 rows, spp 1024
        tex = scenes.stone_texture()
        run("c5 share, spp 1024", scenes.scene_c5(tex), scenes.cam_dof(), 8192, 8192, 1024, 5, reps=1, rows=1024, row_offset=3584)

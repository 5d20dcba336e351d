"""Ad-hoc probe (not a test): CGRT_GRID_SPLIT_SAMPLES on / off for a few workloads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, cgraytracing_amd as cg, scenes

def t(name, sc, W, H, spp, cam, split, rows=None, off=0, reps=3):
    rows = H if rows is None else rows
    out = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    kw = dict(rows=rows, row_offset=off, out=out, nhit=False, counters=cnt, split_samples=split)
    sc.trace_grid(W, H, spp, cam, 5, 12345, **kw)
    torch.cuda.synchronize(); cnt.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sc.trace_grid(W, H, spp, cam, 5, 12345, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-34s split=%d  %9.3f ms  %9.1f Mrays/s" % (name, split, ms, int(cnt[0]) / reps / ms / 1e3), flush=True)

cam = scenes.cam_dof()
sc = cg.Scene(scenes.scene_c2())
for sp in (False, True):
    t("C2 1920x1080 spp64", sc, 1920, 1080, 64, cam, sp)
    t("C2 512x512 spp256", sc, 512, 512, 256, cam, sp)
sc.close()
sc = cg.Scene(scenes.scene_dragon())
for sp in (False, True):
    t("dragon 512x512 spp64", sc, 512, 512, 64, cam, sp)
    t("dragon 2048x2048 spp64", sc, 2048, 2048, 64, cam, sp, reps=1)
sc.close()
sc = cg.Scene(scenes.scene_c3(True))
for sp in (False, True):
    t("C3 2048x2048 spp64", sc, 2048, 2048, 64, cam, sp, reps=1)
    t("bunny 512x512 spp256", sc, 512, 512, 256, cam, sp, reps=1)
sc.close()

"""Ad-hoc probe (not a test): is C2 bound by the critical path of its heaviest tiles?  Times the full frame and single
tile rows (8 image rows = 60 workgroups) through the glass / mirror spheres and through plain wall."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cgraytracing_amd as cg, scenes

sc = cg.Scene(scenes.scene_c2())
cam = scenes.cam_dof()
W, H, SPP = 1920, 1080, 64

def t(rows, off, reps=5):
    out = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    sc.trace_grid(W, H, SPP, cam, 5, 12345, rows=rows, row_offset=off, out=out, nhit=False, counters=cnt)
    torch.cuda.synchronize(); cnt.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sc.trace_grid(W, H, SPP, cam, 5, 12345, rows=rows, row_offset=off, out=out, nhit=False, counters=cnt)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, int(cnt[0]) // reps

print("full frame        %.3f ms  %d rays" % t(H, 0))
for off in (0, 96, 144, 184, 232, 280, 400, 800):
    ms, rays = t(8, off)
    print("rows %4d..%4d   %.3f ms  %d rays (%.2f per sample)" % (off, off + 7, ms, rays, rays / (8 * W * SPP)))
ms, rays = t(400, 0); print("rows 0..399       %.3f ms  %d rays" % (ms, rays))
ms, rays = t(680, 400); print("rows 400..1079    %.3f ms  %d rays" % (ms, rays))

#!/usr/bin/env python3
"""Development aid (not a test): frame time of the BASELINE configurations with and without cost-aware scheduling.
  [CGRT_HEAVY_DIV=n] [CGRT_UNITS_PER_ITEM=n] python tools/sched_probe.py [c2 c3 c4 c5band c5share ...] [--spp-scale f]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import cgraytracing_amd as cg
import scenes


def run(name, objs, cam, W, H, spp, rows=None, row_offset=0, reps=3):
    sc = cg.Scene(objs)
    rows = H if rows is None else rows
    out = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    res = {}
    for mode, reorder in (("sched", True), ("image-order", False)):
        kw = dict(rows=rows, row_offset=row_offset, out=out, nhit=False, counters=cnt, reorder=reorder)
        sc.trace_grid(W, H, spp, cam, 5, 12345, **kw)
        torch.cuda.synchronize()
        cnt.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            sc.trace_grid(W, H, spp, cam, 5, 12345, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rays = int(cnt[0]) // reps
        res[mode] = (ms, rays, out.double().sum().item())
    sc.close()
    a, b = res["sched"], res["image-order"]
    print("%-10s %5dx%-5d spp%-4d sched %9.3f ms  image-order %9.3f ms  x%.2f  %8.1f Mrays/s  same_rays=%s same_sum=%s  [div=%s upi=%s]"
          % (name, W, rows, spp, a[0], b[0], b[0] / a[0], a[1] / a[0] / 1e3, a[1] == b[1], a[2] == b[2],
             os.environ.get("CGRT_HEAVY_DIV", "-"), os.environ.get("CGRT_UNITS_PER_ITEM", "-")), flush=True)


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c2", "c3", "c4", "c5band"]
    dof = scenes.cam_dof()
    if "c2" in which:
        run("c2", scenes.scene_c2(), dof, 1920, 1080, 64, reps=10)
    if "c3" in which:
        run("c3", scenes.scene_c3(True), dof, 2048, 2048, 64)
    if "c4" in which:
        run("c4", scenes.scene_dragon(), dof, 4096, 4096, 64, reps=2)
    if "c4full" in which:
        run("c4full", scenes.scene_dragon(), dof, 4096, 4096, 256, reps=1)
    if "c5band" in which:
        run("c5band", scenes.scene_c5(scenes.stone_texture()), dof, 8192, 8192, 16, rows=256, row_offset=3000, reps=2)
    if "c5share" in which:  # one GPU's share at a quarter of the samples
        run("c5share", scenes.scene_c5(scenes.stone_texture()), dof, 8192, 8192, 256, rows=1024, row_offset=3584, reps=1)

#!/usr/bin/env python3
"""Development aid: what each rank of an N-GPU run would compute, timed one rank at a time on ONE GPU.

  python tools/scaling_emulation.py [c2 c2strong c4 c5] [--out profiles/rNN_scaling_emulated.json] [--c5-spp N]

For N in {1, 2, 4, 8} and every rank r < N the rank's share of bench.py's configuration (same frame, same block-cyclic
stripes, same spp) is rendered alone on this GPU; the slowest share is the time an N-GPU run would need for its compute
phase.  emulated_efficiency = (rays all ranks trace / slowest share's time) / (N x the N = 1 rate): load balance of the
sharding only -- no gather, no second process, no clock or power interaction between GPUs.  It is NOT a scaling measurement
(the driver's SCALE run on a real 8-GPU node is); it shows whether the stripes deal the work evenly."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import bench
import cgraytracing_amd as cg
import scenes
from cgraytracing_amd.dist import local_rows


class A:  # the bits of bench.py's argument namespace resolve() reads
    def __init__(self, config, scaling=None, spp=None):
        self.config, self.scaling, self.stripe_rows, self.spp = config, scaling, None, spp


def share_time(sc, cfg, rank, reps):
    W, H, spp, S, shares = cfg["W"], cfg["H"], cfg["spp"], cfg["stripe_rows"], cfg["shares"]
    rows = local_rows(H, S, rank, shares)
    stripe = (S, rank, shares) if shares > 1 else None
    out = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    kw = dict(rows=rows, stripe=stripe, out=out, nhit=False, counters=cnt)
    sc.trace_grid(W, H, spp, scenes.cam_dof(), 5, 12345, **kw)
    torch.cuda.synchronize()
    cnt.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sc.trace_grid(W, H, spp, scenes.cam_dof(), 5, 12345, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, int(cnt[0]) // reps


def main():
    which = [a for a in sys.argv[1:] if a in ("c2", "c2strong", "c4", "c5")] or ["c2", "c2strong", "c4", "c5"]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    c5_spp = int(sys.argv[sys.argv.index("--c5-spp") + 1]) if "--c5-spp" in sys.argv else None
    doc = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "device": torch.cuda.get_device_name(0), "configs": {}}
    for name in which:
        config, scaling = ("c2", "strong") if name == "c2strong" else (name, None)
        sc = cg.Scene(bench.build_scene_objects(config))
        rows_doc, base_rate = [], None
        for n in (1, 2, 4, 8):
            cfg = bench.resolve(A(config, scaling, c5_spp if config == "c5" else None), n)
            reps = 5 if config == "c2" else 1
            ranks = range(n)
            if config == "c5" and n > 1:  # every share is rendered once (N = 8); smaller N use the first N of them
                ranks = range(n) if n == 8 else []
            times, rays = [], []
            for r in ranks:
                ms, ry = share_time(sc, cfg, r, reps)
                times.append(ms)
                rays.append(ry)
            if config == "c5" and n in (2, 4):
                continue
            if config == "c5" and n == 8:  # derive N = 2, 4 from the same eight shares
                for m in (2, 4, 8):
                    t, ry = max(times[:m]), sum(rays[:m])
                    rows_doc.append({"n": m, "frame": "%dx%d" % (cfg["W"], cfg["H"]), "spp": cfg["spp"], "slowest_share_ms": round(t, 3),
                                     "fastest_share_ms": round(min(times[:m]), 3), "rays": ry,
                                     "emulated_efficiency": round(ry / t / (m * base_rate), 4)})
                continue
            t, ry = max(times), sum(rays)
            if n == 1:
                base_rate = ry / t
            rows_doc.append({"n": n, "frame": "%dx%d" % (cfg["W"], cfg["H"]), "spp": cfg["spp"], "slowest_share_ms": round(t, 3),
                             "fastest_share_ms": round(min(times), 3), "rays": ry,
                             "emulated_efficiency": round(ry / t / (n * base_rate), 4)})
            print(name, rows_doc[-1], flush=True)
        sc.close()
        doc["configs"][name] = {"scaling": bench.resolve(A(config, scaling), 2)["scaling"], "rows": rows_doc}
    print(json.dumps(doc, indent=1))
    if out:
        json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()

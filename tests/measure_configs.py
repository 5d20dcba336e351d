"""Measurement driver (not a test): Mrays/s of every BASELINE.json configuration on one MI355X, with the CPU
reference (oracle/_ref, 1 thread) or the oracle port timed on a small crop of the same workload beside it.
Writes one JSON document to stdout (and to FILE with --out).   python tests/measure_configs.py [--quick] [--out FILE]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import cgraytracing_amd as cg
import scenes
from backends import Backend, BackendScene, have_ref, to_acc32


def gpu_run(objs, cam, W, H, spp, depth, rows=None, row_offset=0, reps=3):
    sc = cg.Scene(objs)
    rows = H if rows is None else rows
    out = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
    kw = dict(rows=rows, row_offset=row_offset, out=out, nhit=False, counters=cnt)
    sc.trace_grid(W, H, spp, cam, depth, 12345, **kw)
    torch.cuda.synchronize()
    cnt.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sc.trace_grid(W, H, spp, cam, depth, 12345, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    rays = int(cnt[0]) // reps
    st = sc.stats()
    sc.close()
    return dict(ms=round(ms, 3), rays=rays, mrays_per_s=round(rays / ms / 1e3, 1), scene_bytes=st["scene_bytes_fp64"],
                out=out)


def cpu_run(objs, cam, W, H, spp, depth, row0, nrows, gpu_rows=None):
    kind = "reference" if have_ref() else "port"
    be = Backend("ref" if kind == "reference" else "orc")
    be.set_threads(1)
    sc = BackendScene(be, objs)
    r = sc.trace_grid(cam, W, H, spp, depth, 12345, row0=row0, nrows=nrows, hashsize=1000001 if kind == "reference" else 1)
    sc.close()
    res = dict(kind=kind, cores=1, sample="rows %d..%d of %dx%d at spp %d" % (row0, row0 + nrows - 1, W, H, spp),
               rays=r["nrays"], seconds=round(r["seconds"], 3), mrays_per_s=round(r["nrays"] / r["seconds"] / 1e6, 3))
    if gpu_rows is not None:
        res["linf_vs_gpu"] = float(np.abs(gpu_rows - to_acc32(r["acc_sum"], spp)).max())
    return res


def main():
    quick = "--quick" in sys.argv
    dof, pin = scenes.cam_dof(), scenes.cam_pinhole()
    tex = scenes.stone_texture()
    cfgs = [
        # name, objs, camera, W, H, spp (measured), spp (config), depth, rows rendered, crop rows for CPU (start, n, spp)
        ("C1 256x256 spp1 spheres depth1", scenes.scene_c1(), pin, 256, 256, 1, 1, 1, None, (0, 256, 1)),
        ("C2 1920x1080 spp64 spheres+glass", scenes.scene_c2(), dof, 1920, 1080, 64, 64, 5, None, (200, 16, 64)),
        ("C3 2048x2048 spp64 bunny+chessboard", scenes.scene_c3(True), dof, 2048, 2048, 64, 64, 5, None, (560, 4, 64)),
        ("C4 4096x4096 spp256 dragon (1 GPU renders all rows)", scenes.scene_dragon(), dof, 4096, 4096,
         16 if quick else 256, 256, 5, None, (1000, 2, 4)),
        ("C5 8192x8192 spp1024 bump floor + Bezier (1 GPU share = 1024 rows)", scenes.scene_c5(tex), dof, 8192, 8192,
         4 if quick else 32, 1024, 5, 1024, (3600, 1, 1)),
    ]
    doc = {"device": torch.cuda.get_device_name(0), "configs": []}
    for name, objs, cam, W, H, spp, spp_cfg, depth, rows, (c0, cn, cspp) in cfgs:
        row_offset = 0 if rows is None else (H - rows) // 2
        g = gpu_run(objs, cam, W, H, spp, depth, rows=rows, row_offset=row_offset, reps=1 if spp * W > 2e5 else 3)
        # CPU on a crop at the GPU's spp when that is what the crop uses, else a separate small GPU render for parity
        sc = cg.Scene(objs)
        crop = sc.trace_grid_host(W, H, cspp, cam, depth, 12345, rows=cn, row_offset=c0)["rgb"]
        sc.close()
        c = cpu_run(objs, cam, W, H, cspp, depth, c0, cn, crop)
        rows_eff = H if rows is None else rows
        entry = dict(config=name, gpu_ms=g["ms"], gpu_rays=g["rays"], gpu_mrays_per_s=g["mrays_per_s"],
                     measured_spp=spp, config_spp=spp_cfg,
                     gpu_ms_at_config_spp=round(g["ms"] * spp_cfg / spp, 1),
                     algorithmic_bytes=12 * W * rows_eff + g["scene_bytes"],
                     achieved_gb_per_s=round((12 * W * rows_eff + g["scene_bytes"]) / (g["ms"] * spp_cfg / spp * 1e-3) / 1e9, 3),
                     cpu=c, gpu_over_cpu=round(g["mrays_per_s"] / c["mrays_per_s"], 0))
        doc["configs"].append(entry)
        print(json.dumps(entry), file=sys.stderr, flush=True)
    # the compiled reference prints progress text to stdout from C; "--out FILE" keeps the document clean
    if "--out" in sys.argv:
        with open(sys.argv[sys.argv.index("--out") + 1], "w") as fh:
            json.dump(doc, fh, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Development aid (not a test): when and where every workgroup of one trace_grid launch ran.

  python tools/timeline_probe.py c3|c4|c2|c5band [--spp N] [--split] [--natural] [--out gpurun_out/tl.json]

Sets CGRT_TIMELINE_FILE so that libcgrt.so records, per workgroup, {start, end} on the 100 MHz wall clock, the hardware
id (XCC, SE, CU) and the rays it traced, then prints: launch span, concurrency over time (resident workgroups in 20 time
bins), the distribution of workgroup durations, the share of the span during which fewer than half of the slots were in
use ("tail"), and per-XCD finish times."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else None
    split = "--split" in sys.argv
    natural = "--natural" in sys.argv  # CGRT_GRID_NO_REORDER
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    import cgraytracing_amd as cg
    import scenes
    cam = scenes.cam_dof()
    rows, row_offset = None, 0
    if which == "c3":
        objs, W, H, s0 = scenes.scene_c3(True), 2048, 2048, 64
    elif which == "c4":
        objs, W, H, s0 = scenes.scene_dragon(), 4096, 4096, 64
    elif which == "c2":
        objs, W, H, s0 = scenes.scene_c2(), 1920, 1080, 64
    else:
        objs, W, H, s0 = scenes.scene_c5(scenes.stone_texture()), 8192, 8192, 16
        rows, row_offset = 256, 3000
    spp = spp or s0
    sc = cg.Scene(objs)
    sc.trace_grid_host(W, H, 1, cam, 5, 12345, rows=rows, row_offset=row_offset)  # warm-up
    tf = tempfile.mktemp(suffix=".tl")
    os.environ["CGRT_TIMELINE_FILE"] = tf
    r = sc.trace_grid_host(W, H, spp, cam, 5, 12345, rows=rows, row_offset=row_offset, split_samples=split, reorder=not natural)
    del os.environ["CGRT_TIMELINE_FILE"]
    sc.close()
    raw = np.fromfile(tf, dtype=np.uint64)
    os.unlink(tf)
    nblk, nthr, chunks, xcd_tiles = [int(x) for x in raw[:4]]
    tl = raw[4:].reshape(nblk, 4)
    ran = tl[:, 1] > 0
    t0 = tl[ran, 0].astype(np.int64)
    t1 = tl[ran, 1].astype(np.int64)
    base = t0.min()
    a, b = (t0 - base) / 100.0, (t1 - base) / 100.0  # microseconds
    span = b.max()
    dur = b - a
    xcc = ((tl[ran, 2] >> np.uint64(32)) & np.uint64(15)).astype(int)
    hw = (tl[ran, 2] & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hw >> 8) & 15
    se = (hw >> 13) & 3
    rays = (tl[ran, 3] >> np.uint64(32)).astype(np.int64)
    bins = 20
    edges = np.linspace(0, span, bins + 1)
    conc = [float(np.minimum(b, edges[i + 1]).clip(min=0).__sub__(np.maximum(a, edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]))
            for i in range(bins)]
    order = np.argsort(dur)[::-1]
    doc = {
        "workload": "%s %dx%d spp %d%s%s" % (which, W, rows or H, spp, " split-samples" if split else "", " image order" if natural else " cost order"),
        "workgroups": int(ran.sum()), "threads": nthr, "chunks": chunks, "xcd_tiles": xcd_tiles, "rays": int(r["nrays"]),
        "span_us": round(span, 1),
        "resident_workgroups_by_time_bin": [round(c, 1) for c in conc],
        "workgroup_us": {"mean": round(float(dur.mean()), 1), "p50": round(float(np.percentile(dur, 50)), 1),
                         "p90": round(float(np.percentile(dur, 90)), 1), "p99": round(float(np.percentile(dur, 99)), 1),
                         "max": round(float(dur.max()), 1)},
        "sum_workgroup_us": round(float(dur.sum()), 1),
        "mean_resident_workgroups": round(float(dur.sum() / span), 1),
        "heaviest": [{"us": round(float(dur[i]), 1), "start_us": round(float(a[i]), 1), "rays": int(rays[i]), "xcc": int(xcc[i]),
                      "se": int(se[i]), "cu": int(cu[i])} for i in order[:8]],
        "xcc_finish_us": {str(x): round(float(b[xcc == x].max()), 1) for x in sorted(set(xcc.tolist()))},
        "xcc_busy_us": {str(x): round(float(dur[xcc == x].sum()), 1) for x in sorted(set(xcc.tolist()))},
        "dur_weighted_by_rays_corr": round(float(np.corrcoef(dur, rays)[0, 1]), 3),
    }
    idx = np.nonzero(ran)[0]
    last = np.argsort(b)[::-1][:12]
    doc["last_to_finish"] = [{"block": int(idx[i]), "start_us": round(float(a[i]), 1), "end_us": round(float(b[i]), 1), "rays": int(rays[i])}
                             for i in last]
    late = b > 0.8 * span
    doc["finishing_in_last_fifth"] = {"workgroups": int(late.sum()), "mean_us": round(float(dur[late].mean()), 1) if late.any() else 0,
                                      "started_after_half": int((late & (a > 0.5 * span)).sum())}
    half = 0.5 * max(conc)
    doc["fraction_of_span_below_half_peak_concurrency"] = round(sum(1 for c in conc if c < half) / bins, 2)
    print(json.dumps(doc, indent=1))
    if out:
        json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()

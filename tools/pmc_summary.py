#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (CSV) of one kernel into a small JSON document for profiles/.

  python tools/pmc_summary.py --kernel trace_grid_kernel --out profiles/rNN_x_pmc.json \
         [--command "..."] [--workload "..."] [--min-us 500] DIR_OR_CSV [DIR_OR_CSV ...]

Every pass is its own rocprofv3 run of the same command (counters that do not fit together are collected separately, as
MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes: FETCH_SIZE and WRITE_SIZE never in one pass).  Per counter the mean
over the dispatches of the selected kernel (name substring; dispatches shorter than --min-us are warm-ups and are dropped)
is kept, then the usual ratios are derived:

  hbm_bytes_per_launch   (2*FETCH_SIZE + WRITE_SIZE) * 1024   (both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte
                         requests at 64 bytes, so the read side is doubled -- MI355X_MICROARCH.md, HBM section)
  valu_busy              4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)     [quad-cycles -> cycles]
  lanes_per_valu_inst    SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  waves_per_simd         SQ_WAVE_CYCLES*4 / (GRBM_GUI_ACTIVE/8 * 1024)                   (average resident waves per SIMD)
  wave_wait_fraction     SQ_WAIT_ANY / SQ_WAVE_CYCLES
  l1_hit_rate            1 - TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum
  l2_hit_rate            TCC_HIT_sum / TCC_REQ_sum
"""
import argparse
import csv
import glob
import json
import os
import sys


def read_pass(path, kernel, min_us):
    files = [path] if os.path.isfile(path) else sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True))
    per = {}
    meta = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if kernel not in row["Kernel_Name"]:
                    continue
                dur_us = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                if dur_us < min_us:
                    continue
                per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
                meta = {"kernel_name": row["Kernel_Name"], "vgpr": int(row["VGPR_Count"]), "sgpr": int(row["SGPR_Count"]),
                        "scratch_bytes": int(row["Scratch_Size"]), "lds_bytes": int(row["LDS_Block_Size"]),
                        "workgroup": int(row["Workgroup_Size"]), "grid": int(row["Grid_Size"])}
    out = {}
    for name, d in per.items():
        vals = list(d.values())
        out[name] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
    return out, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--command", default="")
    ap.add_argument("--workload", default="")
    ap.add_argument("--min-us", type=float, default=500.0)
    ap.add_argument("--alg-bytes", type=float, default=None)
    ap.add_argument("paths", nargs="+")
    a = ap.parse_args()
    counters, meta = {}, {}
    for p in a.paths:
        c, m = read_pass(p, a.kernel, a.min_us)
        counters.update(c)
        meta = m or meta
    if not counters:
        sys.exit("no dispatch of %r found" % a.kernel)
    g = lambda k: counters[k]["mean"] if k in counters else None
    d = {}
    if g("GRBM_GUI_ACTIVE"):
        simd_cycles = g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0
        if g("SQ_ACTIVE_INST_VALU"):
            d["valu_busy"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / simd_cycles
        if g("SQ_WAVE_CYCLES"):
            d["waves_per_simd"] = 4.0 * g("SQ_WAVE_CYCLES") / simd_cycles
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        d["lanes_per_valu_inst"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
    if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
        d["wave_wait_fraction"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_WAIT_INST_ANY") and g("SQ_WAVE_CYCLES"):
        d["wave_issue_stall_fraction"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
    if g("TCP_TCC_READ_REQ_sum") and g("TCP_TOTAL_CACHE_ACCESSES_sum"):
        d["l1_hit_rate"] = 1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")
    if g("TCC_HIT_sum") and g("TCC_REQ_sum"):
        d["l2_hit_rate"] = g("TCC_HIT_sum") / g("TCC_REQ_sum")
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("SQ_INSTS_VMEM"):
        d["l1_line_accesses_per_vmem_instruction"] = g("TCP_TOTAL_CACHE_ACCESSES_sum") / g("SQ_INSTS_VMEM")
    doc = {"command": a.command, "workload": a.workload, "kernel": meta.get("kernel_name", a.kernel),
           "dispatch": {k: v for k, v in meta.items() if k != "kernel_name"}, "counters": counters, "derived": d}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        doc["hbm_bytes_per_launch"] = int((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024)
        doc["hbm_read_bytes_per_launch"] = int(2 * g("FETCH_SIZE") * 1024)
        doc["hbm_write_bytes_per_launch"] = int(g("WRITE_SIZE") * 1024)
        if a.alg_bytes:
            doc["algorithmic_bytes_per_launch"] = int(a.alg_bytes)
            doc["traffic_over_algorithmic"] = doc["hbm_bytes_per_launch"] / a.alg_bytes
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps({"kernel": doc["kernel"][:80], "derived": d, "hbm_bytes_per_launch": doc.get("hbm_bytes_per_launch")}, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel resource table (VGPRs, spilled VGPRs, scratch bytes per lane, occupancy) of libcgrt.so's device code.

  python tools/resource_table.py [--out profiles/rNN_resource_usage.json]

Runs `make -C cgraytracing_amd/csrc asm` (hipcc -S --cuda-device-only -Rpass-analysis=kernel-resource-usage, the same
flags as the product build; cross-compiles for gfx950 without a GPU) and parses the compiler's remarks.  The template
arguments of trace_grid_kernel are decoded from the mangled name: <TREES,BEZ,DOF,GLASS,SPH,STATS,HPS,NT>."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle_variant(name):
    m = re.match(r"_Z17trace_grid_kernelIL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)ELi(\d+)EL?b(\d)EE", name)
    if m:
        t = [int(x) for x in m.groups()]
        return "trace_grid_kernel<TREES=%d,BEZ=%d,DOF=%d,GLASS=%d,SPH=%d,STATS=%d,HPS=%d,NT=%d,HEAVY=%d>" % tuple(t)
    m = re.match(r"_Z17trace_grid_kernelIL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)ELi(\d+)EL?b(\d)EL?b(\d)EE", name)
    if m:
        t = [int(x) for x in m.groups()]
        return "trace_grid_kernel<TREES=%d,BEZ=%d,DOF=%d,GLASS=%d,SPH=%d,STATS=%d,HPS=%d,NT=%d,SPILL=%d,HFONLY=%d>" % tuple(t)
    m = re.match(r"_Z23trace_grid_sched_kernelIL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)ELi(\d+)EE", name)
    if m:
        t = [int(x) for x in m.groups()]
        return "trace_grid_sched_kernel<TREES=%d,BEZ=%d,DOF=%d,GLASS=%d,SPH=%d,STATS=%d,NT=%d>" % tuple(t)
    m = re.match(r"_Z17trace_grid_kernelIL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)EL?b(\d)ELi(\d+)EE", name)
    if m:
        t = [int(x) for x in m.groups()]
        return "trace_grid_kernel<TREES=%d,BEZ=%d,DOF=%d,GLASS=%d,SPH=%d,STATS=%d,HPS=%d,NT=%d>" % tuple(t)
    m = re.search(r"19photon_trace_kernelILb(\d)ELb(\d)EE", name)
    if m:
        return "photon_trace_kernel<BEZ=%s,SPILL=%s>" % m.groups()
    m = re.search(r"19primary_walk_kernelILb(\d)EE", name)
    if m:
        return "primary_walk_kernel<DOF=%s>" % m.group(1)
    m = re.match(r"_Z(\d+)", name)
    if m:
        k = int(m.group(1))
        start = m.end()
        return name[start:start + k]
    return name


def main():
    out = None
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "cgraytracing_amd", "csrc"), "asm"], capture_output=True, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stdout + p.stderr)
        raise SystemExit(p.returncode)
    text = p.stdout + p.stderr
    rows, cur = [], None
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
            "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
            "LDS Size [bytes/block]": "static_lds_bytes"}
    for ln in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = {"kernel": demangle_variant(m.group(1)), "mangled": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]+\])?): (\S+) \[-Rpass", ln)
        if m and cur is not None and m.group(1).strip() in keys:
            v = m.group(2)
            cur[keys[m.group(1).strip()]] = int(v) if v.isdigit() else v
    rows = [r for r in rows if "vgprs" in r and "rocprim" not in r["mangled"] and "hipcub" not in r["mangled"]]  # own kernels only
    doc = {"command": "make -C cgraytracing_amd/csrc asm  (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -S --cuda-device-only "
                      "-Rpass-analysis=kernel-resource-usage)",
           "note": "static compiler report; dynamic LDS (pending-ray levels, object list, node cache) is set at launch and not shown",
           "kernels": rows}
    print("%-78s %5s %6s %7s %4s" % ("kernel", "VGPR", "spills", "scratch", "occ"))
    for r in rows:
        print("%-78s %5d %6d %7d %4d" % (r["kernel"][:78], r["vgprs"], r["vgpr_spills"], r["scratch_bytes_per_lane"],
                                         r["occupancy_waves_per_simd"]))
    if out:
        with open(out, "w") as fh:
            json.dump(doc, fh, indent=1)


if __name__ == "__main__":
    main()

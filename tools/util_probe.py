#!/usr/bin/env python3
"""Development aid: where a heavy wave's lanes idle.  Needs the probe build:
  make -C cgraytracing_amd/csrc exp NAME=util DEFS=-DCGRT_UTIL
  CGRT_DEV_LIBS=1 CGRT_LIB=cgraytracing_amd/libcgrt_exp_util.so python tools/util_probe.py [c3|c4] [spp]
Prints, per probe point, the number of wave-level executions and the mean number of lanes active in them."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cgraytracing_amd as cg, scenes
from cgraytracing_amd import _capi
NAMES = {0: "heavy loop iteration (lanes with a ray)", 1: "tree_hit call (lanes on)", 2: "node step", 3: "leaf phase", 4: "triangle-box pretest",
         5: "exact triangle test", 7: "diffuse shading (heavy)", 8: "glass shading (heavy)", 10: "pending-ray pop (heavy)",
         11: "tile loop iteration", 12: "diffuse shading (tile)", 13: "glass shading (tile)", 14: "pending-ray pop (tile)", 16: "walk kernel: 4-wide node step", 17: "walk kernel: leaf phase", 18: "walk kernel: exact triangle test",
         19: "walk kernel: finish phase (lanes pending)", 20: "walk kernel: refill (lanes taking a unit)"}
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
objs, W, H = (scenes.scene_c3(True), 2048, 2048) if name == "c3" else (scenes.scene_dragon(), 4096, 4096)
lib = _capi.lib()
buf = (ctypes.c_ulonglong * 64)()
with cg.Scene(objs) as sc:
    out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    sc.trace_grid(W, H, spp, scenes.cam_dof(), 5, 12345, out=out, nhit=False)
    torch.cuda.synchronize()
    lib.cgrt_util_dump(buf)   # the first call includes the cost probe; count a second, steady one
    sc.trace_grid(W, H, spp, scenes.cam_dof(), 5, 12345, out=out, nhit=False)
    lib.cgrt_util_dump(buf)
for k in range(32):
    n, l = buf[2 * k], buf[2 * k + 1]
    if n:
        print("%2d %-42s executions %12d  lanes/execution %6.2f" % (k, NAMES.get(k, "?"), n, l / n))

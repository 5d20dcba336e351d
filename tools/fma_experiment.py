#!/usr/bin/env python3
"""EXPERIMENT: the eye pass compiled with FMA contraction allowed (make -C cgraytracing_amd/csrc fma ->
libcgrt_fma.so) against the product build and the oracle on the full C2 frame.

  python tools/fma_experiment.py [--out profiles/r02_fma_experiment.json]

Each build is loaded in its own child process (CGRT_LIB); this process holds the oracle.  Reported: frame time, ray count,
L-infinity and number of mismatching pixels of each build against the oracle's fp32 accumulator over ALL 1920x1080 pixels.
The north star's bar is L-inf < 1e-4; the product's own bar is bit equality."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

CHILD = r'''
import os, sys, json
ROOT = sys.argv[1]; out = sys.argv[2]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cgraytracing_amd as cg, scenes
sc = cg.Scene(scenes.scene_c2())
W, H, spp = 1920, 1080, 64
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
sc.trace_grid(W, H, spp, scenes.cam_dof(), 5, 12345, out=buf, nhit=False, counters=cnt); torch.cuda.synchronize(); cnt.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): sc.trace_grid(W, H, spp, scenes.cam_dof(), 5, 12345, out=buf, nhit=False, counters=cnt)
e1.record(); torch.cuda.synchronize()
np.save(out, buf.cpu().numpy())
print(json.dumps({"ms": e0.elapsed_time(e1) / 10, "rays": int(cnt[0]) // 10}))
'''

def run(lib):
    out = tempfile.mktemp(suffix=".npy")
    env = dict(os.environ)
    if lib:
        env["CGRT_LIB"] = lib
        env["CGRT_DEV_LIBS"] = "1"
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT, out], env=env, capture_output=True, text=True)
    if p.returncode: raise SystemExit(p.stderr)
    info = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    img = np.load(out); os.unlink(out)
    return info, img

def main():
    import scenes
    from backends import Backend, BackendScene, to_acc32
    be = Backend("orc"); be.set_threads(os.cpu_count() or 1)
    want = BackendScene(be, scenes.scene_c2()).trace_grid(scenes.cam_dof(), 1920, 1080, 64, 5, 12345)
    ref32 = to_acc32(want["acc_sum"], 64)
    doc = {"workload": "C2 1920x1080 spp 64, all pixels, vs the CPU oracle (rays %d)" % want["nrays"], "builds": {}}
    for name, lib in (("product (-ffp-contract=off)", None), ("experiment (-ffp-contract=fast)", os.path.join(ROOT, "cgraytracing_amd", "libcgrt_fma.so"))):
        if lib and not os.path.exists(lib):
            doc["builds"][name] = "not built (make -C cgraytracing_amd/csrc fma)"; continue
        info, img = run(lib)
        d = np.abs(img.astype(np.float64) - ref32.astype(np.float64)).max(axis=-1)
        doc["builds"][name] = {"ms_per_frame": round(info["ms"], 4), "rays": info["rays"], "rays_equal_oracle": info["rays"] == want["nrays"],
                               "linf": float(d.max()), "pixels_not_bit_equal": int((d > 0).sum()),
                               "pixels_over_1e-4": int((d > 1e-4).sum()), "pixels_over_1e-6": int((d > 1e-6).sum())}
    print(json.dumps(doc, indent=1))
    if "--out" in sys.argv: json.dump(doc, open(sys.argv[sys.argv.index("--out") + 1], "w"), indent=1)

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Algorithmic fp64 operation counts of the eye pass, counted EXACTLY by the instrumented build of the CPU oracle
(oracle/liborc_flops.so: every +, -, *, /, sqrt and pow/sin/cos/atan/exp call the restated algorithm performs;
oracle/cgrt_flopcount.h) -- SURVEY.md section 8d, last row; VERDICT r2 item 4.  CPU only (TEST INFRASTRUCTURE side).

  python tools/flop_count.py [c1 c2 c3 c4 c5 main] [--procs N] [--out profiles/r03_flops.json]

c1 ... c4 are counted over the WHOLE frame at the configuration's full sample count (every pixel, every sample).  c5 is counted
over a stated subset of the same workload (every pixel of the bench share at samples 0..1 of its 1024) -- exact for that
subset; flops per ray of the subset x the frame's ray count is the frame figure bench.py derives.  Rows are dealt to N single-threaded processes (the counters are per thread)."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {
    # name: (scene, camera, W, H, spp counted, spp of the configuration, depth, [(row0, nrows)...] or None = all rows, what)
    "c1": ("c1", "pinhole", 256, 256, 1, 1, 1, None, "whole frame"),
    "c2": ("c2", "dof", 1920, 1080, 64, 64, 5, None, "whole frame, all 64 samples"),
    "c3": ("c3", "dof", 2048, 2048, 64, 64, 5, None, "whole frame, all 64 samples"),
    "c4": ("c4", "dof", 4096, 4096, 256, 256, 5, None, "whole frame, all 256 samples"),
    "c5": ("c5", "dof", 8192, 8192, 2, 1024, 5, [(k * 8 * 16, 16) for k in range(64)],
           "every pixel of share 0 of 8 (the 64 16-row stripes = 0 mod 8: the bench share), samples 0..1 of 1024"),
    "main": ("main", "pinhole", 1024, 768, 1, 1, 5, None, "the reference's committed main() eye pass: whole frame"),
}


def build(scene):
    import scenes
    if scene == "c1":
        return scenes.scene_c1()
    if scene == "c2":
        return scenes.scene_c2()
    if scene == "c3":
        return scenes.scene_c3(True)
    if scene == "c4":
        return scenes.scene_dragon()
    if scene == "c5":
        return scenes.scene_c5(scenes.stone_texture())
    return scenes.planes(scenes.stone_texture()) + [scenes.TriangleMesh.from_triangles(scenes.dragon_tris(), (0.25, 0.25, 0.5), 0.0, 0.0, 1)]


def work(args):
    name, rows = args
    import scenes
    from backends import Backend, BackendScene
    scene, camn, W, H, spp, _, depth, _, _ = CONFIGS[name]
    be = Backend("flops")
    with open(os.devnull, "w") as dn:  # the reference-style progress prints of scene construction
        sc = BackendScene(be, build(scene))
    cam = scenes.cam_dof() if camn == "dof" else scenes.cam_pinhole()
    tot = {}
    rays = hps = 0
    for r0, nr in rows:
        be.flop_reset()
        r = sc.trace_grid(cam, W, H, spp, depth, seed=12345, row0=r0, nrows=nr)
        c = be.flop_counts()
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
        rays += r["nrays"]
        hps += int(r["nhit"].sum())
    sc.close()
    return tot, rays, hps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["c1", "c2", "c3", "c4", "c5", "main"])
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    out = {"what": "operation counts of the eye pass by the instrumented CPU oracle (tools/flop_count.py; add_sub, mul, div, sqrt each count "
                   "as one flop; transcendental calls listed separately); command: python tools/flop_count.py " + " ".join(a.configs),
           "configs": {}}
    if a.out and os.path.exists(a.out):
        out["configs"] = json.load(open(a.out)).get("configs", {})
    for name in a.configs:
        scene, camn, W, H, spp, spp_cfg, depth, rows, what = CONFIGS[name]
        rows = rows or [(0, H)]
        # deal the rows in chunks, heaviest parts interleaved
        chunks = []
        for r0, nr in rows:
            step = max(1, nr // (4 * a.procs)) if nr > 64 else nr
            chunks += [(r, min(step, r0 + nr - r)) for r in range(r0, r0 + nr, step)]
        jobs = [[] for _ in range(min(a.procs, len(chunks)))]
        for i, c in enumerate(chunks):
            jobs[i % len(jobs)].append(c)
        t0 = time.time()
        with mp.get_context("spawn").Pool(len(jobs)) as pool:
            res = pool.map(work, [(name, j) for j in jobs])
        tot, rays, hps = {}, 0, 0
        for t, r, h in res:
            rays += r
            hps += h
            for k, v in t.items():
                tot[k] = tot.get(k, 0) + v
        rec = {"scene": scene, "width": W, "height": H, "spp_counted": spp, "spp_of_configuration": spp_cfg, "max_depth": depth,
               "subset": what, "rays": rays, "hitpoints": hps, "counts": tot, "flops_per_ray": tot["flops"] / rays,
               "transcendental_calls_per_ray": tot["transcendental_calls"] / rays, "seconds": round(time.time() - t0, 1)}
        out["configs"][name] = rec
        print(name, json.dumps(rec), flush=True)
        if a.out:
            json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

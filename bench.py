#!/usr/bin/env python3
"""bench.py -- Mrays/s of the eye pass on BASELINE.json's headline configuration.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full frame of configs[1]: 1920x1080, spp = 64 thin-lens samples, recursion depth 5, the
8-sphere wall/diffuse/mirror/glass scene of SURVEY.md §8d (C2).  With N > 1 the grid is sharded by rows (weak
scaling as SURVEY.md §8e defines it: every GPU renders 1080 rows; the frame is 1920 x 1080*N), stripes are
block-cyclic, and each step ends with the RCCL gather of the framebuffer to rank 0 and its un-permute.
Inputs (the committed scene) are resident in HBM before the timed region; outputs stay in HBM.

The JSON line carries, besides the contract's fields:
  roofline     : the trace_grid kernel against the HBM roof.  achieved = algorithmic bytes per launch
                 (12 B per pixel framebuffer store + S_scene read once) / average kernel time
                 measured with HIP events on the launch stream.  The path is FP64-VALU bound, not HBM bound:
                 frac << 1 is the expected, stated result.
  cpu_baseline : the compiled REFERENCE (oracle/_ref, unmodified main.cpp trace()) when its prebuilt library
                 travelled with the repo, else our CPU oracle ("port"); 1 thread like the reference's serial
                 HitPointPass, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, SPP, DEPTH, SEED = 1920, 1080, 64, 5, 12345
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(sample_spp):
    """Times the reference's own eye pass (or the oracle port) on host cores; returns the JSON object."""
    import scenes
    from backends import Backend, BackendScene, have_ref

    kind = "reference" if have_ref() else "port"
    be = Backend("ref" if kind == "reference" else "orc")
    be.set_threads(1)
    sc = BackendScene(be, scenes.scene_c2())
    r = sc.trace_grid(scenes.cam_dof(), W, H, spp=sample_spp, depth=DEPTH, seed=SEED,
                      hashsize=1000001 if kind == "reference" else 1)
    sc.close()
    return {
        "value": round(r["nrays"] / r["seconds"] / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": kind,
        "sample": "%dx%d spp=%d (samples 0..%d of the same scene/camera/seed), %d rays in %.2f s, 1 thread "
                  "(the reference's HitPointPass is serial, main.cpp:185-219)" %
                  (W, H, sample_spp, sample_spp - 1, r["nrays"], r["seconds"]),
    }, r


def cpu_baseline_mt(sample_spp):
    import scenes
    from backends import Backend, BackendScene

    be = Backend("orc")
    n = os.cpu_count() or 1
    be.set_threads(n)
    sc = BackendScene(be, scenes.scene_c2())
    r = sc.trace_grid(scenes.cam_dof(), W, H, spp=sample_spp, depth=DEPTH, seed=SEED)
    sc.close()
    return {"value": round(r["nrays"] / r["seconds"] / 1e6, 4), "unit": "Mrays/s", "cores": n, "kind": "port",
            "sample": "%dx%d spp=%d, OpenMP over rows" % (W, H, sample_spp)}


def other_configs(dev_index):
    """Secondary, single-frame measurements on this GPU (N = 1 only; a few seconds in total): BASELINE.json configs[2]
    and configs[3] at full size on one GPU, and the reference's committed main() configuration end to end (eye pass,
    20.48 M photons, gather, tone map).  Reported beside the headline value, never part of it."""
    import torch

    import cgraytracing_amd as cg
    import scenes

    out = {}

    def eye(name, objs, cam, w, h, spp):
        sc = cg.Scene(objs, device=dev_index)
        buf = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:%d" % dev_index)
        cnt = torch.zeros(8, dtype=torch.int64, device=buf.device)
        sc.trace_grid(w, h, 1, cam, DEPTH, SEED, out=buf, nhit=False, counters=cnt)  # warm-up: one sample per pixel
        torch.cuda.synchronize()
        cnt.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sc.trace_grid(w, h, spp, cam, DEPTH, SEED, out=buf, nhit=False, counters=cnt)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        rays = int(cnt[0].item())
        sc.close()
        out[name] = {"ms_per_frame": round(ms, 2), "rays": rays, "mrays_per_s": round(rays / ms / 1e3, 1)}

    eye("C3 2048x2048 spp64 glass bunny + ChessBoard floor, thin lens", scenes.scene_c3(True), scenes.cam_dof(), 2048, 2048, 64)
    eye("C4 4096x4096 spp256 dragon (all rows on one GPU), thin lens", scenes.scene_dragon(), scenes.cam_dof(), 4096, 4096, 256)
    tex = scenes.stone_texture()
    objs = scenes.planes(tex) + [scenes.TriangleMesh.from_triangles(scenes.dragon_tris(), (0.25, 0.25, 0.5), 0.0, 0.0, 1)]
    sc = cg.Scene(objs, device=dev_index)
    sc.ppm_render(64, 48, 1, scenes.cam_pinhole(), 5, SEED, nphotons=1000)
    r = sc.ppm_render(1024, 768, 1, scenes.cam_pinhole(), 5, SEED, nphotons=20480000)
    sc.close()
    out["reference main() configuration: 1024x768 spp1, bump floor + dragon, 20 480 000 photons (rows f1/f2)"] = {
        "ms_total": round(sum(r["ms"].values()), 1), "stage_ms": {k: round(v, 2) for k, v in r["ms"].items()},
        "photons_per_s": round(20480000 / (r["ms"]["photons"] / 1e3)), "photon_events": r["n_events"]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-spp", type=int, default=24, help="spp of the CPU baseline sample (0 = skip); 24 = about 10 s on one core")
    ap.add_argument("--stripe-rows", type=int, default=8)
    ap.add_argument("--check", action="store_true", help="verify a crop of the frame against the oracle")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the secondary single-frame measurements")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import cgraytracing_amd as cg
    import scenes
    from cgraytracing_amd.dist import StripedRenderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    n = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # one rank per GPU; CGRT_BENCH_BACKEND=gloo lets several ranks share a GPU to rehearse the N > 1 control path
    backend = os.environ.get("CGRT_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if n > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    Ht = H * n  # weak scaling: 1080 rows per GPU
    scene = cg.Scene(scenes.scene_c2(), device=dev_index)
    cam = scenes.cam_dof()
    stats = scene.stats()
    counters = torch.zeros(8, dtype=torch.int64, device=dev)

    sr = StripedRenderer(W, Ht, stripe_rows=args.stripe_rows)
    rows_local = sr.rows_local
    # Two output buffers: with N > 1 the gather of frame k (side stream) overlaps the render of frame k+1.
    outs = [torch.zeros((rows_local, W, 3), dtype=torch.float32, device=dev) for _ in range(2 if n > 1 else 1)]
    kernel_events = []
    comm = torch.cuda.Stream(device=dev) if n > 1 else None
    done_evt = [torch.cuda.Event() for _ in outs]
    free_evt = [None for _ in outs]
    state = {"k": 0}

    def step(record=False):
        k = state["k"] % len(outs)
        state["k"] += 1
        out = outs[k]
        cur = torch.cuda.current_stream(dev)
        if free_evt[k] is not None:
            cur.wait_event(free_evt[k])  # the gather that last read this buffer has finished
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        scene.trace_grid(W, Ht, SPP, cam, DEPTH, SEED, rows=rows_local, stripe=sr.stripe, out=out, nhit=False,
                         counters=counters)
        if record:
            e1.record()
            kernel_events.append((e0, e1))
        if n == 1:
            return out[:Ht]
        done_evt[k].record(cur)
        with torch.cuda.stream(comm):
            comm.wait_event(done_evt[k])
            frame = sr.gather(out)  # RCCL gather to rank 0 + un-permute, on the side stream
            ev = torch.cuda.Event()
            ev.record(comm)
            free_evt[k] = ev
        return frame

    def fence():
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    counters.zero_()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame = step(record=True)
    fence()
    dt = time.perf_counter() - t0

    red_dev = dev if backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    cnt = counters.clone().to(red_dev)
    if n > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    total_rays = int(cnt[0].item())
    kern_ms = sum(a.elapsed_time(b) for a, b in kernel_events) / max(1, len(kernel_events))

    if rank == 0:
        rays_per_step = total_rays / args.steps
        value = total_rays / dt / 1e6
        # algorithmic HBM bytes of one launch on one GPU: fp32 RGB store + scene read once
        alg_bytes = 12 * W * rows_local + stats["scene_bytes_fp64"]
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        # PMC figures come from the committed rocprofv3 passes of this same command (profiles/r01_pmc.json):
        # HBM bytes per launch, and -- because the binding resource is fp64 VALU issue, not HBM -- how busy the VALUs were
        traffic, valu = None, None
        ppath = os.path.join(ROOT, "profiles", "r01_pmc.json")
        if n == 1 and os.path.exists(ppath):
            try:
                prof = json.load(open(ppath))
                traffic = prof.get("hbm_bytes_per_launch")
                cn = prof["counters"]
                simd_cycles = 1024 * cn["GRBM_GUI_ACTIVE"]["mean"] / 8.0
                valu = {"busy_frac": round(4.0 * cn["SQ_ACTIVE_INST_VALU"]["mean"] / simd_cycles, 3),
                        "lanes_active_frac": round(cn["SQ_THREAD_CYCLES_VALU"]["mean"] /
                                                   (64.0 * cn["SQ_ACTIVE_INST_VALU"]["mean"]), 3),
                        "wave_insts_per_launch": int(cn["SQ_INSTS_VALU"]["mean"]), "source": "profiles/r01_pmc.json"}
            except Exception:
                traffic, valu = None, None
        line = {
            "metric": "Mrays/sec (primary+secondary) at 1920x1080 spp=64",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: 1920x1080 spp=64 spheres+mirror+glass, depth 5, thin lens (BASELINE.json configs[1])",
                       "width": W, "height": Ht, "rows_per_gpu": rows_local, "spp": SPP, "max_depth": DEPTH,
                       "rays_per_step": int(rays_per_step), "sharding": "rows, block-cyclic %d-row stripes, "
                       "gather to rank 0" % args.stripe_rows if n > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                         "kernel": "trace_grid_kernel<TREES=0,BEZ=0,DOF=1,GLASS=1,SPH=1,STATS=0>", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes": int(alg_bytes),
                         "note": "FP64-VALU/divergence bound by design (SURVEY.md §8d H5); HBM fraction reported as required",
                         "kernel_mrays_per_s": round(rays_per_step / n / (kern_ms * 1e-3) / 1e6, 2),
                         "lane_utilisation": round(total_rays / max(1, 64 * int(cnt[2].item())), 4),
                         "valu": valu},
        }
        if args.cpu_spp > 0:
            base, _ = cpu_baseline(args.cpu_spp)
            line["cpu_baseline"] = base
            line["gpu_over_cpu"] = round(value / base["value"], 1)
            try:
                line["cpu_baseline_all_cores"] = cpu_baseline_mt(args.cpu_spp)
            except Exception as e:  # pragma: no cover
                line["cpu_baseline_all_cores"] = {"error": str(e)}
        if n == 1 and not args.no_other_configs:
            try:
                line["other_configs"] = other_configs(dev_index)
            except Exception as e:  # pragma: no cover - never let a secondary measurement lose the headline line
                line["other_configs"] = {"error": repr(e)}
        if args.check and n == 1:
            from backends import Backend, BackendScene, to_acc32
            import numpy as np
            be = Backend("orc")
            be.set_threads(os.cpu_count() or 1)
            osc = BackendScene(be, scenes.scene_c2())
            r0 = 200
            want = osc.trace_grid(cam, W, H, SPP, DEPTH, SEED, row0=r0, nrows=16)
            got = frame[r0:r0 + 16].cpu().numpy()
            line["check_linf_rows_200_215"] = float(np.abs(got - to_acc32(want["acc_sum"], SPP)).max())
        print(json.dumps(line), flush=True)
    if n > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

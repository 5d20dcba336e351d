#!/usr/bin/env python3
"""bench.py -- Mrays/s of the eye pass on BASELINE.json's configurations.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4|c5] [--scaling weak|strong]

N > 1 needs one process per GPU.  Either an external launcher starts them (the driver's
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: WORLD_SIZE/RANK/LOCAL_RANK are then set), or
plain `python bench.py --gpus N` starts them ITSELF: the parent -- which never imports torch, never touches a GPU and never
execs -- spawns N fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, waits, and
exits with their worst code; rank 0 prints the one JSON line.  CGRT_BENCH_BACKEND=gloo lets several ranks share one GPU
(rehearsal of the N > 1 control path on a one-GPU box).

A "step" is one frame of the chosen configuration.  Inputs (the committed scene) are resident in HBM before the timed
region; outputs stay in HBM.  With N > 1 rows are dealt in block-cyclic stripes and each step ends with the gather of the
framebuffer to rank 0 (torch `nccl` = RCCL over xGMI) and its un-permute; the gather of frame k runs on a side stream under
the render of frame k+1.

  c2 (default) BASELINE.json configs[1], the configuration the metric is quoted on: 1920x1080, spp 64 thin-lens samples,
     depth 5, the 8-sphere wall / diffuse / mirror / glass scene.  N = 1 is exactly that frame.
       --scaling weak (default for N > 1): same camera, same view, N x the pixels -- the frame grows to
         (1920 sqrt N) x (1080 sqrt N), rounded to the 32 x 8 tile, so per-GPU work is fixed (within 0.3 %) and the
         ray mix (rays per pixel-sample, share of glass / mirror pixels) is the one of the N = 1 frame.  [Round 1 grew the
         frame to 1920 x 1080 N with the same camera: the aspect ratio, hence the ray mix, changed with N.]
       --scaling strong: the 1920x1080 frame itself is split over the N GPUs.
  c4 configs[3]: 4096x4096, spp 256, dragon (100 000 triangles), STRONG scaling: 16-row block-cyclic stripes over N GPUs,
     gather to rank 0.  N = 1 renders all rows.
  c5 configs[4]: 8192x8192, spp 1024, Bezier vase + bump-mapped stone floor, WEAK scaling: the frame is dealt in 16-row
     stripes to 8 shares of 8192 x 1024 rows; N GPUs render shares 0..N-1 (N = 8: the whole frame; N = 1: one share --
     a uniform 1/8 sample of the frame's rows, so per-GPU work and ray mix do not depend on N).

The JSON line carries, besides the contract's fields:
  roofline     : the trace_grid kernel against the HBM roof.  achieved = algorithmic bytes per launch (12 B per pixel
                 framebuffer store + S_scene read once) / average kernel time measured in THIS run with HIP
                 events on the launch stream.  The path is FP64-VALU bound, not HBM bound: frac << 1 is the expected,
                 stated result.  `traffic` is null in the live line: PMC counters cannot be read from
                 inside the process; `traffic_from_profile` / `valu_from_profile` REPLAY the committed rocprofv3 --pmc
                 passes of this same command (profiles/) and say so.
  cpu_baseline : the compiled REFERENCE (oracle/_ref, unmodified main.cpp trace(); `kind` "reference") when its prebuilt
                 library travelled with the repo -- it is git-ignored and only exists where __graft_entry__.build() ran
                 with /root/reference present -- else our CPU oracle ("port"); `ref_library_found` records which.
                 1 thread like the reference's serial HitPointPass, on a bounded sample of the same workload.
  --rehearse   : no GPU: the kernel launch is replaced by a closed-form row pattern so that launcher, rendezvous, stripe
                 mapping, gather and the JSON line can be exercised on CPU with gloo (tests/test_bench_launch.py).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DEPTH, SEED = 5, 12345
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
C5_SHARES = 8           # configs[4] is defined on 8 GPUs: 8 shares of 8192 x 1024 rows


# ---------------------------------------------------------------------------------------------------------------------
# configurations
# ---------------------------------------------------------------------------------------------------------------------
def weak_dims(n):
    """c2 weak scaling: the same view at N x the pixels, rounded to whole 32 x 8 tiles."""
    k = math.sqrt(n)
    return 32 * int(round(1920 * k / 32)), 8 * int(round(1080 * k / 8))


def resolve(args, n):
    """Frame, sharding and defaults of (config, scaling, N).  No scene objects are built here (the rehearsal needs none)."""
    c = args.config
    if c == "c2":
        scaling = args.scaling or "weak"
        W, H = (1920, 1080) if (n == 1 or scaling == "strong") else weak_dims(n)
        cfg = dict(name="c2", W=W, H=H, spp=64, stripe_rows=args.stripe_rows or 8, shares=n, scaling=scaling,
                   steps=20, warmup=3,
                   metric="Mrays/sec (primary+secondary) at 1920x1080 spp=64",
                   workload="C2: 1920x1080 spp=64 spheres+mirror+glass, depth 5, thin lens (BASELINE.json configs[1])")
        if n > 1:
            cfg["workload"] += ("; strong scaling: the same frame split over %d GPUs" % n if scaling == "strong" else
                                "; weak scaling: same view at %d x the pixels (%dx%d)" % (n, W, H))
            if scaling != "strong":  # the frame that was rendered, not the N = 1 frame's size
                cfg["metric"] = "Mrays/sec (primary+secondary) at %dx%d spp=64 (C2's view at %d x the pixels of 1920x1080)" % (W, H, n)
    elif c == "c4":
        cfg = dict(name="c4", W=4096, H=4096, spp=256, stripe_rows=args.stripe_rows or 16, shares=n, scaling="strong",
                   steps=5, warmup=1,
                   metric="Mrays/sec (primary+secondary) at 4096x4096 spp=256",
                   workload="C4: 4096x4096 spp=256 dragon.txt mesh (100 000 triangles) + 5 planes, depth 5, thin lens, "
                            "row-tiled in 16-row block-cyclic stripes (BASELINE.json configs[3])")
    elif c == "c5":
        if n > C5_SHARES:
            raise SystemExit("c5 is defined on at most %d GPUs" % C5_SHARES)
        cfg = dict(name="c5", W=8192, H=8192, spp=1024, stripe_rows=args.stripe_rows or 16, shares=C5_SHARES, scaling="weak",
                   steps=2, warmup=1,
                   metric="Mrays/sec (primary+secondary) at 8192x8192 spp=1024",
                   workload="C5: 8192x8192 spp=1024 Bezier vase + bump-mapped stone.jpg floor + 4 planes, depth 5, thin lens; "
                            "weak scale: %d of 8 shares of 8192x1024 rows (16-row block-cyclic stripes) "
                            "(BASELINE.json configs[4])" % n)
    else:
        raise SystemExit("unknown --config %r" % c)
    if args.spp:
        cfg["spp"] = args.spp
        cfg["workload"] += " [spp overridden to %d: NOT the named configuration]" % args.spp
    return cfg


def build_scene_objects(name):
    import scenes
    if name == "c2":
        return scenes.scene_c2()
    if name == "c4":
        return scenes.scene_dragon()
    return scenes.scene_c5(scenes.stone_texture())


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines (reported beside the GPU number, never part of it)
# ---------------------------------------------------------------------------------------------------------------------
class stdout_to_stderr:
    """The compiled reference prints progress text from C (objects.h:482,503, bezier.h:66); stdout must carry exactly one
    JSON line, so file descriptor 1 points at stderr while CPU baselines and secondary measurements run."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def cpu_sample_plan(cfg, sample_spp):
    """(spp, [(row0, nrows), ...], description): a bounded, uniform sample of the workload -- about 10-30 s on one host core."""
    if cfg["name"] == "c2":
        return sample_spp or 24, [(0, cfg["H"])], "all rows"
    S = 16
    every = 32 if cfg["name"] == "c4" else 64
    rows = [(s * S, S) for s in range(0, cfg["H"] // S, every)]
    return sample_spp or (32 if cfg["name"] == "c4" else 1), rows, "every %dth 16-row stripe (%d stripes)" % (every, len(rows))


def cpu_baseline(cfg, sample_spp, threads=1):
    """Times the reference's own eye pass (threads == 1; the oracle port if the reference library did not travel) or the
    oracle port on all cores (threads > 1) on the sample; returns the JSON object."""
    import scenes
    from backends import REF_SO, Backend, BackendScene, have_ref

    found = have_ref()
    kind = "reference" if (found and threads == 1) else "port"
    be = Backend("ref" if kind == "reference" else "orc")
    be.set_threads(threads)
    # the full-frame configurations are sampled on the N = 1 frame of the configuration (the reference is one process)
    W, H = (1920, 1080) if cfg["name"] == "c2" else (cfg["W"], cfg["H"])
    spp, rows, what = cpu_sample_plan(dict(cfg, H=H), sample_spp)
    sc = BackendScene(be, build_scene_objects(cfg["name"]))
    nrays, secs = 0, 0.0
    for r0, nr in rows:
        r = sc.trace_grid(scenes.cam_dof(), W, H, spp=spp, depth=DEPTH, seed=SEED, row0=r0, nrows=nr,
                          hashsize=1000001 if kind == "reference" else 1)
        nrays += r["nrays"]
        secs += r["seconds"]
    sc.close()
    out = {"value": round(nrays / secs / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": kind,
           "sample": "%dx%d, %s, spp=%d (samples 0..%d of the same scene/camera/seed): %d rays in %.2f s, %s" %
                     (W, H, what, spp, spp - 1, nrays, secs,
                      "1 thread (the reference's HitPointPass is serial, main.cpp:185-219)" if threads == 1
                      else "OpenMP over rows")}
    if threads == 1:
        out["ref_library_found"] = bool(found)
        out["ref_library"] = os.path.relpath(REF_SO, ROOT) + (
            " (built by __graft_entry__.build() where /root/reference exists; git-ignored, travels with the working tree)"
            if found else " NOT FOUND: timed the CPU oracle port instead")
    return out


def other_configs(dev_index):
    """Secondary, single-frame measurements on this GPU (c2, N = 1 only; a few seconds in total): BASELINE.json configs[2]
    and configs[3] at full size on one GPU, and the reference's committed main() configuration end to end (eye pass,
    20.48 M photons, gather, tone map).  Reported beside the headline value, never part of it."""
    import torch

    import cgraytracing_amd as cg
    import scenes

    out = {}

    flops = load_flops()

    def eye(name, objs, cam, w, h, spp, rows=None, stripe=None, flop_key=None):
        sc = cg.Scene(objs, device=dev_index)
        rows = h if rows is None else rows
        buf = torch.zeros((rows, w, 3), dtype=torch.float32, device="cuda:%d" % dev_index)
        cnt = torch.zeros(8, dtype=torch.int64, device=buf.device)
        kw = dict(rows=rows, stripe=stripe, out=buf, nhit=False, counters=cnt)
        # warm-up: the same launch once (a scheduled launch sizes the scene handle's scratch by its sample count on first use)
        sc.trace_grid(w, h, spp, cam, DEPTH, SEED, **kw)
        torch.cuda.synchronize()
        cnt.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sc.trace_grid(w, h, spp, cam, DEPTH, SEED, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        rays = int(cnt[0].item())
        alg = 12 * w * rows + sc.stats()["scene_bytes_fp64"]
        rec = {"ms_per_frame": round(ms, 2), "rays": rays, "mrays_per_s": round(rays / ms / 1e3, 1),
               "kernel": sc.kernel_variant(w, h, spp, cam, DEPTH, rows=rows, stripe=stripe),
               "kernel_ms": round(ms, 3), "kernel_ms_note": "HIP events around the whole launch sequence of one frame (probe, plan, "
               "scheduled kernel + light-tile kernel, ordered sum)",
               "algorithmic_bytes": int(alg), "hbm_achieved_gbps": round(alg / (ms * 1e-3) / 1e9, 3),
               "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6)}
        rec.update(flop_fields(flops, flop_key, rays, ms))
        sc.close()
        out[name] = rec

    eye("C3 2048x2048 spp64 glass bunny + ChessBoard floor, thin lens", scenes.scene_c3(True), scenes.cam_dof(), 2048, 2048, 64, flop_key="c3")
    eye("C4 4096x4096 spp256 dragon (all rows on one GPU), thin lens", scenes.scene_dragon(), scenes.cam_dof(), 4096, 4096, 256, flop_key="c4")
    tex = scenes.stone_texture()
    # configs[4] at its per-GPU size: share 0 of 8 (16-row block-cyclic stripes) of the 8192 x 8192 frame, all 1024 samples --
    # the frame `bench.py --config c5` times at N = 1
    eye("C5 8192x8192 spp1024 Bezier vase + stone.jpg bump floor: one GPU's share (8192x1024 rows, stripes = 0 mod 8), thin lens",
        scenes.scene_c5(tex), scenes.cam_dof(), 8192, 8192, 1024, rows=1024, stripe=(16, 0, C5_SHARES), flop_key="c5")
    objs = scenes.planes(tex) + [scenes.TriangleMesh.from_triangles(scenes.dragon_tris(), (0.25, 0.25, 0.5), 0.0, 0.0, 1)]
    sc = cg.Scene(objs, device=dev_index)
    sc.ppm_render(64, 48, 1, scenes.cam_pinhole(), 5, SEED, nphotons=1000)
    r_first = sc.ppm_render(1024, 768, 1, scenes.cam_pinhole(), 5, SEED, nphotons=20480000)
    r = sc.ppm_render(1024, 768, 1, scenes.cam_pinhole(), 5, SEED, nphotons=20480000)  # same call again: buffers of that size warm
    sc.close()
    out["reference main() configuration: 1024x768 spp1, stone.jpg bump floor + dragon, 20 480 000 photons (rows f1/f2)"] = {
        "ms_total": round(sum(r["ms"].values()), 1), "stage_ms": {k: round(v, 2) for k, v in r["ms"].items()},
        "photons_per_s": round(20480000 / (r["ms"]["photons"] / 1e3)), "photon_events": r["n_events"],
        "first_call_ms_total": round(sum(r_first["ms"].values()), 1)}
    return out


FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X_MICROARCH.md: fp64 vector FMA peak (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)
VALU_ISSUE_PEAK_GINST = 614.4     # wave-instructions/s when every VALU instruction is a 4-cycle fp64 one: 1024 SIMDs x 2.4 GHz / 4


def load_flops():
    """profiles/r03_flops.json: operation counts of the eye pass by the instrumented CPU oracle (tools/flop_count.py)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r03_flops.json")))["configs"]
    except Exception:
        return {}


def flop_fields(flops, key, rays, ms):
    """Algorithmic fp64 flops of `rays` rays of configuration `key` and the rate they were done at in `ms` milliseconds."""
    f = flops.get(key) if key else None
    if not f:
        return {}
    total = f["flops_per_ray"] * rays
    exact = f["spp_counted"] == f["spp_of_configuration"] and "whole frame" in f["subset"] and f["rays"] == rays
    tf = total / (ms * 1e-3) / 1e12
    return {"algorithmic_flops": int(f["counts"]["flops"]) if exact else int(total),
            "algorithmic_flops_source": "profiles/r03_flops.json: instrumented CPU oracle, %s (%s)" % (
                f["subset"], "the frame's exact count" if exact else "%.2f flops per ray of that subset x this frame's %d rays" % (f["flops_per_ray"], rays)),
            "flops_per_ray": round(f["flops_per_ray"], 2), "achieved_tflops_fp64": round(tf, 3),
            "frac_of_fp64_valu_peak": round(tf / FP64_VALU_PEAK_TFLOPS, 4),
            "flops_are": "the operations of the REFERENCE's algorithm (both-children tree traversal, every Newton solve of every ray in the "
                         "box); the device executes fewer (pruned walks, height-field walk, shell cull), so this is an equivalent rate and "
                         "may exceed the peak where those exact shortcuts remove most of the work (C5)"}


def profile_replay(cfg, n):
    """PMC figures of the committed rocprofv3 passes of this same command (not measured in this run)."""
    if cfg["name"] != "c2" or n != 1:
        return None, None
    for fn in ("r03_pmc.json", "r02_pmc.json", "r01_pmc.json"):
        ppath = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(ppath):
            continue
        try:
            prof = json.load(open(ppath))
            cn = prof["counters"]
            simd_cycles = 1024 * cn["GRBM_GUI_ACTIVE"]["mean"] / 8.0
            src = "profiles/%s (rocprofv3 --pmc passes of `python bench.py`, committed; NOT measured in this run)" % fn
            traffic = {"hbm_bytes_per_launch": prof.get("hbm_bytes_per_launch"), "source": src}
            valu = {"busy_frac": round(4.0 * cn["SQ_ACTIVE_INST_VALU"]["mean"] / simd_cycles, 3),
                    "lanes_active_frac": round(cn["SQ_THREAD_CYCLES_VALU"]["mean"] / (64.0 * cn["SQ_ACTIVE_INST_VALU"]["mean"]), 3),
                    "wave_insts_per_launch": int(cn["SQ_INSTS_VALU"]["mean"]), "source": src}
            return traffic, valu
        except Exception:
            continue
    return None, None


# ---------------------------------------------------------------------------------------------------------------------
# launcher: plain `python bench.py --gpus N` spawns its own ranks
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    """Parent side.  Nothing here imports torch or initialises a GPU; children are fresh interpreters (no fork of GPU
    state, no exec after GPU init)."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
    env["WORLD_SIZE"] = str(n)
    env["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e))

    def stop_rest(grace=5.0):
        """terminate exactly the children we started that still run; kill the ones that ignore it"""
        live = [q for q in procs if q.poll() is None]
        for q in live:
            q.terminate()
        t_end = time.monotonic() + grace
        for q in live:
            try:
                q.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    # All children are watched together: a rank that dies first (out of memory, RCCL init failure) would otherwise leave
    # the others waiting in a collective -- and this parent waiting on rank 0 -- until the backend's own timeout.
    worst = 0
    try:
        while True:
            codes = [q.poll() for q in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                worst = bad[0]
                stop_rest()
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        stop_rest()
        raise
    return worst


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frames (default: 20 for c2, 5 for c4, 2 for c5)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["c2", "c4", "c5"], default="c2")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None, help="c2 only (c4 is strong, c5 weak by definition)")
    ap.add_argument("--cpu-spp", type=int, default=None, help="spp of the CPU baseline sample (0 = skip)")
    ap.add_argument("--stripe-rows", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="override the configuration's spp (marks the line as NOT the named config)")
    ap.add_argument("--check", action="store_true", help="verify a crop of the frame against the oracle (c2, N = 1)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the secondary single-frame measurements")
    ap.add_argument("--rehearse", action="store_true", help="no GPU: pattern instead of the kernel, gloo (control path only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    n = world
    if os.environ.get("CGRT_BENCH_FAIL_RANK") == str(rank):  # test hook (tests/test_bench_launch.py): this rank dies before rendezvous
        sys.exit(7)
    cfg = resolve(args, n)
    steps = args.steps if args.steps is not None else cfg["steps"]
    warmup = args.warmup if args.warmup is not None else cfg["warmup"]

    import torch
    import torch.distributed as dist

    from cgraytracing_amd.dist import StripedRenderer, global_row

    rehearse = args.rehearse
    if not rehearse and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback; --rehearse exercises the control path only)")
    # one rank per GPU; CGRT_BENCH_BACKEND=gloo lets several ranks share a GPU to rehearse the N > 1 control path
    backend = "gloo" if rehearse else os.environ.get("CGRT_BENCH_BACKEND", "nccl")
    if rehearse:
        dev_index, dev = 0, torch.device("cpu")
    else:
        dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    if n > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    W, H, SPP = cfg["W"], cfg["H"], cfg["spp"]
    sr = StripedRenderer(W, H, stripe_rows=cfg["stripe_rows"], nshares=cfg["shares"])
    rows_local = sr.rows_local
    if rehearse:
        scene = cam = stats = None
        counters = torch.zeros(8, dtype=torch.int64)
        variant = "rehearsal (no kernel)"
    else:
        import cgraytracing_amd as cg
        import scenes
        scene = cg.Scene(build_scene_objects(cfg["name"]), device=dev_index)
        cam = scenes.cam_dof()
        stats = scene.stats()
        counters = torch.zeros(8, dtype=torch.int64, device=dev)
        variant = scene.kernel_variant(W, H, SPP, cam, DEPTH, rows=rows_local, stripe=sr.stripe)

    # Two output buffers: with N > 1 the gather of frame k (side stream) overlaps the render of frame k+1.
    outs = [torch.zeros((rows_local, W, 3), dtype=torch.float32, device=dev) for _ in range(2 if n > 1 else 1)]
    kernel_events = []
    comm = torch.cuda.Stream(device=dev) if (n > 1 and not rehearse) else None
    done_evt = [None if rehearse else torch.cuda.Event() for _ in outs]
    free_evt = [None for _ in outs]
    state = {"k": 0}

    def fake_render(out):  # rehearsal: every pixel carries its global row, so a mis-mapped stripe shows
        gr = torch.tensor([global_row(j, sr.stripe_rows, sr.rank, sr.nshares) if sr.nshares > 1 else j
                           for j in range(rows_local)], dtype=torch.float32)
        out[:] = torch.where(gr < H, gr, torch.zeros(()))[:, None, None]
        counters[0] += int((gr < H).sum()) * W

    def step(record=False):
        k = state["k"] % len(outs)
        state["k"] += 1
        out = outs[k]
        if rehearse:
            fake_render(out)
            return sr.gather(out)
        cur = torch.cuda.current_stream(dev)
        if free_evt[k] is not None:
            cur.wait_event(free_evt[k])  # the gather that last read this buffer has finished
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        scene.trace_grid(W, H, SPP, cam, DEPTH, SEED, rows=rows_local, stripe=sr.stripe, out=out, nhit=False,
                         counters=counters)
        if record:
            e1.record()
            kernel_events.append((e0, e1))
        if n == 1:
            return out
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
        if not rehearse:
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    counters.zero_()
    fence()
    t0 = time.perf_counter()
    frame = None
    for _ in range(steps):
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
    kern_ms = sum(a.elapsed_time(b) for a, b in kernel_events) / max(1, len(kernel_events)) if kernel_events else None

    if rank == 0:
        rays_per_step = total_rays / max(1, steps)
        value = total_rays / dt / 1e6
        line = {
            "metric": cfg["metric"], "value": round(value, 2), "unit": "Mrays/s", "n_gpus": n, "steps": steps,
            "warmup": warmup, "ms_per_step": round(dt / max(1, steps) * 1e3, 4), "higher_is_better": True,
            "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["workload"], "name": cfg["name"], "width": W, "height": H,
                       "rows_per_gpu": rows_local, "spp": SPP, "max_depth": DEPTH, "rays_per_step": int(rays_per_step),
                       "sharding": ("rows, block-cyclic %d-row stripes over %d shares, gather to rank 0 (%s)" %
                                    (cfg["stripe_rows"], cfg["shares"], backend)) if cfg["shares"] > 1 else "single GPU"},
        }
        if rehearse:
            line["rehearsal"] = True
            line["value"] = None
            line["frame_rows_ok"] = bool(frame is not None and frame.shape[0] == H and _rehearsal_frame_ok(frame, cfg, n))
        else:
            # algorithmic HBM bytes of one launch on one GPU: fp32 RGB store + scene read once
            alg_bytes = 12 * W * rows_local + stats["scene_bytes_fp64"]
            if not kern_ms:  # --steps 0: nothing was timed
                print(json.dumps(dict(line, value=None, roofline=None)), flush=True)
                if n > 1:
                    dist.barrier()
                    dist.destroy_process_group()
                return
            achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
            traffic_prof, valu_prof = profile_replay(cfg, n)
            line["roofline"] = {
                "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": None,
                "kernel": variant, "kernel_ms": round(kern_ms, 4), "algorithmic_bytes": int(alg_bytes),
                "note": "FP64-VALU/divergence bound by design (SURVEY.md 8d H5); HBM fraction reported as required. "
                        "kernel_ms = HIP events around each launch of this run%s" %
                        (" (+ the chunk-sum finalize kernel: Bezier scenes split a tile's samples)" if cfg["name"] == "c5" else ""),
                "kernel_mrays_per_s": round(rays_per_step / n / (kern_ms * 1e-3) / 1e6, 2),
                "lane_utilisation": round(total_rays / max(1, 64 * int(cnt[2].item())), 4),
                "traffic_from_profile": traffic_prof, "valu_from_profile": valu_prof,
                "stated_bound": "fp64 VALU issue (see roofline_valu): HBM is reported because the north star asks for it"}
            # the roof that binds: fp64 vector ALU.  flops: counted exactly by the instrumented CPU oracle for this frame
            # (profiles/r03_flops.json); instructions: SQ_INSTS_VALU of the committed PMC pass of this same command.
            rv = {"bound": "fp64_valu", "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "kernel_ms": round(kern_ms, 4)}
            ff = flop_fields(load_flops(), cfg["name"], int(rays_per_step / n), kern_ms)
            if ff:
                rv.update({"achieved": ff["achieved_tflops_fp64"], "frac": ff["frac_of_fp64_valu_peak"],
                           "algorithmic_flops": ff["algorithmic_flops"], "flops_per_ray": ff["flops_per_ray"],
                           "algorithmic_flops_source": ff["algorithmic_flops_source"]})
            else:
                rv.update({"achieved": None, "frac": None})
            if valu_prof:
                gi = valu_prof["wave_insts_per_launch"] / (kern_ms * 1e-3) / 1e9
                rv["valu_issue"] = {"wave_insts_per_launch": valu_prof["wave_insts_per_launch"], "achieved_ginst_per_s": round(gi, 1),
                                    "peak_ginst_per_s": VALU_ISSUE_PEAK_GINST, "frac": round(gi / VALU_ISSUE_PEAK_GINST, 4),
                                    "peak_note": "1024 SIMDs x 2.4 GHz / 4 cycles per fp64 wave-instruction",
                                    "source": valu_prof["source"] + "; time = this run's HIP events"}
            line["roofline_valu"] = rv
            cpu_spp = args.cpu_spp
            if cpu_spp is None or cpu_spp > 0:
                with stdout_to_stderr():
                    base = cpu_baseline(cfg, cpu_spp)
                line["cpu_baseline"] = base
                line["gpu_over_cpu"] = round(value / base["value"], 1)
                if cfg["name"] == "c2":
                    try:
                        with stdout_to_stderr():
                            line["cpu_baseline_all_cores"] = cpu_baseline(cfg, cpu_spp, threads=os.cpu_count() or 1)
                    except Exception as e:  # pragma: no cover
                        line["cpu_baseline_all_cores"] = {"error": str(e)}
            if n == 1 and cfg["name"] == "c2" and not args.no_other_configs:
                try:
                    with stdout_to_stderr():
                        line["other_configs"] = other_configs(dev_index)
                except Exception as e:  # pragma: no cover - never let a secondary measurement lose the headline line
                    line["other_configs"] = {"error": repr(e)}
            if args.check and n == 1 and cfg["name"] == "c2":
                import numpy as np
                from backends import Backend, BackendScene, to_acc32
                be = Backend("orc")
                be.set_threads(os.cpu_count() or 1)
                osc = BackendScene(be, build_scene_objects("c2"))
                r0 = 200
                want = osc.trace_grid(cam, W, H, SPP, DEPTH, SEED, row0=r0, nrows=16)
                got = frame[r0:r0 + 16].cpu().numpy()
                line["check_linf_rows_200_215"] = float(np.abs(got - to_acc32(want["acc_sum"], SPP)).max())
        print(json.dumps(line), flush=True)
    if n > 1:
        dist.barrier()
        dist.destroy_process_group()


def _rehearsal_frame_ok(frame, cfg, n):
    """Rehearsal: row h of the assembled frame holds h where a rank rendered it, 0 where no rank did (c5 with N < 8)."""
    import torch
    S, shares, H = cfg["stripe_rows"], cfg["shares"], cfg["H"]
    h = torch.arange(H, dtype=torch.float32)
    owner = (torch.arange(H) // S) % shares
    want = torch.where(owner < n, h, torch.zeros(())) if shares > 1 else h
    return bool(torch.equal(frame[:, 0, 0].cpu(), want) and torch.equal(frame[:, -1, 2].cpu(), want))


if __name__ == "__main__":
    main()

"""CPU: bench.py's own launcher and multi-rank control path (no GPU, gloo).  `python bench.py --gpus N --rehearse` must
spawn its N ranks itself (no torchrun), rendezvous on 127.0.0.1, deal the stripes of every configuration, gather and
un-permute the frame on rank 0 and print exactly one JSON line on stdout.  The kernel launch is replaced by a closed-form
row pattern (bench.py --rehearse); everything else is the code the GPU run uses."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=timeout)
    assert p.returncode == 0, p.stdout + p.stderr
    # ONE JSON line, from rank 0 (the gloo transport itself announces its connections on stdout; RCCL does not)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("args,n,W,H,rows,scaling", [
    (["--gpus", "2", "--config", "c2"], 2, 2720, 1528, 768, "weak"),            # same view, 2 x the pixels
    (["--gpus", "2", "--config", "c2", "--scaling", "strong"], 2, 1920, 1080, 544, "strong"),
    (["--gpus", "2", "--config", "c4"], 2, 4096, 4096, 2048, "strong"),         # 16-row stripes
    (["--gpus", "2", "--config", "c5"], 2, 8192, 8192, 1024, "weak"),           # 2 of 8 shares
    (["--gpus", "1", "--config", "c5"], 1, 8192, 8192, 1024, "weak"),           # 1 of 8 shares, no process group
    (["--gpus", "1"], 1, 1920, 1080, 1080, "weak"),
])
def test_self_launch_rehearsal(args, n, W, H, rows, scaling):
    line = _run(args + ["--rehearse", "--steps", "2", "--warmup", "1"])
    assert line["rehearsal"] is True and line["n_gpus"] == n and line["scaling"] == scaling
    c = line["config"]
    assert (c["width"], c["height"], c["rows_per_gpu"]) == (W, H, rows)
    assert line["frame_rows_ok"] is True
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data"):
        assert k in line


def test_weak_dims_keep_view_and_work():
    sys.path.insert(0, ROOT)
    import bench
    for n in (1, 2, 4, 8):
        W, H = bench.weak_dims(n)
        assert W % 32 == 0 and H % 8 == 0
        assert abs(W * H / (1920 * 1080 * n) - 1) < 0.003      # per-GPU work fixed
        assert abs((W / H) / (1920 / 1080) - 1) < 0.002        # same aspect ratio, hence the same view


def test_external_launcher_env_is_respected():
    """The driver's form: ranks started by an external launcher (WORLD_SIZE etc. set) must not self-launch again."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "1",
                                       "--warmup", "0", "--config", "c4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    noise = lambda t: [ln for ln in t.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]
    assert noise(outs[1][0]) == []
    assert json.loads(noise(outs[0][0])[0])["frame_rows_ok"] is True


def test_dead_rank_stops_the_launch_quickly():
    """a rank that exits non-zero first must not leave the parent waiting on rank 0 (which sits in the rendezvous /
    a collective until the backend's timeout): the parent polls all children, stops the rest and returns non-zero at once."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["CGRT_BENCH_FAIL_RANK"] = "1"
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=120)
    dt = time.monotonic() - t0
    assert p.returncode == 7, (p.returncode, p.stdout, p.stderr)
    assert dt < 60, dt  # import torch + rendezvous start dominate; the gloo rendezvous timeout would be 30 min


def test_cpu_baseline_falls_back_to_the_port(monkeypatch):
    """`cpu_baseline.kind` is "reference" only where oracle/_ref/libcgrt_ref.so travelled with the tree; everywhere else the CPU
    oracle port is timed and the line says so (keep that fallback tested).  A one-sample pass of the C2 frame."""
    sys.path.insert(0, ROOT)
    import backends
    import bench
    monkeypatch.setattr(backends, "have_ref", lambda: False)
    cfg = dict(name="c2", W=1920, H=1080, spp=64)
    out = bench.cpu_baseline(cfg, 1)
    assert out["kind"] == "port" and out["ref_library_found"] is False and "NOT FOUND" in out["ref_library"]
    assert out["cores"] == 1 and out["value"] > 0.1 and "spp=1" in out["sample"]

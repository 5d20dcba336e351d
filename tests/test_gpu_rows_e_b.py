"""GPU: what rows (b) and (e) of SURVEY.md section 8 promise and ONE GPU can show.

(b) "Threading: one host thread per GPU, callable concurrently" (include/cgrt.h): two host threads drive two scenes through the
    C ABI at the same time, each on its own stream of the same device, through the cost-scheduled path whose per-handle scratch
    grows from launch to launch; every frame must equal the one the same call gives alone.
(e) the un-permute that follows the framebuffer gather runs on the device (cgrt_unpermute_stripes); the C++ sharded host program
    goes through it with its shares rendered one after another on GPU 0 (--emulate: everything of the N > 1 path but RCCL)."""
import os
import subprocess
import threading

import numpy as np
import pytest

import scenes
from backends import BackendScene, to_acc32

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_host_threads_two_streams_one_device(gpu_ready):
    import torch
    import cgraytracing_amd as cg
    jobs = [("c2", scenes.scene_c2(), dict(force_reorder=True), 192, 112), ("c3", scenes.scene_c3(True), {}, 160, 128)]
    spps = [4, 8, 32, 16, 64, 8]  # the handle's deferred buffer is sized by spp: grows, is reused, grows again
    cam = scenes.cam_dof()

    def run(objs, kw, W, H, stream, barrier=None):
        sc = cg.Scene(objs)
        outs = []
        for spp in spps:
            if barrier is not None:
                barrier.wait()  # both threads enter the library together
            cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
            with torch.cuda.stream(stream):
                rgb, nhit, _ = sc.trace_grid(W, H, spp, cam, 5, 12345, counters=cnt, stream=stream.cuda_stream, **kw)
            stream.synchronize()
            outs.append((rgb.cpu().numpy().copy(), nhit.cpu().numpy().copy(), cnt.cpu().numpy().copy()))
        sc.close()
        return outs

    alone = [run(objs, kw, W, H, torch.cuda.Stream()) for _, objs, kw, W, H in jobs]
    for rounds in range(2):
        barrier = threading.Barrier(2)
        got, errs = [None, None], []

        def worker(i):
            try:
                _, objs, kw, W, H = jobs[i]
                got[i] = run(objs, kw, W, H, torch.cuda.Stream(), barrier)
            except BaseException as e:  # noqa: BLE001 - reported below, and the other thread must not hang on the barrier
                errs.append(e)
                barrier.abort()

        th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        assert not errs, errs
        for i, (name, *_r) in enumerate(jobs):
            for k, spp in enumerate(spps):
                for a, b, what in zip(got[i][k], alone[i][k], ("rgb", "nhit", "counters")):
                    if what == "counters":
                        a, b = a[:2], b[:2]  # rays, hitpoints (wave iterations depend on which lane took which unit)
                    assert np.array_equal(a, b), "%s spp=%d %s differs between concurrent and single-thread launches" % (name, spp, what)


def test_unpermute_stripes_on_device(gpu_ready):
    import ctypes as C
    import torch
    from cgraytracing_amd import _capi
    from cgraytracing_amd.dist import assemble, local_rows
    L = _capi.lib()
    g = torch.Generator(device="cpu").manual_seed(3)
    for W, H, S, nshares, present, ch in [(96, 88, 8, 3, 3, 3), (130, 100, 16, 8, 2, 3), (64, 64, 8, 2, 2, 1), (33, 17, 8, 4, 4, 3)]:
        rows = local_rows(H, S, 0, nshares)
        shares = torch.rand((present, rows, W, ch), generator=g).cuda()
        frame = torch.full((H, W, ch), -1.0, device="cuda")
        _capi.check(L.cgrt_unpermute_stripes(shares.data_ptr(), present, nshares, W, H, S, rows, ch, frame.data_ptr(),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        want = assemble(shares, H, S, nshares)
        assert torch.equal(frame, want), (W, H, S, nshares, present)
    bad = L.cgrt_unpermute_stripes(shares.data_ptr(), 5, 4, 8, 8, 8, 8, 3, frame.data_ptr(), None)
    assert bad == -1


@pytest.mark.parametrize("shares,stripe", [(3, 8), (8, 16)])
def test_cpp_sharded_host_program_emulated_shares(gpu_ready, orc, tmp_path, shares, stripe):
    """include/cgrt_host_sharded.hpp with its stripes dealt to several shares, rendered one after another on GPU 0 and
    assembled by the device un-permute: the oracle's frame, bit for bit.  (The RCCL transfers between real GPUs remain
    unverified on this one-GPU box.)"""
    exe = os.path.join(ROOT, "cgraytracing_amd", "cgrt_sharded")
    assert os.path.exists(exe), "build it with make -C cgraytracing_amd/csrc all"
    raw = str(tmp_path / "c2.f32")
    W, H, spp = 160, 104, 4
    out = subprocess.run([exe, "--emulate", str(shares), "--stripe", str(stripe), "--width", str(W), "--height", str(H), "--spp", str(spp),
                          "--dof", "--raw", raw], capture_output=True, text=True, check=True).stdout
    want = BackendScene(orc, scenes.scene_c2()).trace_grid(scenes.cam_dof(), W, H, spp, 5, seed=12345)
    got = np.fromfile(raw, np.float32).reshape(H, W, 3)
    assert np.array_equal(got, to_acc32(want["acc_sum"], spp))
    assert "gpus: %d rays: %d " % (shares, want["nrays"]) in out

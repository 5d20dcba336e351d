"""CPU, world_size 2 (gloo): the row sharding, gather and un-permute of cgraytracing_amd.dist, with the
kernel launch replaced by a closed-form 'image' of the global row/column so that any mis-mapped row shows."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cgraytracing_amd.dist import StripedRenderer, assemble, global_row, local_rows


def _pattern(rows_global, W):
    r = torch.as_tensor(rows_global, dtype=torch.float32)[:, None, None]
    c = torch.arange(W, dtype=torch.float32)[None, :, None]
    ch = torch.arange(3, dtype=torch.float32)[None, None, :]
    return r * 1000.0 + c + ch * 0.25


def _worker(rank, world, port, W, H, S, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def render_local(rows, stripe):
            s_rows, s_rank, s_n = stripe
            gr = [global_row(j, s_rows, s_rank, s_n) for j in range(rows)]
            img = _pattern(gr, W)
            img[torch.as_tensor(gr) >= H] = 0  # rows past the image are left zero, like the kernel
            return img.contiguous()

        sr = StripedRenderer(W, H, stripe_rows=S, render_local=render_local)
        assert sr.nranks == world and sr.rows_local == local_rows(H, S, rank, world)
        for _ in range(2):  # twice: the gather buffer is reused
            frame = sr.frame()
        if rank == 0:
            q.put(frame.numpy())
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("W,H,S", [(16, 64, 8), (5, 37, 8), (8, 48, 16)])
def test_striped_gather_world2(W, H, S):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, H, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    frame = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = _pattern(list(range(H)), W).numpy()
    assert frame.shape == (H, W, 3)
    assert np.array_equal(frame, want)


def test_stripe_mapping_is_a_partition():
    for H, S, N in [(1080, 8, 8), (1080, 16, 4), (37, 8, 2), (4096, 16, 8), (8, 8, 8)]:
        seen = []
        for r in range(N):
            rows = local_rows(H, S, r, N)
            assert rows % S == 0
            seen += [global_row(j, S, r, N) for j in range(rows)]
        inside = sorted(g for g in seen if g < H)
        assert inside == list(range(H)), (H, S, N)
        assert len(set(seen)) == len(seen)


def test_assemble_single_rank_is_identity():
    x = torch.arange(2 * 3 * 4 * 3, dtype=torch.float32).reshape(1, 6, 4, 3)
    assert torch.equal(assemble(x, 5, 8, 1), x[0][:5])


class _FakePpmScene:
    """Stands in for cgraytracing_amd.Scene under gloo (no GPU): ppm_render returns a closed-form float64 'image' of
    the global row, so render_ppm_striped's argument plumbing (rows, stripe) and the float64 gather are exercised."""
    device = 0

    def ppm_render(self, W, H, spp, camera, max_depth, seed, rows=None, stripe=None, **kw):
        assert kw == {"nphotons": 1234}
        s_rows, s_rank, s_n = stripe
        gr = [global_row(j, s_rows, s_rank, s_n) for j in range(rows)]
        img = _pattern(gr, W).double() * (1.0 + 1e-12)  # needs float64 to survive
        img[torch.as_tensor(gr) >= H] = 0
        return {"image": img.numpy()}


def _ppm_worker(rank, world, port, W, H, S, q):
    from cgraytracing_amd.dist import render_ppm_striped
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sr = StripedRenderer(W, H, stripe_rows=S)
        frame = render_ppm_striped(_FakePpmScene(), sr, spp=2, nphotons=1234)
        if rank == 0:
            q.put(frame.numpy())
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_striped_ppm_gather_world2():
    """Row f1 across ranks (cgraytracing_amd.dist.render_ppm_striped): float64 image rows gathered and un-permuted."""
    W, H, S = 12, 40, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ppm_worker, args=(r, 2, port, W, H, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    frame = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = (_pattern(list(range(H)), W).double() * (1.0 + 1e-12)).numpy()
    assert frame.dtype == np.float64 and np.array_equal(frame, want)


def test_assemble_partial_shares():
    """Weak scaling of a frame defined on 8 shares (bench.py --config c5): with fewer ranks than shares the gathered buffers
    fill only their own stripes; the stripes of absent shares stay zero and the row order is the global one."""
    S, shares, n, W, H = 16, 8, 3, 4, 16 * 8 * 2
    rows_local = local_rows(H, S, 0, shares)
    gathered = torch.stack([_pattern([global_row(j, S, r, shares) for j in range(rows_local)], W) for r in range(n)])
    frame = assemble(gathered, H, S, shares)
    want = _pattern(list(range(H)), W)
    owner = (torch.arange(H) // S) % shares
    want[owner >= n] = 0
    assert frame.shape == (H, W, 3) and torch.equal(frame, want)
    # all shares present: the plain permutation
    full = torch.stack([_pattern([global_row(j, S, r, shares) for j in range(rows_local)], W) for r in range(shares)])
    assert torch.equal(assemble(full, H, S, shares), _pattern(list(range(H)), W))


def test_striped_renderer_shares_without_process_group():
    """One process, 8 shares (bench.py --config c5 --gpus 1): the renderer owns share 0, gather() returns the partial frame."""
    sr = StripedRenderer(4, 256, stripe_rows=16, nshares=8)
    assert sr.nranks == 1 and sr.nshares == 8 and sr.rows_local == 32 and sr.stripe == (16, 0, 8)
    local = _pattern([global_row(j, 16, 0, 8) for j in range(32)], 4)
    frame = sr.gather(local)
    assert frame.shape == (256, 4, 3)
    assert torch.equal(frame[:16], local[:16]) and torch.equal(frame[128:144], local[16:]) and not frame[16:128].any()

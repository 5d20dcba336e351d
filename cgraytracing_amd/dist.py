"""Multi-GPU eye pass: one process per GPU, image rows sharded in block-cyclic stripes, framebuffer gathered
to rank 0 with one collective per frame (RCCL over xGMI when the backend is "nccl").

Why stripes: rays that enter a mesh's bounding box cost 40-200x a miss and meshes sit in the lower third of
the frame, so contiguous H/N bands are badly imbalanced.  Stripe s of `stripe_rows` rows
belongs to rank s % N; each rank's local buffer is the concatenation of its stripes, so the gather moves one
contiguous [rows_local, W, 3] fp32 block per rank (7 receives on 7 distinct xGMI links at N = 8) and rank 0
un-permutes with a single strided copy.  The scene (<= tens of MB) is replicated; there is no other exchange.
"""
from __future__ import annotations

import math


def local_rows(height, stripe_rows, rank, nranks):
    """Rows of the local buffer of `rank` (all ranks use the same, padded, count so the gather is regular)."""
    if nranks <= 1:
        return height
    nstripes = math.ceil(height / stripe_rows)
    per_rank = math.ceil(nstripes / nranks)
    return per_rank * stripe_rows


def global_row(j, stripe_rows, rank, nranks, row_offset=0):
    """Same mapping as the kernel (cgrt.h, cgrt_grid): local row j -> global row."""
    if nranks <= 1:
        return row_offset + j
    return ((j // stripe_rows) * nranks + rank) * stripe_rows + (j % stripe_rows)


def assemble(gathered, height, stripe_rows, nshares, out=None):
    """gathered: tensor [n, rows_local, W, C] (rank-major), the buffers of shares 0..n-1 of `nshares` (n == nshares when
    every share has a rank).  Returns [height, W, C] in global row order; rows of shares nobody rendered are zero.
    `out`: optional preallocated [per_share, nshares, S, W, C] scratch for the partial case."""
    if nshares <= 1:
        return gathered[0][:height]
    n, rows_local, W, Cc = gathered.shape
    per = rows_local // stripe_rows
    # [share, stripe_in_share, S, W, C] -> [stripe_in_share, share, S, W, C]: global stripe = k*nshares + share
    x = gathered.reshape(n, per, stripe_rows, W, Cc).permute(1, 0, 2, 3, 4)
    if n < nshares:
        if out is None:
            out = gathered.new_zeros((per, nshares, stripe_rows, W, Cc))
        out[:, :n] = x
        x = out
    return x.reshape(per * nshares * stripe_rows, W, Cc)[:height]


class StripedRenderer:
    """Renders one W x H frame across the ranks of a torch.distributed process group.

    render_local(rows_local, stripe) -> tensor [rows_local, W, 3] produces this rank's stripes; by default it
    launches cgrt_trace_grid on this rank's GPU.  It is injectable so the sharding / gather logic can be
    exercised with the gloo backend on CPU-only machines.

    nshares (default: the number of ranks) is the number of shares the stripes are dealt to; rank r renders share r.
    nshares > ranks renders only the first `ranks` shares of the frame (weak scaling of a frame defined on nshares GPUs:
    every GPU's work and ray mix are those of the full configuration whatever the number of GPUs present)."""

    def __init__(self, width, height, stripe_rows=8, group=None, render_local=None, device=None, nshares=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.W, self.H = int(width), int(height)
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.nranks = dist.get_world_size(group) if self.distributed else 1
        if stripe_rows % 8:
            raise ValueError("stripe_rows must be a multiple of 8 (kernel tile height)")
        self.stripe_rows = int(stripe_rows)
        self.nshares = int(nshares) if nshares else self.nranks
        if self.nshares < self.nranks:
            raise ValueError("nshares must be >= the number of ranks")
        self.rows_local = local_rows(self.H, self.stripe_rows, self.rank, self.nshares)
        self.render_local = render_local
        self.device = device
        self._gather_buf = None
        self._partial = None

    @property
    def stripe(self):
        return (self.stripe_rows, self.rank, self.nshares) if self.nshares > 1 else None

    def _assemble(self, gathered):
        if gathered.shape[0] < self.nshares:  # scratch for the partial frame: the rows of absent shares stay zero
            shape = (gathered.shape[1] // self.stripe_rows, self.nshares, self.stripe_rows) + tuple(gathered.shape[2:])
            if self._partial is None or tuple(self._partial.shape) != shape or self._partial.device != gathered.device:
                self._partial = gathered.new_zeros(shape)
        return assemble(gathered, self.H, self.stripe_rows, self.nshares, out=self._partial)

    def gather(self, local):
        """One gather of the local [rows_local, W, 3] blocks to rank 0; returns the assembled frame on rank 0
        (None elsewhere)."""
        import torch

        if self.nranks == 1:
            return local[: self.H] if self.nshares == 1 else self._assemble(local[None])
        if local.is_cuda and self.dist.get_backend(self.group) == "gloo":
            local = local.cpu()  # rehearsal on machines without RCCL-capable peers: gloo gathers host tensors
        if self.rank == 0:
            if self._gather_buf is None or self._gather_buf.shape[1:] != local.shape or \
                    self._gather_buf.device != local.device:
                self._gather_buf = torch.empty((self.nranks,) + tuple(local.shape), dtype=local.dtype,
                                               device=local.device)
            lst = list(self._gather_buf.unbind(0))
            self.dist.gather(local, gather_list=lst, dst=0, group=self.group)
            return self._assemble(self._gather_buf)
        self.dist.gather(local, gather_list=None, dst=0, group=self.group)
        return None

    def frame(self):
        local = self.render_local(self.rows_local, self.stripe)
        return self.gather(local)


def render_ppm_striped(scene, renderer, spp=1, camera=None, max_depth=5, seed=12345, **photon_args):
    """The whole of render() (eye pass, photon pass, final gather) across the ranks of
    `renderer` (a StripedRenderer): every rank traces all photons -- photon paths do not depend on hitpoints -- but
    owns only the hitpoints of its stripes, so the expensive search / replay stages shard N ways and the assembled frame
    equals a single-GPU render bit for bit.  Returns the [H, W, 3] float64 image on rank 0 (None elsewhere)."""
    import torch

    r = scene.ppm_render(renderer.W, renderer.H, spp, camera, max_depth, seed, rows=renderer.rows_local,
                         stripe=renderer.stripe, **photon_args)
    local = torch.from_numpy(r["image"])
    if renderer.distributed and renderer.dist.get_backend(renderer.group) == "nccl":
        local = local.to(torch.device("cuda", scene.device))
    return renderer.gather(local)

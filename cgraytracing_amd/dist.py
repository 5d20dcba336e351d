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


def assemble(gathered, height, stripe_rows, nranks):
    """gathered: tensor [nranks, rows_local, W, C] (rank-major).  Returns [height, W, C] in global row order."""
    if nranks <= 1:
        return gathered[0][:height]
    n, rows_local, W, Cc = gathered.shape
    per_rank = rows_local // stripe_rows
    # [rank, stripe_in_rank, S, W, C] -> [stripe_in_rank, rank, S, W, C]: global stripe = k*nranks + rank
    x = gathered.reshape(n, per_rank, stripe_rows, W, Cc).permute(1, 0, 2, 3, 4)
    return x.reshape(per_rank * n * stripe_rows, W, Cc)[:height]


class StripedRenderer:
    """Renders one W x H frame across the ranks of a torch.distributed process group.

    render_local(rows_local, stripe) -> tensor [rows_local, W, 3] produces this rank's stripes; by default it
    launches cgrt_trace_grid on this rank's GPU.  It is injectable so the sharding / gather logic can be
    exercised with the gloo backend on CPU-only machines."""

    def __init__(self, width, height, stripe_rows=8, group=None, render_local=None, device=None):
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
        self.rows_local = local_rows(self.H, self.stripe_rows, self.rank, self.nranks)
        self.render_local = render_local
        self.device = device
        self._gather_buf = None

    @property
    def stripe(self):
        return (self.stripe_rows, self.rank, self.nranks) if self.nranks > 1 else None

    def gather(self, local):
        """One gather of the local [rows_local, W, 3] blocks to rank 0; returns the assembled frame on rank 0
        (None elsewhere)."""
        import torch

        if self.nranks == 1:
            return local[: self.H]
        if local.is_cuda and self.dist.get_backend(self.group) == "gloo":
            local = local.cpu()  # rehearsal on machines without RCCL-capable peers: gloo gathers host tensors
        if self.rank == 0:
            if self._gather_buf is None or self._gather_buf.shape[1:] != local.shape or \
                    self._gather_buf.device != local.device:
                self._gather_buf = torch.empty((self.nranks,) + tuple(local.shape), dtype=local.dtype,
                                               device=local.device)
            lst = list(self._gather_buf.unbind(0))
            self.dist.gather(local, gather_list=lst, dst=0, group=self.group)
            return assemble(self._gather_buf, self.H, self.stripe_rows, self.nranks)
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

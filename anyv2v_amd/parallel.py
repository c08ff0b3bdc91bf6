"""Multi-GPU sharding of the clip / edit list.

The reference loops over ``configs_list`` in one process on one GPU (``run_group_pnp_edit.py:74``); entries are
independent, so here entry ``i`` goes to rank ``i % world`` (one process per GPU, full weight replica) and the ONLY
collective is one ``all_gather`` of the edited latents at the end (RCCL over xGMI on MI355X; ``gloo`` in the CPU
tests).  No communication happens inside the denoising loops.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend: Optional[str] = None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, local_rank, world = dist_env()
    if world == 1:
        return rank, local_rank, world
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    return rank, local_rank, world


def shard_entries(entries: Sequence, rank: int, world: int) -> List:
    """Active entries are dealt round-robin; inactive ones are dropped first so that the load stays balanced."""
    active = [e for e in entries if e.get("active", True) is not False]
    return [e for i, e in enumerate(active) if i % world == rank]


def shard_by_clip(inv_entries: Sequence, edit_entries: Sequence, rank: int, world: int):
    """Clip-wise dealing for the fused runner: the active inversion entries round-robin, and every edit entry to the rank that inverts
    its clip (``video_name``), so that a rank holds whole clips and can pipeline them (``run_group_anyv2v.main_pipelined``); edits of a
    clip nobody inverts here are dealt round-robin among themselves.  Returns (this rank's inversion entries, its edit entries)."""
    inv = [e for e in inv_entries if e.get("active", True) is not False]
    owner = {}
    for i, e in enumerate(inv):
        owner.setdefault(e.get("video_name"), i % world)
    edits = [e for e in edit_entries if e.get("active", True) is not False]
    mine, loose = [], 0
    for e in edits:
        r = owner.get(e.get("video_name"))
        if r is None:
            r, loose = loose % world, loose + 1
        if r == rank:
            mine.append(e)
    return [e for i, e in enumerate(inv) if i % world == rank], mine


def seed_for_entry(base_seed: int, entry_index: int) -> int:
    """The reference seeds once per process and draws per entry from the global RNG (``run_group_pnp_edit.py:124,213``),
    which ties results to the processing order.  Sharded runs re-seed per entry so that any rank reproduces the
    same numbers for the same entry regardless of world size."""
    return int(base_seed) + 1000003 * int(entry_index)


def gather_latents(latents: Sequence[torch.Tensor], n_entries: int, like_shape, dtype, device, indices: Sequence[int] = None) -> torch.Tensor:
    """The ONE collective of the sharded job: all_gather of every rank's edited latents (``[1,4,F,h,w]`` per entry,
    512 KiB at 16f x 512^2; RCCL over xGMI on the node, gloo in the CPU tests).

    ``latents``: this rank's results; ``indices``: the entry index of each (any dealing: round-robin ``shard_entries``, clip-wise
    ``shard_by_clip``, unbalanced).  Without ``indices`` the round-robin dealing is assumed (entry ``rank``, ``rank + world``, ...).
    A rank may hold several entries (14 demo edits on 8 GPUs) or none: the ranks first exchange their index lists, every rank then
    contributes ``max over ranks of its entry count`` slots (zero-padded) in ONE message, and the result is scattered to entry
    order by index.  Returns ``[n_entries, 4, F, h, w]``."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    n_entries = int(n_entries)
    like_shape = tuple(like_shape)[-4:]
    if indices is None:
        indices = [rank + k * world for k in range(len(latents))]
    indices = [int(i) for i in indices]
    # Agree on validity BEFORE the collective: a rank that raised on its own (a clip with another geometry, an index out of range)
    # would leave the others blocked in all_gather until the backend's timeout.  Every rank reports (ok, its indices, its geometry);
    # all of them then raise together, or none does.
    mine = [tuple(x.shape[-4:]) for x in latents]
    ok = len(indices) == len(latents) and all(sh == like_shape for sh in mine) and all(0 <= i < n_entries for i in indices)
    reports = [None] * world
    dist.all_gather_object(reports, (bool(ok), indices, sorted(set(mine)), like_shape))
    held = sorted(i for r in reports for i in r[1])
    if not all(r[0] for r in reports) or len({r[3] for r in reports}) != 1 or held != list(range(n_entries)):
        raise ValueError(f"gather_latents: ranks disagree or hold invalid results (ok, entry indices, geometries, expected) per rank: "
                         f"{reports}; the {n_entries} entries must each be held by exactly one of the {world} ranks and share one geometry")
    slots = max(1, max(len(r[1]) for r in reports))
    buf = torch.zeros((slots,) + like_shape, dtype=dtype, device=device)
    for k, x in enumerate(latents):
        buf[k].copy_(x.reshape(like_shape))
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    where = {i: (r, k) for r, rep in enumerate(reports) for k, i in enumerate(rep[1])}
    return torch.stack([out[where[i][0]][where[i][1]] for i in range(n_entries)])


# --------------------------------------------------------------------------------------------------------------
class FrameParallel:
    """One clip over ``world`` GPUs for the long-clip mode (128 frames, ``gradio_demo.py:129-131``).

    Every rank holds ``F / world`` consecutive frames of every batch element, as token matrices ``[(b f_loc)(h w), C]``.
    Spatial layers (ResNet convs, per-frame GroupNorm, spatial / cross attention, feed-forward) are independent per frame
    and need no communication.  Around each temporal layer (``TemporalConvLayer``, ``TransformerTemporalModel``) the
    tokens are re-sharded frames -> pixels with ONE all-to-all (rank r then holds pixels ``[r HW/world, (r+1) HW/world)``
    of ALL frames: the temporal conv and the temporal attention are independent per pixel), and back with another; the
    5-D GroupNorm statistics inside those layers are the only reduction (``ops.groupnorm(shard=...)``: one all-reduce
    of a few KiB of fp32 partial sums).  The v-prediction (8 channels) is all-gathered at the end of the forward, so the
    scheduler step, the PnP bookkeeping and the pipeline loops run unchanged -- and identically -- on every rank.

    Per 64x64 layer at B=3, F=128 the re-shard moves 7/8 of a 1 GiB activation: 7 x 16 MiB per rank and direction,
    one message per xGMI link -- the pattern the point-to-point fabric is built for (no ring, no NVSwitch assumed).
    Transport: ``torch.distributed`` (backend "nccl" = RCCL); with the ``gloo`` backend device tensors are staged
    through host memory (tests / bring-up only).
    """

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("FrameParallel needs an initialised torch.distributed process group (init_distributed())")
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.host_staged = dist.get_backend(group) == "gloo"
        self.bytes_moved = 0  # all-to-all payload sent by this rank (for the comm accounting in DESIGN.md / tests)

    # ---- geometry
    def check(self, F: int, H: int, W: int, levels: int):
        hw_min = (H >> (levels - 1)) * (W >> (levels - 1))
        if F % self.world or hw_min % self.world or hw_min == 0:
            raise ValueError(f"frame-parallel clip: F={F} and the coarsest level's {hw_min} pixels must both be multiples "
                             f"of the world size {self.world}")

    def frames(self, F: int):
        n = F // self.world
        return self.rank * n, (self.rank + 1) * n

    # ---- collectives
    def _a2a(self, send: torch.Tensor) -> torch.Tensor:
        self.bytes_moved += send.numel() * send.element_size() * (self.world - 1) // self.world
        if self.host_staged and send.is_cuda:
            s = send.cpu()
            r = torch.empty_like(s)
            self.dist.all_to_all_single(r, s, group=self.group)
            return r.to(send.device)
        recv = torch.empty_like(send)
        self.dist.all_to_all_single(recv, send, group=self.group)
        return recv

    def all_reduce_sum(self, t: torch.Tensor):
        if self.host_staged and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, group=self.group)
        return t

    def frames_to_pixels(self, x: torch.Tensor, B: int, Fl: int, HW: int) -> torch.Tensor:
        """[(B Fl)(HW), C] (my frames, all pixels) -> [(B F)(HW/world), C] (all frames, my pixels)."""
        N, C = self.world, x.shape[1]
        send = x.view(B, Fl, N, HW // N, C).permute(2, 0, 1, 3, 4).contiguous()      # [dst, B, Fl, HWl, C]
        recv = self._a2a(send)                                                         # [src = frame block, ...]
        return recv.permute(1, 0, 2, 3, 4).reshape(B * N * Fl * (HW // N), C)          # frames of rank s sit at s*Fl..

    def pixels_to_frames(self, y: torch.Tensor, B: int, Fl: int, HW: int) -> torch.Tensor:
        N, C = self.world, y.shape[1]
        send = y.view(B, N, Fl, HW // N, C).permute(1, 0, 2, 3, 4).contiguous()       # [dst = frame block, B, Fl, HWl, C]
        recv = self._a2a(send)                                                         # [src = pixel block, ...]
        return recv.permute(1, 2, 0, 3, 4).reshape(B * Fl * HW, C)

    def gather_frames(self, v: torch.Tensor, B: int, Fl: int, HW: int) -> torch.Tensor:
        """[(B Fl)(HW), C] per rank -> the full [(B F)(HW), C] on every rank (the 8-channel v-prediction)."""
        N, C = self.world, v.shape[1]
        if self.host_staged and v.is_cuda:
            parts = [torch.empty(v.shape, dtype=v.dtype) for _ in range(N)]
            self.dist.all_gather(parts, v.cpu(), group=self.group)
            parts = [p.to(v.device) for p in parts]
        else:
            parts = [torch.empty_like(v) for _ in range(N)]
            self.dist.all_gather(parts, v.contiguous(), group=self.group)
        full = torch.stack([p.view(B, Fl, HW, C) for p in parts], 1)                  # [B, N, Fl, HW, C]
        return full.reshape(B * N * Fl * HW, C)

    def gather_video(self, video: torch.Tensor) -> torch.Tensor:
        """Decoded frames [1,3,F/world,H,W] (fp32, host or device, as ``decode_video`` returns them) -> the whole clip on every rank."""
        if self.host_staged and video.is_cuda:  # gloo bring-up: stage device frames through the host
            return self.gather_video(video.cpu()).to(video.device)
        parts = [torch.empty_like(video) for _ in range(self.world)]
        if self.host_staged or not video.is_cuda:
            if self.host_staged:
                self.dist.all_gather(parts, video.contiguous(), group=self.group)
            else:  # RCCL moves device buffers: stage the (host) frames through HBM
                dev = torch.device("cuda", torch.cuda.current_device())
                dparts = [torch.empty(video.shape, dtype=video.dtype, device=dev) for _ in range(self.world)]
                self.dist.all_gather(dparts, video.to(dev).contiguous(), group=self.group)
                parts = [p.cpu() for p in dparts]
        else:
            self.dist.all_gather(parts, video.contiguous(), group=self.group)
        return torch.cat(parts, 2)

    # ---- the wrapper the temporal layers call
    def temporal(self, ctx, x: torch.Tensor, HW: int, body):
        """``body(ctx_t, x_pixels, HW_local, shard)`` runs the layer on the pixel-sharded tokens of ALL frames."""
        tctx = ctx.__dict__.get("_temporal_view")
        if tctx is None:
            import copy
            tctx = copy.copy(ctx)
            tctx.F = ctx.F * self.world
            ctx._temporal_view = tctx
        xp = self.frames_to_pixels(x, ctx.B, ctx.F, HW)
        y = body(tctx, xp, HW // self.world, (self.world, self.all_reduce_sum))
        return self.pixels_to_frames(y, ctx.B, ctx.F, HW)

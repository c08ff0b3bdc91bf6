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


def seed_for_entry(base_seed: int, entry_index: int) -> int:
    """The reference seeds once per process and draws per entry from the global RNG (``run_group_pnp_edit.py:124,213``),
    which ties results to the processing order.  Sharded runs re-seed per entry so that any rank reproduces the
    same numbers for the same entry regardless of world size."""
    return int(base_seed) + 1000003 * int(entry_index)


def gather_latents(latents: Optional[torch.Tensor], like_shape, dtype, device) -> List[torch.Tensor]:
    """all_gather of one ``[1,4,F,h,w]`` latent per rank (512 KiB at 16f x 512^2).  Ranks that had no entry
    contribute zeros."""
    import torch.distributed as dist
    world = dist.get_world_size()
    x = latents if latents is not None else torch.zeros(like_shape, dtype=dtype, device=device)
    out = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(out, x.contiguous())
    return out

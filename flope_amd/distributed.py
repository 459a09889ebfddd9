"""Data-parallel driver: one process per GPU, crops sharded by contiguous blocks, no
collective on the data path, ONE all-gather of the finished poses (RCCL over xGMI when the
backend is "nccl"; "gloo" on CPU for the world_size-2 tests).

The reference has no multi-GPU code at all: every crop is independent in
eval mode, so the path shards embarrassingly; the pose record is 16 float32 = row-major 4x4 Rt.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

POSE_FLOATS = 16


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block partition of [0, n_items): rank r gets [lo, hi); sizes differ by <= 1."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_from_env(backend: str | None = None):
    """Join the job torch.distributed.run started (RANK / WORLD_SIZE / MASTER_* in the env).
    -> (rank, world, local_rank).  Single-process when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_poses(local: torch.Tensor, counts=None) -> torch.Tensor:
    """All-gather [n_local, 16] pose records into [sum n, 16] on every rank, in rank order.
    Equal shard sizes take the single all_gather_into_tensor fast path; ragged shards are
    padded to the largest and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    local = local.contiguous()
    if counts is None:
        counts = [local.shape[0]] * world
    width = tuple(local.shape[1:])
    if len(set(counts)) == 1:
        out = torch.empty((world * counts[0],) + width, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out
    m = max(counts)
    padded = torch.zeros((m,) + width, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * m,) + width, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * m: r * m + c] for r, c in enumerate(counts)], dim=0)


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()

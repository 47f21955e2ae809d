"""Multi-GPU plumbing for inference: molecules are independent, so a
batch shards by contiguous molecule ranges with NO data-path collective.  The only
communication is the benchmark's barrier + timing reduction (RCCL on GPUs, gloo in
the CPU tests)."""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_by_edges(ptr: Sequence[int], world_size: int, cost: Sequence[float] = None) -> List[Tuple[int, int]]:
    """Contiguous molecule ranges [g0, g1) per rank, balanced on estimated edge count.

    ``cost[g]`` defaults to n_g * (n_g - 1) (all pairs inside the cutoff, the QM9
    regime); any per-molecule edge estimate can be passed instead.  Every molecule
    is assigned to exactly one rank; ranks may be empty when there are fewer
    molecules than ranks."""
    ptr = np.asarray(ptr, dtype=np.int64)
    n_mol = len(ptr) - 1
    if cost is None:
        n = np.diff(ptr).astype(np.float64)
        cost = n * (n - 1)
    cost = np.asarray(cost, dtype=np.float64)
    csum = np.concatenate([[0.0], np.cumsum(cost)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        g = int(np.searchsorted(csum, target, side="left"))
        # choose the closer of the two neighbouring cut points, keep cuts monotone
        if g > 0 and abs(csum[g - 1] - target) <= abs(csum[min(g, n_mol)] - target):
            g -= 1
        cuts.append(min(max(g, cuts[-1]), n_mol))
    cuts.append(n_mol)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def take_shard(pos: np.ndarray, z: np.ndarray, ptr: np.ndarray, g0: int, g1: int):
    """Slice molecules [g0, g1) out of a batch (re-based ptr)."""
    a, b = int(ptr[g0]), int(ptr[g1])
    return pos[a:b], z[a:b], (np.asarray(ptr[g0 : g1 + 1]) - ptr[g0]).astype(np.int64)


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from torchrun's environment; initialises the
    process group when WORLD_SIZE > 1 (backend "nccl" = RCCL on ROCm)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def reduce_timing(local_seconds: float, local_units: float, device="cpu") -> Tuple[float, float]:
    """(max over ranks of the timed interval, sum over ranks of the units processed)."""
    if not dist.is_initialized():
        return float(local_seconds), float(local_units)
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    u = torch.tensor([local_units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())

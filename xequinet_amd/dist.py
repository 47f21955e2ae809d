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


def plan_chunks(ptr: Sequence[int], max_edges: int, g0: int = 0, g1: int = None) -> List[Tuple[int, int]]:
    """Cut molecules [g0, g1) into contiguous ranges whose edge count cannot exceed ``max_edges``.

    The bound per molecule is n_g (n_g - 1) (every ordered pair inside the cutoff), so a range never overflows the
    32-bit byte offsets of the matrix-core message kernels (xeq_message_wq_fits) whatever the geometry.  A single
    molecule above the cap gets a range of its own (the kernels' own size check then decides)."""
    ptr = np.asarray(ptr, dtype=np.int64)
    g1 = len(ptr) - 1 if g1 is None else g1
    n = np.diff(ptr[g0 : g1 + 1]).astype(np.int64)
    bound = n * (n - 1)

    csum = np.concatenate([[0], np.cumsum(bound)])

    def cut(cap: int) -> List[Tuple[int, int]]:
        # greedy: a range takes molecules while their bounds fit under the cap (at least one); one searchsorted per range --
        # this runs per evaluation on 65 k molecules (runtime.evaluate_in_chunks), a Python loop per molecule cost 90 ms there
        out, start, n_mol = [], 0, len(bound)
        while start < n_mol:
            end = int(np.searchsorted(csum, csum[start] + cap, side="right")) - 1
            end = min(max(end, start + 1), n_mol)
            out.append((g0 + start, g0 + end))
            start = end
        return out or [(g0, g1)]

    # the fewest ranges the cap allows, then the smallest cap that still gives that many: ranges of about equal size instead
    # of full ones and a small remainder (a remainder of a few thousand edges would also take another kernel family than
    # its siblings, ops.prefers_sb, and with it other bits)
    ranges = cut(int(max_edges))
    k, total = len(ranges), int(bound.sum())
    if k > 1:
        lo, hi = max(int(bound.max()), -(-total // k)), int(max_edges)
        if lo > hi:          # a molecule above the cap: the first cut stands
            return ranges
        while lo < hi:
            mid = (lo + hi) // 2
            if len(cut(mid)) <= k:
                hi = mid
            else:
                lo = mid + 1
        ranges = cut(lo)
    return ranges


def gather_shards(local: dict, dst: int = 0):
    """Concatenate per-rank result dicts of numpy arrays in rank order on ``dst`` (None elsewhere).  The ONLY exchange
    of a sharded inference: after the evaluation, off the data path (results concatenated on the host)."""
    if not dist.is_initialized():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    box = [None] * world if rank == dst else None
    dist.gather_object(local, box, dst=dst)
    if rank != dst:
        return None
    return {k: np.concatenate([b[k] for b in box], axis=0) for k in box[0]}


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

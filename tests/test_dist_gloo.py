"""N > 1 path on CPU: molecule sharding and the benchmark's timing reduction over gloo (world_size 2)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from oracle import xpainn_oracle as orc
from xequinet_amd import dist as xdist


def test_shard_by_edges_partitions_and_balances():
    pos, z, ptr = orc.synth_qm9_batch(200, seed=3)
    n = np.diff(ptr)
    cost = n * (n - 1)
    for world in (1, 2, 3, 8, 16):
        shards = xdist.shard_by_edges(ptr, world)
        assert shards[0][0] == 0 and shards[-1][1] == 200
        assert all(a[1] == b[0] for a, b in zip(shards, shards[1:]))          # contiguous, exact cover
        loads = [cost[g0:g1].sum() for g0, g1 in shards]
        assert max(loads) <= cost.sum() / world + cost.max()                   # within one molecule of ideal
    # fewer molecules than ranks: empty shards allowed, nothing lost
    shards = xdist.shard_by_edges(ptr[:4], 8)
    assert sum(g1 - g0 for g0, g1 in shards) == 3
    # slicing re-bases ptr
    p, zz, pp = xdist.take_shard(pos, z, ptr, 5, 9)
    assert pp[0] == 0 and pp[-1] == len(p) == ptr[9] - ptr[5] and len(zz) == len(p)


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = xdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    pos, z, ptr = orc.synth_qm9_batch(40, seed=1)
    g0, g1 = xdist.shard_by_edges(ptr, world)[rank]
    p, zz, pp = xdist.take_shard(pos, z, ptr, g0, g1)
    n = np.diff(pp)
    edges = float((n * (n - 1)).sum())
    xdist.barrier()
    t, total = xdist.reduce_timing(0.1 * (rank + 1), edges)
    out[rank] = (t, total, len(p))
    torch.distributed.destroy_process_group()


def test_gloo_world2_timing_reduction():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    pos, z, ptr = orc.synth_qm9_batch(40, seed=1)
    n = np.diff(ptr)
    assert abs(out[0][0] - 0.2) < 1e-12 and abs(out[1][0] - 0.2) < 1e-12        # MAX over ranks
    assert out[0][1] == out[1][1] == float((n * (n - 1)).sum())                  # SUM over ranks: nothing lost
    assert out[0][2] + out[1][2] == len(pos)

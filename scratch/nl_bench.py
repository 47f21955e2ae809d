"""Neighbour-list time on periodic water boxes: pair sweep (image-pruned) vs cell list."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd.data import radius_graph_pbc
dev = "cuda"
for k in (8, 11, 14, 16, 24):
    pos, z, ptr, cell = syn.synth_water_box(k, seed=5)
    p = torch.tensor(pos, dtype=torch.float32, device=dev); c = torch.tensor(cell, dtype=torch.float32, device=dev).reshape(1, 3, 3)
    n = torch.tensor([len(pos)], device=dev); pbc = torch.tensor([[True, True, True]], device=dev)
    res = {}
    for flag in ("0", "1"):
        os.environ["XEQ_PBC_CELL_LIST"] = flag
        for _ in range(2): ei, co = radius_graph_pbc(p, n, pbc, c, 5.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ei, co = radius_graph_pbc(p, n, pbc, c, 5.0)
        torch.cuda.synchronize(); res[flag] = ((time.perf_counter() - t0) / 5 * 1e3, ei)
    same = torch.equal(res["0"][1], res["1"][1])
    print(f"{len(pos)} atoms, {res['0'][1].shape[1]} edges: pair sweep {res['0'][0]:.2f} ms, cell list {res['1'][0]:.2f} ms, identical: {same}")

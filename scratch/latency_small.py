"""Latency of one energy+force evaluation of small systems: eager vs HIP-graph replay (neighbour list included)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd.data import NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
from xequinet_amd.runtime import GraphedModel
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(dev)
gm = GraphedModel(model)
tr = NeighborTransform(5.0)
def bench(name, pos, z, ptr, **kw):
    p, zz, pp = torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)
    def eager():
        d = tr(XequiBatch(p, zz, pp, **kw)).to_dict()
        with torch.enable_grad():
            return model(d, compute_forces=True)["forces"]
    def graphed():
        return gm(tr(XequiBatch(p, zz, pp, **kw)).to_dict())["forces"]
    res = {}
    for nm, fn in (("eager", eager), ("graph", graphed)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); res[nm] = (time.perf_counter() - t0) / 50 * 1e3
    print(f"{name}: eager {res['eager']:.3f} ms, HIP-graph replay {res['graph']:.3f} ms per evaluation (neighbour list included)")
pos, z, ptr = syn.synth_aspirin()
bench("aspirin (21 atoms)", pos, z, ptr)
pos, z, ptr = syn.synth_qm9_batch(16, seed=3)
bench("16 QM9-shape molecules", pos, z, ptr)
pos, z, ptr, cell = syn.synth_water_box(4, seed=5)
bench("water-64 box (192 atoms, PBC)", pos, z, ptr, pbc=torch.tensor([[True, True, True]], device=dev), cell=torch.tensor(cell, dtype=torch.float32, device=dev))

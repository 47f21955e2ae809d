"""Probe: neighbour list of step k+1 on a side stream while the HIP graph of step k runs."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd import keys
from xequinet_amd.data import NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
from xequinet_amd.runtime import GraphedModel
from xequinet_amd.tuning import enable_gemm_autotune
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(dev)
enable_gemm_autotune()
pos, z, ptr = syn.synth_qm9_batch(1024, seed=1234)
pos_d, z_d, ptr_d = torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)
tr = NeighborTransform(5.0)
gm = GraphedModel(model, tune_gemms=False)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
def seq_step():
    return gm(tr(XequiBatch(pos_d, z_d, ptr_d)).to_dict())
def prepare():
    with torch.cuda.stream(side):
        b = tr(XequiBatch(pos_d, z_d, ptr_d))
        d = b.to_dict()
        ev = torch.cuda.Event(); ev.record(side)
    g = d[keys.EDGE_GRAPH]
    for t in (d["edge_index"], d["batch"], g.c_rowptr, g.n_rowptr, g.n_perm):
        t.record_stream(main)
    return d, ev
for _ in range(5): ref = {k: v.clone() for k, v in seq_step().items()}
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for _ in range(K): out = seq_step()
torch.cuda.synchronize(); t1 = time.perf_counter()
side.wait_stream(main)
d, ev = prepare()
for k in range(K):
    main.wait_event(ev)
    out2 = gm(d)
    if k + 1 < K: d, ev = prepare()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"sequential {1e3*(t1-t0)/K:.3f} ms/step, pipelined {1e3*(t2-t1)/K:.3f} ms/step; same forces: {torch.equal(ref['forces'], out2['forces'])}")

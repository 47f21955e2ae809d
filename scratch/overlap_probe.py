"""Probe: neighbour list of batch k+1 on a side stream while the model runs batch k."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd import keys
from xequinet_amd.data import NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
from xequinet_amd.tuning import enable_gemm_autotune
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(dev)
pos, z, ptr = syn.synth_qm9_batch(1024, seed=1234)
pos_d, z_d, ptr_d = torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)
tr = NeighborTransform(5.0)
enable_gemm_autotune()
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
def build():
    side.wait_stream(main) if False else None
    with torch.cuda.stream(side):
        b = tr(XequiBatch(pos_d.detach(), z_d, ptr_d))
        ev = torch.cuda.Event()
        ev.record(side)
    g = getattr(b, keys.EDGE_GRAPH)
    for t in (b.edge_index, b.batch, g.c_rowptr, g.n_rowptr, g.n_perm):
        t.record_stream(main)
    return b, ev
def run(b, ev):
    main.wait_event(ev)
    with torch.enable_grad():
        return model(b.to_dict(), compute_forces=True, compute_virial=False)
def eager():
    b = tr(XequiBatch(pos_d.detach(), z_d, ptr_d))
    with torch.enable_grad():
        return model(b.to_dict(), compute_forces=True, compute_virial=False)
for _ in range(5): ref = eager()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): out = eager()
torch.cuda.synchronize(); t1 = time.perf_counter()
nxt = build()
for k in range(K):
    cur = nxt
    out2 = run(*cur)
    if k + 1 < K: nxt = build()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"eager {1e3*(t1-t0)/K:.3f} ms/step, pipelined {1e3*(t2-t1)/K:.3f} ms/step; same forces: {torch.equal(out['forces'], out2['forces'])}")

"""Probe: can one evaluation (model forward + force backward) be captured in a HIP graph and replayed?"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd import ops
from xequinet_amd.data import NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
from xequinet_amd.tuning import enable_gemm_autotune
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(dev)
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pos, z, ptr = syn.synth_qm9_batch(nm, seed=1234)
pos_d, z_d, ptr_d = torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)
tr = NeighborTransform(5.0)
if os.path.exists("gpurun_out/gemm_r01_h.csv"):
    enable_gemm_autotune(results_file="gpurun_out/gemm_r01_h.csv")
b = tr(XequiBatch(pos_d.clone(), z_d, ptr_d))
data0 = b.to_dict()
def fwd(data):
    d = dict(data)
    d["pos"] = d["pos"].detach()
    with torch.enable_grad():
        out = model(d, compute_forces=True, compute_virial=False)
    return out["energy"], out["forces"]
for _ in range(3):
    e_ref, f_ref = fwd(data0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        fwd(data0)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    e_g, f_g = fwd(data0)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("replay vs eager: dE", float((e_g - e_ref).abs().max()), "dF", float((f_g - f_ref).abs().max()))
# move the atoms a little (same graph topology) and replay
data0["pos"].add_(0.001)
g.replay(); torch.cuda.synchronize()
e2, f2 = fwd(data0)
print("after moving atoms: dE", float((e_g - e2).abs().max()), "dF", float((f_g - f2).abs().max()))
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(20): fwd(data0)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"molecules {nm}: graph replay {1e3*(t1-t0)/20:.3f} ms, eager model {1e3*(t2-t1)/20:.3f} ms (neighbour list excluded in both)")

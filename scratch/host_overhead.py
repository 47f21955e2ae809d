"""How long the host takes to ENQUEUE one evaluation vs how long the GPU takes to run it."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd.data import NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
from xequinet_amd.tuning import enable_gemm_autotune
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(dev)
pos, z, ptr = syn.synth_qm9_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 1024, seed=1234)
pos_d, z_d, ptr_d = torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)
tr = NeighborTransform(5.0)
if os.path.exists("gpurun_out/gemm_r01_h.csv"):
    enable_gemm_autotune(results_file="gpurun_out/gemm_r01_h.csv")
def step():
    b = tr(XequiBatch(pos_d.detach(), z_d, ptr_d))
    with torch.enable_grad():
        return model(b.to_dict(), compute_forces=True, compute_virial=False)
for _ in range(5): step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): out = step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"impl={os.environ.get('XEQ_MESSAGE_IMPL','auto')} host enqueue {1e3*(t1-t0)/K:.2f} ms/step, wall {1e3*(t2-t0)/K:.2f} ms/step (GPU drains {1e3*(t2-t1):.2f} ms after the last enqueue)")

"""Which torch ops launch the glue kernels of one evaluation (torch.profiler, grouped by op)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd.data import NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(dev)
pos, z, ptr = syn.synth_qm9_batch(1024, seed=1234)
pos_d, z_d, ptr_d = torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)
tr = NeighborTransform(5.0)
def step():
    b = tr(XequiBatch(pos_d.detach(), z_d, ptr_d))
    with torch.enable_grad():
        return model(b.to_dict(), compute_forces=True, compute_virial=False)
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    print(f"{e.count:3d} x {e.key:34s} {e.device_time_total:8.1f} us  {str(e.input_shapes)[:90]}")

import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import runtime
from xequinet_amd.data import synthetic as syn, XequiBatch
from xequinet_amd.nn import resolve_model
dev = "cuda"
torch.manual_seed(0)
model = resolve_model("xpainn").to(dev).eval().requires_grad_(False)
p, z, ptr, _ = syn.make_workload("qm9_1024", seed=1234)
b = XequiBatch(torch.tensor(p, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev))
pad = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
g = runtime.GraphedStep(model, (b.pos.shape[0] + pad, len(ptr) - 1, runtime.pair_capacity(ptr)), compute_forces=True)
for i in range(30):
    g(b.pos, b.atomic_numbers, b.ptr, batch=b.batch)
torch.cuda.synchronize()

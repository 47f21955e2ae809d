import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xequinet_amd import keys, train
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model, training as tr
tr.NATIVE_LINEAR = os.environ.get("XEQ_NATIVE_LINEAR") == "1"
dev = "cuda"
pos, z, ptr = syn.synth_qm9_batch(128, seed=1234)
b = NeighborTransform(5.0)(XequiBatch(torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)))
data = b.to_dict()
torch.manual_seed(0)
model = resolve_model("xpainn").to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(0)
tgt = {keys.TOTAL_ENERGY: torch.randn(128, generator=g).to(dev), keys.FORCES: torch.randn(len(pos), 3, generator=g).to(dev), keys.BATCH_PTR: data["ptr"]}
w = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 10.0}
def step():
    d = {k: v for k, v in data.items() if not k.startswith("_")}
    d["pos"] = d["pos"].detach().clone()
    return train.train_step(model, d, tgt, opt, w)[0]
for _ in range(5): step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(5): step()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=50))

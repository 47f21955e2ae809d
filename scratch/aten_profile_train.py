import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import keys, train
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
dev = "cuda"
n_mol = 1024
pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=1234)
b = NeighborTransform(5.0)(XequiBatch(torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)))
data = b.to_dict()
torch.manual_seed(0)
model = resolve_model("xpainn").to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(0)
tgt = {keys.TOTAL_ENERGY: torch.randn(n_mol, generator=g).to(dev), keys.FORCES: torch.randn(len(pos), 3, generator=g).to(dev), keys.BATCH_PTR: data["ptr"]}
w = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 10.0}
def step():
    d = {k: v for k, v in data.items() if not k.startswith("_")}
    d["pos"] = d["pos"].detach().clone()
    return train.train_step(model, d, tgt, opt, w)[0]
for _ in range(12): step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None: dt = getattr(e, "self_cuda_time_total", 0)
    if dt > 0: rows.append((dt / 3, e.count / 3, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"device time per step by operator and shapes: {tot / 1e3:.2f} ms")
for dt, cnt, key, shp in rows[:40]:
    print(f"{dt:9.1f} us {cnt:6.1f} x  {key[:42]:42s} {shp}")

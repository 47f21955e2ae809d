import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import keys, train
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model, training as tr, training_ops as tops
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
def timeit(label):
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); print(f"{label}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per step")
orig_mlp, orig_linear = tr._mlp, tops.linear
tr.NATIVE_LINEAR = False; timeit("torch.nn everywhere")
tr.NATIVE_LINEAR = True; timeit("xeq::linear everywhere")
tr._mlp = lambda seq, x: seq(x); timeit("xeq::linear for the o3 pairs only")
tr._mlp = orig_mlp
tops.linear = lambda x, W, b: torch.nn.functional.linear(x, W, b)
import xequinet_amd.nn.training_ops as _t
timeit("xeq::linear replaced by F.linear inside the same call sites (graph shape of the variant, library products)")
tops.linear = orig_linear
# the wgrad kernel alone
a = torch.randn(2352, 576, device=dev); bb = torch.randn(2352, 128, device=dev)
from xequinet_amd.nn.fused import _wgrad
for _ in range(5): _wgrad(a, bb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): _wgrad(a, bb)
torch.cuda.synchronize(); print(f"_wgrad [2352 x 576]^T [2352 x 128]: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call (host + device)")

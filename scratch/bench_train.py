"""Training-step time on one GPU: QM9-shaped batch, Adam, l2 loss.  An energy loss takes the native pass (fused kernels with parameter
gradients, nn/fused.py); `energy-aten` forces the differentiable form of the same step (nn/training.py), as `forces` needs anyway.
usage: python scratch/bench_train.py [n_mol] [energy|energy-graph|energy-aten|forces]"""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import keys, train
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model

n_mol = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else "forces"
dev = "cuda"
pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=1234)
b = NeighborTransform(5.0)(XequiBatch(torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)))
data = b.to_dict()
E = data["edge_index"].shape[1]
torch.manual_seed(0)
model = resolve_model("xpainn").to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(0)
tgt = {keys.TOTAL_ENERGY: torch.randn(n_mol, generator=g).to(dev), keys.FORCES: torch.randn(len(pos), 3, generator=g).to(dev), keys.BATCH_PTR: data["ptr"]}
w = {keys.TOTAL_ENERGY: 1.0} if mode.startswith("energy") else {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 10.0}
model.native_training = mode != "energy-aten"
if os.environ.get("XEQ_NATIVE_LINEAR"):
    from xequinet_amd.nn import training as _tr
    _tr.NATIVE_LINEAR = os.environ["XEQ_NATIVE_LINEAR"] == "1"
if mode == "forces-graph":   # energy + forces, the twice-differentiable pass inside ONE captured graph
    from xequinet_amd import runtime
    from xequinet_amd.nn import training as tr
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
    gstep = train.GraphedTrainStep(model, opt, (len(pos) + 64, n_mol, runtime.pair_capacity(ptr)), forces_weight=10.0)
    p_d, z_d, ptr_d, b_d = data["pos"].detach(), data["atomic_numbers"], data["ptr"], data["batch"]
    for _ in range(3): l = gstep(p_d, z_d, ptr_d, tgt[keys.TOTAL_ENERGY], batch=b_d, target_forces=tgt[keys.FORCES])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): l = gstep(p_d, z_d, ptr_d, tgt[keys.TOTAL_ENERGY], batch=b_d, target_forces=tgt[keys.FORCES])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"train step (energy + forces, ONE captured graph incl. the neighbour list, linear layers LinearFn inside the capture) n_mol={n_mol} N={len(pos)} E={E}: {dt*1e3:.2f} ms/step, {E/dt/1e6:.1f} M edges/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB, loss {l.item():.4f}")
    sys.exit(0)
if mode == "energy-graph":   # the whole step (neighbour list included) as ONE captured HIP graph: train.GraphedTrainStep
    from xequinet_amd import runtime
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
    gstep = train.GraphedTrainStep(model, opt, (len(pos) + 64, n_mol, runtime.pair_capacity(ptr)))
    p_d, z_d, ptr_d, b_d = data["pos"].detach(), data["atomic_numbers"], data["ptr"], data["batch"]
    for _ in range(3): l = gstep(p_d, z_d, ptr_d, tgt[keys.TOTAL_ENERGY], batch=b_d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): l = gstep(p_d, z_d, ptr_d, tgt[keys.TOTAL_ENERGY], batch=b_d)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"train step (energy loss, ONE captured graph incl. the neighbour list) n_mol={n_mol} N={len(pos)} E={E}: {dt*1e3:.2f} ms/step, {E/dt/1e6:.1f} M edges/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB, loss {l.item():.4f}")
    sys.exit(0)
def step():
    d = {k: v for k, v in data.items() if not k.startswith("_")}
    d["pos"] = d["pos"].detach().clone()
    return train.train_step(model, d, tgt, opt, w)[0]
for _ in range(int(os.environ.get("XEQ_WARMUP", "15"))): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 10
for _ in range(K): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"train step ({mode} loss) n_mol={n_mol} N={len(pos)} E={E}: {dt*1e3:.2f} ms/step, {E/dt/1e6:.1f} M edges/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB, loss {l.item():.4f}")
from xequinet_amd import ops
ops.KERNEL_TIMER.reset(enabled=True)
for _ in range(5): step()
print({k: f"{v['total_ms'] / v['launches'] * 1e3:.0f} us x {v['launches'] / 5:.0f}" for k, v in ops.KERNEL_TIMER.summary().items()})

"""One small open-boundary system through the whole-step graph (runtime.GraphedStep), 30 replays: python scratch/md_step.py [n_molecules] [aspirin|qm9]
(used under rocprofv3 --kernel-trace by scratch/exp14.sh: where the time of an MD-sized step goes)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import runtime
from xequinet_amd.data import synthetic as syn
from xequinet_amd.nn import resolve_model
dev = torch.device("cuda", 0)
n_mol = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kind = sys.argv[2] if len(sys.argv) > 2 else "aspirin"
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(dev)
if kind == "aspirin":
    p0, z0, _ = syn.synth_aspirin()
    pos = np.concatenate([p0 + 50.0 * i for i in range(n_mol)]); z = np.tile(z0, n_mol); ptr = np.arange(0, (n_mol + 1) * len(z0), len(z0), dtype=np.int64)
else:
    pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=3)
p = torch.tensor(pos, dtype=torch.float32, device=dev); zz = torch.tensor(z, device=dev); pp = torch.tensor(ptr, device=dev)
step = runtime.GraphedStep(model, (len(pos) + 8, n_mol, runtime.pair_capacity(ptr)), compute_forces=True)
for _ in range(5): out = step(p, zz, pp)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): out = step(p, zz, pp)
torch.cuda.synchronize()
print(f"{kind} x {n_mol}: N = {len(pos)}, E = {int(out['n_edges'])}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per whole-step replay")

"""Host-side profile of one GROMACS-style whole-step call (XPaiNNGMX(replay=True, whole_step=True)) on a 192-atom water box."""
import os, sys, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.data import synthetic as syn
from xequinet_amd.interface import XPaiNNGMX
from xequinet_amd.utils import set_default_units
dev = torch.device("cuda", 0)
set_default_units({"energy": "eV"})
torch.manual_seed(0)
g = XPaiNNGMX(replay=True, whole_step=True).eval().requires_grad_(False).to(dev)
pos, z, ptr, cell = syn.synth_water_box(4, seed=5)
p = torch.tensor(pos, dtype=torch.float32, device=dev); zz = torch.tensor(z, device=dev)
c = torch.tensor(cell[0], dtype=torch.float32, device=dev); pbc = torch.tensor([True, True, True], device=dev)
def gstep():
    x = (p / 10).requires_grad_(True)
    e = g(x, zz, c / 10, pbc)
    return torch.autograd.grad(e.sum(), x)[0]
for _ in range(10): gstep()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): gstep()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) * 10:.3f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(100): gstep()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

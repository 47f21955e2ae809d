"""The LAMMPS-style front (neighbour list given) replaying one system 40 times: python scratch/md_lmp.py aspirin|water64|water512 [gmx]
(under rocprofv3 --kernel-trace by scratch/exp27.sh: per-kernel time of an MD-sized step)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.data import synthetic as syn
from xequinet_amd.data import single_radius_graph
from xequinet_amd.cluster import radius_graph
from xequinet_amd.interface import XPaiNNGMX, XPaiNNLMP
from xequinet_amd.utils import set_default_units
dev = torch.device("cuda", 0)
set_default_units({"energy": "eV"})
name = sys.argv[1] if len(sys.argv) > 1 else "aspirin"
front = sys.argv[2] if len(sys.argv) > 2 else "lmp"
if name == "aspirin":
    pos, z, ptr = syn.synth_aspirin(); cell = None
else:
    pos, z, ptr, cell = syn.synth_water_box(4 if name == "water64" else 8, seed=5)
p = torch.tensor(pos, dtype=torch.float32, device=dev); zz = torch.tensor(z, device=dev)
c = None if cell is None else torch.tensor(cell[0], dtype=torch.float32, device=dev)
pbc = None if cell is None else torch.tensor([True, True, True], device=dev)
torch.manual_seed(0)
if front == "lmp":
    if cell is None:
        ei = radius_graph(p, 5.0, ptr=torch.tensor([0, len(z)], device=dev)); extra = {}
    else:
        ei, co = single_radius_graph(p, pbc, c, 5.0); extra = {"cell": c[None], "cell_offsets": co, "pbc": pbc[None]}
    m = XPaiNNLMP(unit_style="metal", replay=True).eval().requires_grad_(False).to(dev)
    def step():
        with torch.enable_grad():
            return m({"pos": p, "atomic_numbers": zz, "edge_index": ei, **extra}, True, False)["forces"]
else:
    g = XPaiNNGMX(replay=True, whole_step=True).eval().requires_grad_(False).to(dev)
    def step():
        x = (p / 10).requires_grad_(True)
        e = g(x, zz, None if c is None else c / 10, pbc)
        return torch.autograd.grad(e.sum(), x)[0]
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 40
for _ in range(N): step()
torch.cuda.synchronize()
print(f"{name} {front}: {(time.perf_counter() - t0) / N * 1e3:.3f} ms per step", flush=True)

"""LAMMPS-style replay of one MD-sized system, for rocprofv3: python scratch/prof_md.py [aspirin|water64|water512] [steps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.data import synthetic as syn, single_radius_graph
from xequinet_amd.cluster import radius_graph
from xequinet_amd.interface import XPaiNNLMP
from xequinet_amd.utils import set_default_units
dev = torch.device("cuda", 0)
set_default_units({"energy": "eV"})
which = sys.argv[1] if len(sys.argv) > 1 else "aspirin"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if which == "aspirin":
    pos, z, ptr = syn.synth_aspirin(); cell = None
else:
    pos, z, ptr, cell = syn.synth_water_box(4 if which == "water64" else 8, seed=5)
p = torch.tensor(pos, dtype=torch.float32, device=dev); zz = torch.tensor(z, device=dev)
if cell is None:
    ei = radius_graph(p, 5.0, ptr=torch.tensor([0, len(z)], device=dev)); extra = {}
else:
    c = torch.tensor(cell[0], dtype=torch.float32, device=dev); pbc = torch.tensor([True, True, True], device=dev)
    ei, co = single_radius_graph(p, pbc, c, 5.0); extra = {"cell": c[None], "cell_offsets": co, "pbc": pbc[None]}
torch.manual_seed(0)
m = XPaiNNLMP(unit_style="metal", replay=True).eval().requires_grad_(False).to(dev)
for _ in range(steps):
    with torch.enable_grad():
        f = m({"pos": p, "atomic_numbers": zz, "edge_index": ei, **extra}, True, False)["forces"]
torch.cuda.synchronize()
print(which, "steps", steps)

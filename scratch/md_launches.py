"""Kernel list of one replayed LAMMPS-style step on aspirin: run under rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.data import synthetic as syn
from xequinet_amd.cluster import radius_graph
from xequinet_amd.interface import XPaiNNLMP
from xequinet_amd.utils import set_default_units
dev = torch.device("cuda", 0)
set_default_units({"energy": "eV"})
torch.manual_seed(0)
m = XPaiNNLMP(unit_style="metal", replay=True).eval().requires_grad_(False).to(dev)
pos, z, ptr = syn.synth_aspirin()
p = torch.tensor(pos, dtype=torch.float32, device=dev); zz = torch.tensor(z, device=dev)
ei = radius_graph(p, 5.0, ptr=torch.tensor([0, len(z)], device=dev))
N = int(os.environ.get("STEPS", "200"))
for _ in range(N):
    with torch.enable_grad():
        f = m({"pos": p, "atomic_numbers": zz, "edge_index": ei}, True, False)["forces"]
torch.cuda.synchronize()
print("done")

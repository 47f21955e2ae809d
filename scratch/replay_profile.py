"""Replay one small periodic system many times (for rocprofv3 --kernel-trace --stats)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd.data import single_radius_graph
from xequinet_amd.interface import XPaiNNLMP
from xequinet_amd.utils import set_default_units
dev = torch.device("cuda", 0)
set_default_units({"energy": "eV"})
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if len(sys.argv) > 2:   # file of library-GEMM picks: written by a first (un-profiled) run, replayed under the profiler
    from xequinet_amd.tuning import enable_gemm_autotune
    enable_gemm_autotune(results_file=sys.argv[2])
torch.manual_seed(0)
m = XPaiNNLMP(unit_style="metal", replay=True).eval().requires_grad_(False).to(dev)
pos, z, ptr, cell = syn.synth_water_box(n, seed=5)
p = torch.tensor(pos, dtype=torch.float32, device=dev); zz = torch.tensor(z, device=dev)
c = torch.tensor(cell[0], dtype=torch.float32, device=dev); pbc = torch.tensor([True, True, True], device=dev)
ei, co = single_radius_graph(p, pbc, c, 5.0)
d = {"pos": p, "atomic_numbers": zz, "edge_index": ei, "cell": c[None], "cell_offsets": co, "pbc": pbc[None]}
for _ in range(100):
    m(dict(d), True, False)
torch.cuda.synchronize()
print("done", ei.shape[1])

"""LAMMPS-style replay latency with pieces switched: message impl, fused node kernels on/off."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.data import synthetic as syn, single_radius_graph
from xequinet_amd.cluster import radius_graph
from xequinet_amd.interface import XPaiNNLMP
from xequinet_amd.nn import fused
from xequinet_amd.utils import set_default_units
dev = torch.device("cuda", 0)
set_default_units({"energy": "eV"})
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def run(name, pos, z, cell=None):
    p = torch.tensor(pos, dtype=torch.float32, device=dev); zz = torch.tensor(z, device=dev)
    if cell is None:
        ei = radius_graph(p, 5.0, ptr=torch.tensor([0, len(z)], device=dev)); extra = {}
    else:
        c = torch.tensor(cell[0], dtype=torch.float32, device=dev); pbc = torch.tensor([True, True, True], device=dev)
        ei, co = single_radius_graph(p, pbc, c, 5.0); extra = {"cell": c[None], "cell_offsets": co, "pbc": pbc[None]}
    res = {}
    orig_mlp, orig_uv = fused._mlp_packs, fused._packed_uv_frag
    for impl in ("wq", "wm", "sb"):
        for node in ("fused", "library"):
            os.environ["XEQ_MESSAGE_IMPL"] = impl
            fused._mlp_packs = orig_mlp if node == "fused" else (lambda seq: None)
            fused._packed_uv_frag = orig_uv if node == "fused" else (lambda module: None)
            torch.manual_seed(0)
            m = XPaiNNLMP(unit_style="metal", replay=True).eval().requires_grad_(False).to(dev)
            def step():
                with torch.enable_grad():
                    return m({"pos": p, "atomic_numbers": zz, "edge_index": ei, **extra}, True, False)["forces"]
            res[f"{impl}/{node}"] = timeit(step)
    fused._mlp_packs, fused._packed_uv_frag = orig_mlp, orig_uv
    print(name, "|", ", ".join(f"{k} {v:.3f}" for k, v in res.items()), flush=True)
pos, z, ptr = syn.synth_aspirin(); run("aspirin", pos, z)
pos, z, ptr, cell = syn.synth_water_box(4, seed=5); run("water-64", pos, z, cell)
pos, z, ptr, cell = syn.synth_water_box(8, seed=5); run("water-512", pos, z, cell)

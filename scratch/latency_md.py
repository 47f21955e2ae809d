"""Per-step latency of the MD front ends: GROMACS-style model (neighbour search inside, energy +
autograd forces), LAMMPS-style model (neighbour list given) eager vs HIP-graph replay, ASE-style calculator."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd.data import single_radius_graph
from xequinet_amd.cluster import radius_graph
from xequinet_amd.interface import XPaiNNGMX, XPaiNNLMP, XequiCalculator
from xequinet_amd.nn import resolve_model
from xequinet_amd.utils import set_default_units
dev = torch.device("cuda", 0)
set_default_units({"energy": "eV"})
def mk(cls, **kw):
    torch.manual_seed(0)
    return cls(**kw).eval().requires_grad_(False).to(dev)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def run(name, pos, z, cell=None):
    p = torch.tensor(pos, dtype=torch.float32, device=dev); zz = torch.tensor(z, device=dev)
    c = None if cell is None else torch.tensor(cell[0], dtype=torch.float32, device=dev)
    pbc = None if cell is None else torch.tensor([True, True, True], device=dev)
    if cell is None:
        ei = radius_graph(p, 5.0, ptr=torch.tensor([0, len(z)], device=dev)); extra = {}
    else:
        ei, co = single_radius_graph(p, pbc, c, 5.0); extra = {"cell": c[None], "cell_offsets": co, "pbc": pbc[None]}
    out = {}
    for tag, replay in (("lmp eager", False), ("lmp replay", True)):
        m = mk(XPaiNNLMP, unit_style="metal", replay=replay)
        def step():
            with torch.enable_grad():
                return m({"pos": p, "atomic_numbers": zz, "edge_index": ei, **extra}, True, False)["forces"]
        out[tag] = timeit(step)
        if replay:      # the engine hands over a DIFFERENT list every step (here: two orders of the same pairs, alternating)
            alt = [ei, ei.flip(1).contiguous()]
            ex = [extra, {k: (v.flip(0).contiguous() if k == "cell_offsets" else v) for k, v in extra.items()}]
            cnt = [0]
            def step_new():
                cnt[0] += 1
                i = cnt[0] & 1
                with torch.enable_grad():
                    return m({"pos": p, "atomic_numbers": zz, "edge_index": alt[i], **ex[i]}, True, False)["forces"]
            out["lmp replay, new list every step"] = timeit(step_new)
    try:    # the scriptable model: the whole evaluation enqueued from C++ (xeq::xpainn_eval), no capture
        from xequinet_amd.interface.scripted import XPaiNNLMPScript
        torch.manual_seed(0)
        sm = XPaiNNLMPScript(mk(XPaiNNLMP, unit_style="metal"), unit_style="metal")
        def sstep():
            return sm({"pos": p, "atomic_numbers": zz, "edge_index": ei, **{k: v for k, v in extra.items() if k != "pbc"}}, True, False)["forces"]
        out["lmp scripted (one registered operator)"] = timeit(sstep)
    except Exception as err:
        out["lmp scripted: " + str(err)[:60]] = float("nan")
    for tag, replay, whole in (("gmx eager (search + energy + autograd)", False, False), ("gmx replay", True, False),
                               ("gmx whole-step graph (search inside the graph)", True, True)):
        g = mk(XPaiNNGMX, replay=replay, whole_step=whole)
        def gstep():
            x = (p / 10).requires_grad_(True)
            e = g(x, zz, None if c is None else c / 10, pbc)
            return torch.autograd.grad(e.sum(), x)[0]
        out[tag] = timeit(gstep)
    print(name, "E =", ei.shape[1], "|", ", ".join(f"{k} {v:.3f} ms" for k, v in out.items()), flush=True)
pos, z, ptr = syn.synth_aspirin(); run("aspirin (21 atoms)", pos, z)
pos, z, ptr, cell = syn.synth_water_box(4, seed=5); run("water-64 (192 atoms, PBC)", pos, z, cell)
pos, z, ptr, cell = syn.synth_water_box(8, seed=5); run("water-512 (1536 atoms, PBC)", pos, z, cell)

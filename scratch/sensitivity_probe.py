"""Which intermediate tensor amplifies fp32 rounding in an ill-conditioned molecule?  fp64 oracle with relative noise of one fp32 ulp
injected at ONE place at a time (forward value or gradient), force change reported.  CPU only."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd.nn import resolve_model
torch.manual_seed(0)
model = resolve_model("xpainn")
sd = {k: v.detach().double().clone() for k, v in model.state_dict().items()}
o = orc.XPaiNNOracle(sd)
pos, z, ptr, _ = syn.make_workload("qm9_1024", seed=1234)
mols = np.sort(np.random.default_rng(7).choice(len(ptr) - 1, size=160, replace=False))
g = mols[int(sys.argv[1]) if len(sys.argv) > 1 else 115]
p = pos[ptr[g]:ptr[g+1]].astype(np.float32).astype(np.float64); zz = z[ptr[g]:ptr[g+1]]
pp = np.array([0, len(p)])
ei = orc.radius_graph_canonical(p.astype(np.float32), pp, 5.0)
EPS = 6e-8
def run(inject=None, grad_inject=None, seed=0):
    gen = torch.Generator().manual_seed(seed)
    d = {"pos": torch.tensor(p), "atomic_numbers": torch.tensor(zz.astype(np.int64)), "edge_index": torch.tensor(ei),
         "batch": torch.zeros(len(p), dtype=torch.long), "ptr": torch.tensor(pp)}
    d = orc.compute_edge_data(d, True, False)
    d = o.embedding(d)
    def touch(name):
        for key in ("node_invariant", "node_equivariant"):
            t = d[key]
            if inject == (name, key):
                t = t * (1 + EPS * torch.randn(t.shape, generator=gen, dtype=t.dtype))
            if grad_inject == (name, key) and t.requires_grad:
                t = t.clone(); t.register_hook(lambda gr: gr * (1 + EPS * torch.randn(gr.shape, generator=gen, dtype=gr.dtype)))
            d[key] = t
    touch("embed")
    for i in range(o.blocks):
        d = o.message(i, d); touch(f"msg{i}")
        d = o.update(i, d); touch(f"upd{i}")
    d = o.energy_out(d)
    (gr,) = torch.autograd.grad([d["energy"]], [d[orc.POSITIONS]], [torch.ones_like(d["energy"])])
    return -gr
F0 = run()
print(f"molecule {g}: {len(p)} atoms, {ei.shape[1]} edges, max |F| {F0.abs().max():.3f}")
names = ["embed"] + [f"{k}{i}" for i in range(o.blocks) for k in ("msg", "upd")]
for nm in names:
    for key in ("node_invariant", "node_equivariant"):
        ev = np.mean([float((run(inject=(nm, key), seed=s) - F0).pow(2).mean().sqrt()) for s in range(3)])
        eg = np.mean([float((run(grad_inject=(nm, key), seed=s) - F0).pow(2).mean().sqrt()) for s in range(3)])
        print(f"  {nm:6s} {key:17s}: value noise -> dF rms {ev:.2e}   gradient noise -> dF rms {eg:.2e}")
# edge-level quantities, computed once and read by every block
def run_edge(key, seed=0, rel=EPS):
    gen = torch.Generator().manual_seed(seed)
    d = {"pos": torch.tensor(p), "atomic_numbers": torch.tensor(zz.astype(np.int64)), "edge_index": torch.tensor(ei),
         "batch": torch.zeros(len(p), dtype=torch.long), "ptr": torch.tensor(pp)}
    d = orc.compute_edge_data(d, True, False)
    if key in ("edge_vector", "edge_length"):
        d[key] = d[key] * (1 + rel * torch.randn(d[key].shape, generator=gen, dtype=torch.float64))
    d = o.embedding(d)
    if key in ("rbf", "fcut", "rsh"):
        d[key] = d[key] * (1 + rel * torch.randn(d[key].shape, generator=gen, dtype=torch.float64))
    if key == "rbf_abs":   # absolute noise of one fp32 ulp of the PHASE omega_k d in sin(omega_k d): what evaluating the basis in fp32 does
        freq = sd["mods.embedding.rbf.freq"].reshape(1, -1)
        dist = d["edge_length"].unsqueeze(-1)
        phase = freq * dist
        noisy = phase * (1 + rel * torch.randn(phase.shape, generator=gen, dtype=torch.float64))
        d["rbf"] = (2.0 / o.cutoff) ** 0.5 * torch.sin(noisy) / (dist + 1e-5)
    for i in range(o.blocks):
        d = o.message(i, d); d = o.update(i, d)
    d = o.energy_out(d)
    (gr,) = torch.autograd.grad([d["energy"]], [d[orc.POSITIONS]], [torch.ones_like(d["energy"])])
    return -gr
for key in ("edge_vector", "edge_length", "rbf", "rbf_abs", "fcut", "rsh"):
    ev = np.mean([float((run_edge(key, seed=s) - F0).pow(2).mean().sqrt()) for s in range(3)])
    print(f"  {key:12s}: one-ulp noise -> dF rms {ev:.2e}")
dist = torch.tensor(p)[ei[0]] - torch.tensor(p)[ei[1]]
dd = dist.norm(dim=1)
print("distances: min %.4f max %.4f; pairs within 0.02 of the cutoff: %d" % (dd.min(), dd.max(), int((dd > 4.98).sum())))
# inside the update block: absolute noise of one fp32 ulp of the typical magnitude on V (input of Invariant) / U, V (EquivariantDot)
import oracle.xpainn_oracle as O
def run_patched(which, seed=0):
    gen = torch.Generator().manual_seed(seed)
    inv0, lin0 = O.invariant, O.o3_linear
    def noisy_linear(irreps, x, w, b):
        y = lin0(irreps, x, w, b)
        return y + EPS * y.abs().mean() * torch.randn(y.shape, generator=gen, dtype=y.dtype) if which == "o3_linear" else y
    O.o3_linear = noisy_linear
    try:
        return run()
    finally:
        O.invariant, O.o3_linear = inv0, lin0
ev = np.mean([float((run_patched("o3_linear", seed=s) - F0).pow(2).mean().sqrt()) for s in range(3)])
print(f"  U, V = o3.Linear(xhat) with ABSOLUTE noise of one fp32 ulp of their mean magnitude -> dF rms {ev:.2e}")
# the per-edge gradient dE/dvec_e (what the reverse message kernels produce) and its scatter onto the atoms
def run_gvec(seed=0, rel=EPS, report=False):
    gen = torch.Generator().manual_seed(seed)
    d = {"pos": torch.tensor(p), "atomic_numbers": torch.tensor(zz.astype(np.int64)), "edge_index": torch.tensor(ei),
         "batch": torch.zeros(len(p), dtype=torch.long), "ptr": torch.tensor(pp)}
    d = orc.compute_edge_data(d, True, False)
    keep = {}
    v = d["edge_vector"].clone()
    def hook(gr):
        keep["g"] = gr.detach().clone()
        return gr * (1 + rel * torch.randn(gr.shape, generator=gen, dtype=gr.dtype))
    v.register_hook(hook)
    d["edge_vector"] = v
    d["edge_length"] = torch.linalg.norm(v, dim=-1)
    d = o.embedding(d)
    for i in range(o.blocks):
        d = o.message(i, d); d = o.update(i, d)
    d = o.energy_out(d)
    (gr,) = torch.autograd.grad([d["energy"]], [d[orc.POSITIONS]], [torch.ones_like(d["energy"])])
    if report:
        print(f"  |dE/dvec_e|: max {keep['g'].abs().max():.3e} rms {keep['g'].pow(2).mean().sqrt():.3e}  against max |F| {F0.abs().max():.3f}")
    return -gr
ev = np.mean([float((run_gvec(seed=s, report=(s == 0)) - F0).pow(2).mean().sqrt()) for s in range(3)])
print(f"  dE/dvec_e with one-ulp relative noise -> dF rms {ev:.2e}")

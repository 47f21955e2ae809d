"""Where does the fp32 HIP path lose accuracy?  Node features after every block, HIP fp32 and oracle fp32, each against the fp64
oracle on the same molecules: python scratch/stage_probe.py [workload] [n_molecules]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from tests.test_gpu_parity import _build, f32_twin, DEV, _t
from xequinet_amd import keys
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
from xequinet_amd.nn.basic import compute_edge_data

name = sys.argv[1] if len(sys.argv) > 1 else "qm9_1024"
n_s = int(sys.argv[2]) if len(sys.argv) > 2 else 160
model, oracle = _build(torch.float32)
pos, z, ptr, _ = syn.make_workload(name, seed=1234)
mols = np.sort(np.random.default_rng(7).choice(len(ptr) - 1, size=n_s, replace=False))
idx = np.concatenate([np.arange(ptr[g], ptr[g + 1]) for g in mols])
p = pos[idx].astype(np.float32); zz = z[idx]
pp = np.concatenate([[0], np.cumsum(np.diff(ptr)[mols])]).astype(np.int64)

def oracle_stages(o, dtype):
    ei = orc.radius_graph_canonical(p, pp, 5.0)
    d = {"pos": torch.tensor(p.astype(np.float64)).to(dtype), "atomic_numbers": torch.tensor(zz.astype(np.int64)), "edge_index": torch.tensor(ei),
         "batch": torch.tensor(np.repeat(np.arange(len(mols)), np.diff(pp))), "ptr": torch.tensor(pp)}
    d[orc.POSITIONS] = d[orc.POSITIONS].detach().clone()
    d = orc.compute_edge_data(d, True, False)
    d = o.embedding(d)
    st = [("embed", d["node_invariant"].detach().clone(), d["node_equivariant"].detach().clone())]
    for i in range(o.blocks):
        d = o.message(i, d); st.append((f"msg{i}", d["node_invariant"].detach().clone(), d["node_equivariant"].detach().clone()))
        d = o.update(i, d); st.append((f"upd{i}", d["node_invariant"].detach().clone(), d["node_equivariant"].detach().clone()))
    d = o.energy_out(d)
    (g,) = torch.autograd.grad([d["energy"]], [d[orc.POSITIONS]], [torch.ones_like(d["energy"])])
    return st, (-g).detach()

s64, F64 = oracle_stages(oracle, torch.float64)
s32, F32 = oracle_stages(f32_twin(oracle), torch.float32)
# HIP, module by module
b = NeighborTransform(5.0)(XequiBatch(_t(p, torch.float32), _t(zz), _t(pp)))
data = b.to_dict()
with torch.enable_grad():
    from xequinet_amd.nn import training
    data[training.PARAM_GRADS] = False; data[training.TRAIN_PASS] = False
    data = compute_edge_data(data=data, compute_forces=True, compute_virial=False)
    sh = []
    for nm, mod in model.mods.items():
        data = mod(data)
        if nm.startswith(("embedding", "message", "update")):
            x = data[keys.NODE_EQUIVARIANT]
            sh.append((nm, data[keys.NODE_INVARIANT].detach().double().cpu(), None if x is None else x.detach().double().cpu()))
    from xequinet_amd.nn.basic import compute_properties
    out = compute_properties(data=data, compute_forces=True, compute_virial=False, training=False, extra_properties=[])
FH = out["forces"].detach().double().cpu()
def rel(a, ref):
    if a is None: return "      --        "
    e = (a.double() - ref); return f"{e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt():.2e} max {e.abs().max():.2e}"
print(f"{name}: {len(mols)} molecules, {len(idx)} atoms; relative rms / max abs error against the fp64 oracle")
print(f"{'stage':8s} | s: HIP              | s: oracle32         | x: HIP              | x: oracle32")
for (n64, a, c), (_, a32, c32), (nh, ah, ch) in zip(s64, s32, sh):
    print(f"{n64:8s} | {rel(ah, a)} | {rel(a32, a)} | {rel(ch, c)} | {rel(c32, c)}")
eh, e32 = (FH - F64).abs(), (F32.double() - F64).abs()
print(f"forces: HIP rms {eh.pow(2).mean().sqrt():.2e} max {eh.max():.2e} | oracle32 rms {e32.pow(2).mean().sqrt():.2e} max {e32.max():.2e}")
w = int(eh.max(1).values.argmax()); g = int(np.searchsorted(pp, w, side='right') - 1)
print(f"worst HIP atom {w}: molecule {g} of {pp[g+1]-pp[g]} atoms, Z={zz[w]}, |F|={F64[w].norm():.3f}, HIP err {eh[w].max():.2e}, oracle32 err {e32[w].max():.2e}; molecule rms HIP {eh[pp[g]:pp[g+1]].pow(2).mean().sqrt():.2e} o32 {e32[pp[g]:pp[g+1]].pow(2).mean().sqrt():.2e}")
for k, (n64, a, c) in enumerate(s64):
    sl = slice(pp[g], pp[g+1])
    print(f"   {n64:7s} worst molecule: s HIP {rel(sh[k][1][sl], a[sl])} o32 {rel(s32[k][1][sl], a[sl])} | x HIP {rel(sh[k][2][sl] if sh[k][2] is not None else None, c[sl])} o32 {rel(s32[k][2][sl], c[sl])}")
# the same forces through the differentiable form of the blocks (nn/training.py: ATen f32 ops on the GPU, autograd's own reverse pass)
import copy
m2 = copy.deepcopy(model).train().requires_grad_(True)
b2 = NeighborTransform(5.0)(XequiBatch(_t(p, torch.float32), _t(zz), _t(pp)))
with torch.enable_grad():
    o2 = m2(b2.to_dict(), compute_forces=True)
FD = o2["forces"].detach().double().cpu()
ed = (FD - F64).abs()
sl = slice(pp[g], pp[g+1])
print(f"differentiable form (ATen f32 + autograd): rms {ed.pow(2).mean().sqrt():.2e} max {ed.max():.2e}; worst-HIP molecule rms {ed[sl].pow(2).mean().sqrt():.2e}")
print("per-molecule rms ratio HIP / oracle32, ten worst:")
r = []
for k in range(len(mols)):
    s_ = slice(pp[k], pp[k+1]); r.append((float(eh[s_].pow(2).mean().sqrt()), float(e32[s_].pow(2).mean().sqrt()), float(ed[s_].pow(2).mean().sqrt()), k, pp[k+1]-pp[k]))
for a in sorted(r, reverse=True)[:10]: print(f"   mol {a[3]:4d} ({a[4]:2d} atoms): HIP {a[0]:.2e} oracle32 {a[1]:.2e} ATen-GPU {a[2]:.2e}")

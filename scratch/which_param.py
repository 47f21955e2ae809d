"""Which of the parity-test model's randomised parameters makes a molecule ill-conditioned in fp32?  CPU only."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.xpainn_oracle as O
from xequinet_amd.data import synthetic as syn
from xequinet_amd.nn import resolve_model
pos, z, ptr, _ = syn.make_workload("qm9_1024", seed=1234)
mols = np.sort(np.random.default_rng(7).choice(len(ptr) - 1, size=160, replace=False))
ms = [mols[k] for k in (115, 141, 75, 53, 27, 61, 93, 30, 79, 105)]
idx = np.concatenate([np.arange(ptr[g], ptr[g + 1]) for g in ms])
p = pos[idx].astype(np.float32); pp = np.concatenate([[0], np.cumsum(np.diff(ptr)[ms])]).astype(np.int64)
ei = O.radius_graph_canonical(p, pp, 5.0)
def err(sd):
    out = []
    for dt in (torch.float64, torch.float32):
        s_ = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
        d = {"pos": torch.tensor(p.astype(np.float64)).to(dt), "atomic_numbers": torch.tensor(z[idx].astype(np.int64)), "edge_index": torch.tensor(ei),
             "batch": torch.tensor(np.repeat(np.arange(len(ms)), np.diff(pp))), "ptr": torch.tensor(pp)}
        out.append(O.XPaiNNOracle(s_)(d, compute_forces=True)["forces"].double())
    e = (out[1] - out[0]).abs()
    return float(e.pow(2).mean().sqrt()), float(e.max()), float(out[0].abs().max())
def build(skip=()):
    torch.manual_seed(0)
    model = resolve_model("xpainn")
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for name, q in model.named_parameters():
            if name.endswith(("norm.weight", "affine_weight")):
                v = 1.0 + 0.2 * torch.randn(q.shape, generator=g)
                if not any(s in name for s in skip): q.copy_(v)
            elif name.endswith(("bias", "affine_bias")):
                v = 0.1 * torch.randn(q.shape, generator=g)
                if not any(s in name for s in skip): q.copy_(v)
    return {k: v.detach().double().clone() for k, v in model.state_dict().items()}
print("default init                      : rms %.2e max %.2e (max |F| %.2f)" % err(build(skip=("",))))
print("test model (all randomised)       : rms %.2e max %.2e (max |F| %.2f)" % err(build()))
for s in ("affine_bias", "affine_weight", "norm.weight", "norm.bias", "update_U.bias", "update_V.bias", "scalar_mlp", "update_mlp", "rbf_lin.bias", "embedding", "out_mlp"):
    print("test model without randomised %-16s: rms %.2e max %.2e (max |F| %.2f)" % ((s,) + err(build(skip=(s,)))))

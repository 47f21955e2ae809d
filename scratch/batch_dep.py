import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.xpainn_oracle as O
from xequinet_amd.data import synthetic as syn
from xequinet_amd.nn import resolve_model
torch.manual_seed(0)
model = resolve_model("xpainn")
sd = {k: v.detach().double().clone() for k, v in model.state_dict().items()}
sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
o64, o32 = O.XPaiNNOracle(sd), O.XPaiNNOracle(sd32)
pos, z, ptr, _ = syn.make_workload("qm9_1024", seed=1234)
mols = np.sort(np.random.default_rng(7).choice(len(ptr) - 1, size=160, replace=False))
def ev(o, ms, dt):
    idx = np.concatenate([np.arange(ptr[g], ptr[g + 1]) for g in ms])
    p = pos[idx].astype(np.float32); pp = np.concatenate([[0], np.cumsum(np.diff(ptr)[ms])]).astype(np.int64)
    ei = O.radius_graph_canonical(p, pp, 5.0)
    d = {"pos": torch.tensor(p.astype(np.float64)).to(dt), "atomic_numbers": torch.tensor(z[idx].astype(np.int64)), "edge_index": torch.tensor(ei),
         "batch": torch.tensor(np.repeat(np.arange(len(ms)), np.diff(pp))), "ptr": torch.tensor(pp)}
    return o(d, compute_forces=True)["forces"].double(), pp
k = 115
Fa64, _ = ev(o64, [mols[k]], torch.float64); Fa32, _ = ev(o32, [mols[k]], torch.float32)
print("alone: f32 vs f64 rms", float((Fa32 - Fa64).pow(2).mean().sqrt()), "pos range", pos[ptr[mols[k]]:ptr[mols[k]+1]].min(), pos[ptr[mols[k]]:ptr[mols[k]+1]].max())
for n in (2, 8, 40, 160):
    ms = list(mols[max(0, k - n // 2): max(0, k - n // 2) + n]);  j = ms.index(mols[k])
    F64, pp = ev(o64, ms, torch.float64); F32, _ = ev(o32, ms, torch.float32)
    sl = slice(pp[j], pp[j + 1])
    print(f"in a batch of {n:3d}: this molecule f32 vs f64 rms {float((F32[sl] - F64[sl]).pow(2).mean().sqrt()):.2e}; f64 batch vs f64 alone {float((F64[sl] - Fa64).abs().max()):.1e}; whole batch rms {float((F32 - F64).pow(2).mean().sqrt()):.2e}")

"""Which module makes a node's bits depend on its batch?  Evaluates 512 molecules whole and the first 100 alone, compares the
node features behind every module (forward) and the forces: python scratch/batch_bits_probe.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import _build, _t
from xequinet_amd import keys
from xequinet_amd.data import NeighborTransform, XequiBatch, synthetic as syn
model, _ = _build(torch.float32)
pos, z, ptr = syn.synth_qm9_batch(512, seed=99)
def run(g1):
    a = int(ptr[g1])
    b = NeighborTransform(5.0)(XequiBatch(_t(pos[:a], torch.float32), _t(z[:a]), _t(ptr[: g1 + 1])))
    feats = {}
    hooks = []
    for name, m in model.mods.items():
        hooks.append(m.register_forward_hook(lambda mod, inp, out, name=name: feats.__setitem__(name, (out[keys.NODE_INVARIANT].detach().clone(), None if out.get(keys.NODE_EQUIVARIANT) is None else out[keys.NODE_EQUIVARIANT].detach().clone()) if keys.NODE_INVARIANT in out else None)))
    with torch.enable_grad():
        out = model(b.to_dict(), compute_forces=True)
    for h in hooks: h.remove()
    return feats, out["energy"].detach(), out["forces"].detach(), a
fa, Ea, Fa, _ = run(512)
fb, Eb, Fb, nb = run(100)
for name in fa:
    if fa[name] is None: continue
    sa, xa = fa[name]; sb, xb = fb[name]
    ds = float((sa[:nb] - sb).abs().max())
    dx = None if xa is None or xb is None else float((xa[:nb] - xb).abs().max())
    print(f"{name:12s} max |ds| {ds:.3e}  max |dx| {dx if dx is None else format(dx, '.3e')}")
print("energy", float((Ea[:100] - Eb).abs().max()), "forces", float((Fa[:nb] - Fb).abs().max()))

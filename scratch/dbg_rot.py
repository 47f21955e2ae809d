import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from tests.test_gpu_parity import _build, _t
from xequinet_amd.data import NeighborTransform, XequiBatch
model, _ = _build(torch.float32)
pos, z, ptr = syn.synth_qm9_batch(1024, seed=1234)
def run(p, zz, pp):
    b = NeighborTransform(5.0)(XequiBatch(_t(p, torch.float32), _t(zz), _t(pp)))
    with torch.enable_grad():
        out = model(b.to_dict(), compute_forces=True)
    return out["energy"].detach().cpu().double().numpy(), out["forces"].cpu().double().numpy(), b.edge_index.cpu().numpy()
E, Fo, ei = run(pos, z, ptr)
rng = np.random.default_rng(0)
Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
if np.linalg.det(Q) < 0: Q[:, 0] *= -1
E2, F2, ei2 = run(pos @ Q.T + np.array([1.0, -2.0, 0.5]), z, ptr)
d = np.abs(F2 - Fo @ Q.T)
print('Fmax', np.abs(Fo).max(), 'edges', ei.shape, ei2.shape, 'same graph', ei.shape == ei2.shape and (ei == ei2).all())
idx = np.argsort(-d.max(1))[:8]
seg = np.searchsorted(ptr, idx, side='right') - 1
for i, g in zip(idx, seg):
    print(i, 'mol', g, 'err', d[i].max(), '|F|', np.abs(Fo[i]).max(), 'Emol', E[g], E2[g])
# same input twice: determinism
E3, F3, _ = run(pos, z, ptr)
print('repeat identical:', np.array_equal(Fo, F3), np.abs(Fo-F3).max())

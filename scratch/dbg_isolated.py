import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_training import _model, SMALL, DEV
from oracle import xpainn_oracle as orc
from xequinet_amd import keys, train
from xequinet_amd.data import synthetic as syn
from xequinet_amd.nn import training as tr
n_iso = int(sys.argv[1]) if len(sys.argv) > 1 else 13
pos, z, ptr = syn.synth_qm9_batch(14, seed=41)
iso = np.zeros((n_iso, 3)); iso[:, 0] = 1.0e4 + 100.0 * np.arange(n_iso)
pos = np.concatenate([pos, iso]); z = np.concatenate([z, np.ones(n_iso, dtype=z.dtype)]); ptr = np.concatenate([ptr, [ptr[-1] + n_iso]])
ei = orc.radius_graph_canonical(pos, ptr, 5.0)
batch = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
dt = torch.float64
dev = {"pos": torch.tensor(pos, dtype=dt, device=DEV), "atomic_numbers": torch.tensor(z.astype(np.int64), device=DEV), "edge_index": torch.tensor(ei, device=DEV),
       "batch": torch.tensor(batch, device=DEV), "ptr": torch.tensor(ptr, device=DEV)}
g = torch.Generator().manual_seed(2)
G, n = len(ptr) - 1, len(pos)
e_t = torch.randn(G, generator=g, dtype=dt).to(DEV); f_t = torch.randn(n, 3, generator=g, dtype=dt).to(DEV)
mask_g = torch.ones(G, dtype=dt, device=DEV); mask_g[-1] = 0
mask_a = torch.ones(n, 1, dtype=dt, device=DEV); mask_a[-n_iso:] = 0
print("N", n, "E", ei.shape[1])
grads = {}
for msg, node in ((False, False), (True, False), (False, True), (True, True)):
    tr.NATIVE_MESSAGE, tr.NATIVE_NODE = msg, node
    model = _model(dt, **SMALL).train()
    out = model(dict(dev), True, False)
    loss = (((out[keys.TOTAL_ENERGY] - e_t) ** 2) * mask_g).sum() / mask_g.sum() + 5.0 * (((out[keys.FORCES] - f_t) ** 2) * mask_a).sum() / (3 * mask_a.sum())
    loss.backward()
    grads[(msg, node)] = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    if (msg, node) != (False, False):
        ref = grads[(False, False)]
        worst = max(((float((ref[k] - grads[(msg, node)][k]).abs().max() / ref[k].abs().max().clamp_min(1e-30))), k) for k in ref)
        print("message", msg, "node", node, "loss", loss.item(), "worst", worst)

import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_training import _batch, _targets, _model, SMALL, DEV
from xequinet_amd import keys, train
from xequinet_amd.data import NeighborTransform, XequiBatch
w = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 5.0}
for n_mol, seed in ((20, 40), (14, 41), (24, 42)):
    res = {}
    for dt in (torch.float64, torch.float32):
        host, dev = _batch(n_mol, seed, dt)
        tgt = _targets(host, 30 + seed, False)
        model = _model(dt, **SMALL).train()
        e_t, f_t = tgt[keys.TOTAL_ENERGY].to(dt).to(DEV), tgt[keys.FORCES].to(dt).to(DEV)
        l, _ = train.weighted_loss(model(dict(dev), True, False), {keys.TOTAL_ENERGY: e_t, keys.FORCES: f_t, keys.BATCH_PTR: dev["ptr"]}, w)
        l.backward()
        res[dt] = {k: p.grad.double().clone() for k, p in model.named_parameters() if p.grad is not None}
    a, b = res[torch.float64], res[torch.float32]
    rel = sorted(((float((a[k] - b[k]).abs().max() / a[k].abs().max().clamp_min(1e-30)), k) for k in a), reverse=True)
    print(n_mol, "fp32 against fp64 gradients, worst:", rel[:3], "median", rel[len(rel) // 2][0])

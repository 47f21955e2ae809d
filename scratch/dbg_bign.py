import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_training import _batch, _targets, _model, SMALL, DEV
from xequinet_amd import keys, train
from xequinet_amd.nn import training as tr
n_mol = int(sys.argv[1]) if len(sys.argv) > 1 else 20
host, dev = _batch(n_mol, 5, torch.float64)
tgt = {k: v.to(DEV) for k, v in _targets(host, 7, False).items()}
w = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 10.0}
print("N", host["pos"].shape[0], "E", host["edge_index"].shape[1])
grads = {}
for msg, node in ((False, False), (True, False), (False, True), (True, True)):
    tr.NATIVE_MESSAGE, tr.NATIVE_NODE = msg, node
    model = _model(torch.float64, **SMALL).train()
    loss, _ = train.weighted_loss(model(dict(dev), True, False), tgt, w)
    loss.backward()
    grads[(msg, node)] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    if (msg, node) != (False, False):
        ref = grads[(False, False)]
        worst = max(((float((ref[k] - grads[(msg, node)][k]).abs().max() / ref[k].abs().max().clamp_min(1e-30))), k) for k in ref)
        print("message", msg, "node", node, "loss", loss.item(), "worst", worst)

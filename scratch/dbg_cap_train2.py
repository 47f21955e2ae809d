import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_training import _batch, _targets, _model, SMALL, DEV
from xequinet_amd import keys, train, runtime
from xequinet_amd.data import NeighborTransform, XequiBatch
batches = []
for k, n_mol in enumerate((20, 14, 24)):
    host, dev = _batch(n_mol, 40 + k, torch.float32)
    batches.append((host, dev, _targets(host, 70 + k, False)))
cap = (max(b[0]["pos"].shape[0] for b in batches) + 8, max(b[0]["ptr"].numel() - 1 for b in batches), max(runtime.pair_capacity(b[0]["ptr"].numpy()) for b in batches))
fast, slow = _model(torch.float32, **SMALL).train(), _model(torch.float32, **SMALL).train()
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
opt_f = torch.optim.Adam(fast.parameters(), lr=lr, capturable=True)
step = train.GraphedTrainStep(fast, opt_f, cap, energy_weight=1.0, forces_weight=5.0)
w = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 5.0}
for it, (host, dev, tgt) in enumerate(batches + batches[:1]):
    slow.load_state_dict(fast.state_dict())
    e_t, f_t = tgt[keys.TOTAL_ENERGY].float().to(DEV), tgt[keys.FORCES].float().to(DEV)
    loss_f = step(dev["pos"], dev["atomic_numbers"], dev["ptr"], e_t, batch=dev["batch"], target_forces=f_t).item()
    b = NeighborTransform(5.0)(XequiBatch(dev["pos"], dev["atomic_numbers"], dev["ptr"]))
    slow.zero_grad(set_to_none=True)
    loss_s, _ = train.weighted_loss(slow(b.to_dict(), True, False), {keys.TOTAL_ENERGY: e_t, keys.FORCES: f_t, keys.BATCH_PTR: dev["ptr"]}, w)
    loss_s.backward()
    diffs = sorted(((float((p.grad - q.grad).abs().max() / q.grad.abs().max().clamp_min(1e-12)), n) for (n, p), (_, q) in zip(fast.named_parameters(), slow.named_parameters()) if q.grad is not None), reverse=True)
    print(it, "loss", loss_f, loss_s.item(), "worst", diffs[:3])

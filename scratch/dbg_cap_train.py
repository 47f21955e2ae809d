import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_training import _batch, _targets, _model, SMALL, DEV
from xequinet_amd import keys, train, runtime
from xequinet_amd.data import NeighborTransform, XequiBatch
host, dev = _batch(20, 40, torch.float32)
tgt = _targets(host, 70, False)
cap = (host["pos"].shape[0] + 8, host["ptr"].numel() - 1, runtime.pair_capacity(host["ptr"].numpy()))
fast, slow = _model(torch.float32, **SMALL).train(), _model(torch.float32, **SMALL).train()
slow.load_state_dict(fast.state_dict())
opt_f = torch.optim.Adam(fast.parameters(), lr=1e-3, capturable=True)
opt_s = torch.optim.Adam(slow.parameters(), lr=1e-3, capturable=True)
step = train.GraphedTrainStep(fast, opt_f, cap, energy_weight=1.0, forces_weight=5.0)
orig = step._body
def body():
    l = orig()
    if not torch.cuda.is_current_stream_capturing(): print("eager body loss", l.item())
    return l
step._body = body
e_t, f_t = tgt[keys.TOTAL_ENERGY].float().to(DEV), tgt[keys.FORCES].float().to(DEV)
loss_f = step(dev["pos"], dev["atomic_numbers"], dev["ptr"], e_t, batch=dev["batch"], target_forces=f_t).item()
b = NeighborTransform(5.0)(XequiBatch(dev["pos"], dev["atomic_numbers"], dev["ptr"]))
t = {keys.TOTAL_ENERGY: e_t, keys.FORCES: f_t, keys.BATCH_PTR: dev["ptr"]}
res = slow(b.to_dict(), True, False)
l, terms = train.weighted_loss(res, t, {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 5.0})
print("graph", loss_f, "host", l.item(), {k: v.item() for k, v in terms.items()})

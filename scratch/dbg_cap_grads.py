import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_training import _batch, _targets, _model, SMALL, DEV
from xequinet_amd import keys, train, runtime, ops
from xequinet_amd.data import NeighborTransform, XequiBatch
dt = torch.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else torch.float32
host, dev = _batch(20, 40, dt)
tgt = _targets(host, 70, False)
n, G = host["pos"].shape[0], host["ptr"].numel() - 1
model = _model(dt, **SMALL).train()
e_t, f_t = tgt[keys.TOTAL_ENERGY].to(dt).to(DEV), tgt[keys.FORCES].to(dt).to(DEV)
w = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 5.0}
def grads_exact():
    b = NeighborTransform(5.0)(XequiBatch(dev["pos"], dev["atomic_numbers"], dev["ptr"]))
    model.zero_grad(set_to_none=True)
    l, _ = train.weighted_loss(model(b.to_dict(), True, False), {keys.TOTAL_ENERGY: e_t, keys.FORCES: f_t, keys.BATCH_PTR: dev["ptr"]}, w)
    l.backward()
    return l.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
def grads_cap(flag):
    g = runtime.GraphedStep(model, (n + 8, G + 1, runtime.pair_capacity(host["ptr"].numpy())), compute_forces=False, warmup=0, dtype=dt) if False else runtime.GraphedStep(model, (n + 8, G + 1, runtime.pair_capacity(host["ptr"].numpy())), compute_forces=False, warmup=0)
    g._load(dev["pos"].detach(), dev["atomic_numbers"], dev["ptr"], dev["batch"])
    rowptr, count = ops.radius_graph_capacity(g.pos, g.ptr, g.cutoff, g.edge_index)
    eg = ops.EdgeGraph(g.edge_index, g.n_atoms, center_sorted=True, ptr=g.ptr, c_rowptr=rowptr, symmetric=True)
    eg.edge_count_on_device = flag
    data = {keys.POSITIONS: g.pos.detach().clone(), keys.ATOMIC_NUMBERS: g.z, keys.EDGE_INDEX: g.edge_index, keys.BATCH: g.batch, keys.BATCH_PTR: g.ptr, keys.EDGE_GRAPH: eg}
    model.zero_grad(set_to_none=True)
    out = model(data, True, False)
    l = ((out[keys.TOTAL_ENERGY][:G] - e_t) ** 2).mean() + 5.0 * ((out[keys.FORCES][:n] - f_t) ** 2).mean()
    l.backward()
    return l.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
l0, g0 = grads_exact()
l0b, g0b = grads_exact()
print("exact twice", max(float((g0[k] - g0b[k]).abs().max() / g0[k].abs().max().clamp_min(1e-12)) for k in g0))
for flag in (True, False):
    l1, g1 = grads_cap(flag)
    worst = max(((float((g0[k] - g1[k]).abs().max() / g0[k].abs().max().clamp_min(1e-12))), k) for k in g0)
    print("flag", flag, "loss", l0, l1, "worst rel grad diff", worst)

import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_training import _batch, _targets, _model, SMALL, DEV
from xequinet_amd import keys, train, runtime, ops
from xequinet_amd.nn import training as tr
from xequinet_amd.data import NeighborTransform, XequiBatch
host, dev = _batch(14, 41, torch.float32)
tgt = _targets(host, 71, False)
model = _model(torch.float32, **SMALL).train()
w = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 5.0}
n, G = host["pos"].shape[0], host["ptr"].numel() - 1
e_t, f_t = tgt[keys.TOTAL_ENERGY].float().to(DEV), tgt[keys.FORCES].float().to(DEV)
b = NeighborTransform(5.0)(XequiBatch(dev["pos"], dev["atomic_numbers"], dev["ptr"]))
model.zero_grad(set_to_none=True)
l, _ = train.weighted_loss(model(b.to_dict(), True, False), {keys.TOTAL_ENERGY: e_t, keys.FORCES: f_t, keys.BATCH_PTR: dev["ptr"]}, w)
l.backward()
g0 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
for pad_atoms, pad_graphs, pad_edges in ((11, 1, 0), (12, 1, 0), (13, 1, 0), (16, 1, 0), (32, 1, 0), (64, 1, 0)):
    cap = (n + pad_atoms, G + pad_graphs, runtime.pair_capacity(host["ptr"].numpy()) + pad_edges)
    g = runtime.GraphedStep(model, cap, compute_forces=False, warmup=0)
    g._load(dev["pos"].detach(), dev["atomic_numbers"], dev["ptr"], dev["batch"])
    rowptr, count = ops.radius_graph_capacity(g.pos, g.ptr, g.cutoff, g.edge_index)
    eg = ops.EdgeGraph(g.edge_index, g.n_atoms, center_sorted=True, ptr=g.ptr, c_rowptr=rowptr, symmetric=True)
    eg.edge_count_on_device = True
    data = {keys.POSITIONS: g.pos.detach().clone(), keys.ATOMIC_NUMBERS: g.z, keys.EDGE_INDEX: g.edge_index, keys.BATCH: g.batch, keys.BATCH_PTR: g.ptr, keys.EDGE_GRAPH: eg}
    model.zero_grad(set_to_none=True)
    out = model(data, True, False)
    l1 = ((out[keys.TOTAL_ENERGY][:G] - e_t) ** 2).mean() + 5.0 * ((out[keys.FORCES][:n] - f_t) ** 2).mean()
    l1.backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    worst = max(((float((g0[k] - g1[k]).abs().max() / g0[k].abs().max().clamp_min(1e-12))), k) for k in g0)
    print("pads: atoms", pad_atoms, "graphs", pad_graphs, "edge slots", g.n_edges - int(count), "worst", worst)

import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_training import _batch, _targets, _model, SMALL, DEV
from xequinet_amd import keys, train, runtime, ops
from xequinet_amd.nn import training as tr
from xequinet_amd.data import NeighborTransform, XequiBatch
batches = []
for k, n_mol in enumerate((20, 14)):
    host, dev = _batch(n_mol, 40 + k, torch.float32)
    batches.append((host, dev, _targets(host, 70 + k, False)))
cap = (max(b[0]["pos"].shape[0] for b in batches) + 8, max(b[0]["ptr"].numel() - 1 for b in batches) + 1, max(runtime.pair_capacity(b[0]["ptr"].numpy()) for b in batches))
model = _model(torch.float32, **SMALL).train()
w = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 5.0}
g = runtime.GraphedStep(model, cap, compute_forces=False, warmup=0)
def cap_grads(dev, tgt, n, G):
    e_t, f_t = tgt[keys.TOTAL_ENERGY].float().to(DEV), tgt[keys.FORCES].float().to(DEV)
    g._load(dev["pos"].detach(), dev["atomic_numbers"], dev["ptr"], dev["batch"])
    rowptr, count = ops.radius_graph_capacity(g.pos, g.ptr, g.cutoff, g.edge_index)
    if os.environ.get("ZERO_TAIL"): g.edge_index[:, int(count):] = 0
    eg = ops.EdgeGraph(g.edge_index, g.n_atoms, center_sorted=True, ptr=g.ptr, c_rowptr=rowptr, symmetric=True)
    eg.edge_count_on_device = True
    data = {keys.POSITIONS: g.pos.detach().clone(), keys.ATOMIC_NUMBERS: g.z, keys.EDGE_INDEX: g.edge_index, keys.BATCH: g.batch, keys.BATCH_PTR: g.ptr, keys.EDGE_GRAPH: eg}
    model.zero_grad(set_to_none=True)
    out = model(data, True, False)
    l = ((out[keys.TOTAL_ENERGY][:G] - e_t) ** 2).mean() + 5.0 * ((out[keys.FORCES][:n] - f_t) ** 2).mean()
    l.backward()
    return l.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, int(count)
def exact_grads(dev, tgt):
    e_t, f_t = tgt[keys.TOTAL_ENERGY].float().to(DEV), tgt[keys.FORCES].float().to(DEV)
    b = NeighborTransform(5.0)(XequiBatch(dev["pos"], dev["atomic_numbers"], dev["ptr"]))
    model.zero_grad(set_to_none=True)
    l, _ = train.weighted_loss(model(b.to_dict(), True, False), {keys.TOTAL_ENERGY: e_t, keys.FORCES: f_t, keys.BATCH_PTR: dev["ptr"]}, w)
    l.backward()
    return l.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
for msg, node in ((True, True),):
    tr.NATIVE_MESSAGE, tr.NATIVE_NODE = msg, node
    for it, (host, dev, tgt) in (list(enumerate(batches))[::-1] if os.environ.get('REV') else enumerate(batches)):
        n, G = host["pos"].shape[0], host["ptr"].numel() - 1
        try:
            l1, g1, cnt = cap_grads(dev, tgt, n, G)
        except NotImplementedError as e:
            print(msg, node, "tensor form refuses"); break
        l0, g0 = exact_grads(dev, tgt)
        bb = NeighborTransform(5.0)(XequiBatch(dev["pos"], dev["atomic_numbers"], dev["ptr"]))
        print("exact E", bb.to_dict()["edge_index"].shape[1], "atoms", n, "of", g.n_atoms, "graphs", G, "of", g.n_graphs)
        rel = {k: float((g0[k] - g1[k]).abs().max() / g0[k].abs().max().clamp_min(1e-12)) for k in g0}
        print({k.replace("mods.", ""): f"{v:.1e}" for k, v in rel.items() if v > 1e-4})
        worst = max(((float((g0[k] - g1[k]).abs().max() / g0[k].abs().max().clamp_min(1e-12))), k) for k in g0)
        print("native message", msg, "node", node, "batch", it, "edges", cnt, "of", g.n_edges, "loss", l0, l1, "worst", worst)

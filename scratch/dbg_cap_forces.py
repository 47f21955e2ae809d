import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import keys, train, ops, runtime
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
dev = "cuda"
pos, z, ptr = syn.synth_qm9_batch(20, seed=40)
torch.manual_seed(0)
model = resolve_model("xpainn", node_dim=128, node_irreps="128x0e + 64x1o + 32x2e", action_blocks=2, hidden_dim=64).to(dev).train()
b = NeighborTransform(5.0)(XequiBatch(torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)))
ref = model(b.to_dict(), True, False)
n, G = len(pos), len(ptr) - 1
g = runtime.GraphedStep(model, (n + 8, G, runtime.pair_capacity(ptr)), compute_forces=False, warmup=0)
g._load(b.to_dict()["pos"].detach(), b.to_dict()["atomic_numbers"], b.to_dict()["ptr"], b.to_dict()["batch"])
rowptr, count = ops.radius_graph_capacity(g.pos, g.ptr, g.cutoff, g.edge_index)
print("edges", int(count), "capacity", g.n_edges, "ref E", b.to_dict()["edge_index"].shape[1])
for flag in (False, True):
    eg = ops.EdgeGraph(g.edge_index, g.n_atoms, center_sorted=True, ptr=g.ptr, c_rowptr=rowptr, symmetric=True)
    eg.edge_count_on_device = flag
    data = {keys.POSITIONS: g.pos.detach().clone(), keys.ATOMIC_NUMBERS: g.z, keys.EDGE_INDEX: g.edge_index, keys.BATCH: g.batch, keys.BATCH_PTR: g.ptr, keys.EDGE_GRAPH: eg}
    out = model(data, True, False)
    print(flag, "dE", float((out[keys.TOTAL_ENERGY][:G] - ref[keys.TOTAL_ENERGY]).abs().max()), "dF", float((out[keys.FORCES][:n] - ref[keys.FORCES]).abs().max()),
          "F pad", float(out[keys.FORCES][n:].abs().max()), "Fmax", float(ref[keys.FORCES].abs().max()))

import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_training import _batch, _targets, _model, SMALL, DEV
from xequinet_amd import keys, train, runtime, ops
from xequinet_amd.data import NeighborTransform, XequiBatch
host, dev = _batch(14, 41, torch.float32)
model = _model(torch.float32, **SMALL).train()
n, G = host["pos"].shape[0], host["ptr"].numel() - 1
b = NeighborTransform(5.0)(XequiBatch(dev["pos"], dev["atomic_numbers"], dev["ptr"])).to_dict()
for pads in (12, 13):
    g = runtime.GraphedStep(model, (n + pads, G + 1, runtime.pair_capacity(host["ptr"].numpy())), compute_forces=False, warmup=0)
    g._load(dev["pos"].detach(), dev["atomic_numbers"], dev["ptr"], dev["batch"])
    rowptr, count = ops.radius_graph_capacity(g.pos, g.ptr, g.cutoff, g.edge_index)
    c = int(count)
    eg = ops.EdgeGraph(g.edge_index, g.n_atoms, center_sorted=True, ptr=g.ptr, c_rowptr=rowptr, symmetric=True)
    same_list = torch.equal(g.edge_index[:, :c], b["edge_index"])
    ei = g.edge_index[:, :c]
    rev = eg.n_perm[:c].long()
    ok = bool((ei[0][rev] == ei[1]).all() and (ei[1][rev] == ei[0]).all())
    print("pads", pads, "count", c, "same list", same_list, "reverse map ok", ok, "rowptr tail", rowptr[n - 1:].tolist()[:6], "ptr tail", g.ptr[-3:].tolist(), "batch pad", g.batch[n:n + 3].tolist(), "pos pad", g.pos[n:n + 2].tolist())
    print("   n_rowptr is c_rowptr", eg.n_rowptr is eg.c_rowptr, "z pad", g.z[n:n+3].tolist())

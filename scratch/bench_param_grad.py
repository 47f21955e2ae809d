"""Time of the filter parameter-gradient kernels of the training pass on one message block of a QM9-shaped batch.
usage (GPU box): [XEQ_LIB_PATH=scratch/variants/libxeq_<v>.so] python scratch/bench_param_grad.py [n_mol]"""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import ops, lib
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch

n_mol = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda"
pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=1234)
b = NeighborTransform(5.0)(XequiBatch(torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)))
d = b.to_dict()
from xequinet_amd.nn.basic import edge_graph
graph = edge_graph(d)
ei = d["edge_index"]
N, E = len(pos), ei.shape[1]
mul, F, B, rc = (128, 64, 32), 128, 20, 5.0
C, D = 224, 480
H = F + 2 * C
torch.manual_seed(0)
h, xhat = torch.randn(N, H, device=dev), torch.randn(N * D, device=dev)
vec = (d["pos"][ei[0]] - d["pos"][ei[1]]).contiguous()
gs, gx = torch.randn(N, F, device=dev), torch.randn(N, D, device=dev)
W, bb = torch.randn(H, B, device=dev), torch.randn(H, device=dev)
p0 = (math.pi * torch.arange(1, B + 1, device=dev) / rc).float()
cfg = ("bessel", "cosine", B, rc, F, mul, 1)
saved = (h, xhat, vec, W, bb, p0, None, None, None)
ops.KERNEL_TIMER.reset(enabled=True)
for _ in range(10):
    out = ops.message_param_grad(saved, graph, cfg, gs, gx)
print(f"N={N} E={E} lib={os.environ.get('XEQ_LIB_PATH', 'in-tree')}", {k: f"{v['total_ms'] / v['launches'] * 1e3:.0f} us" for k, v in ops.KERNEL_TIMER.summary().items()})

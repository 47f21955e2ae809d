"""Per-launch times of the wq message kernels, general AND first-block forms (QM9-1024): python scratch/bench_wq2.py
(timing only: with -DXEQ_WQ_ONLY_L builds the results are wrong by construction)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import ops, lib
from xequinet_amd.data import NeighborTransform, XequiBatch, synthetic as syn
dev = "cuda"
pos, z, ptr, cell = syn.make_workload("qm9_1024", 1234)
b = NeighborTransform(5.0)(XequiBatch(torch.tensor(pos, dtype=torch.float32), torch.tensor(z), torch.tensor(ptr)).to(dev))
g = getattr(b, "_xeq_edge_graph")
N, E = g.n_nodes, g.n_edges
ei = b.edge_index
vec = (b.pos[ei[0]] - b.pos[ei[1]]).contiguous()
torch.manual_seed(0)
F_, mul = 128, (128, 64, 32); C, D, H, B = 224, 480, 576, 20
h = torch.randn(N, H, device=dev); xhat = torch.randn(N, D, device=dev); s = torch.randn(N, F_, device=dev); x = torch.randn(N, D, device=dev)
xhat0 = torch.zeros(N * D, device=dev); xhat0[: N * F_] = torch.randn(N * F_, device=dev)
W = torch.randn(H, B, device=dev) / B**0.5; bias = torch.randn(H, device=dev)
p0 = (torch.pi * torch.arange(1, B + 1, device=dev) / 5.0).float()
gs = torch.randn(N, F_, device=dev); gx = torch.randn(N, D, device=dev)
os.environ["XEQ_MESSAGE_IMPL"] = "wq"
def run(first):
    for p in (g._wq or {}).values(): p["records"] = None
    cfg = ("bessel", "cosine", B, 5.0, F_, mul) + ((1 | lib.XHAT_HIGHER_L_ZERO,) if first else ())
    vv = vec.clone().requires_grad_()
    if first:
        hh, xx = h, xhat0
    else:
        hh, xx = h.clone().requires_grad_(), xhat.clone().requires_grad_()
    so, xo = ops.FusedMessage.apply(hh, xx, vv, s, x, W, bias, p0, None, g, cfg)
    ((so * gs).sum() + (xo * gx).sum()).backward()
for first in (False, True):
    for _ in range(3): run(first)
    ops.KERNEL_TIMER.reset(True)
    for _ in range(10): run(first)
    r = ops.KERNEL_TIMER.summary(); ops.KERNEL_TIMER.reset(False)
    print(("first-block" if first else "general    "), os.environ.get("XEQ_LIB_PATH", "in-tree").split("/")[-1],
          {k.replace("xeq_message_", ""): f"{v['total_ms'] / v['launches'] * 1e3:.1f}" for k, v in r.items() if "message" in k})

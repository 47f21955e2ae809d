"""Micro-benchmark of the fused message kernels alone on the QM9-1024 workload."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd import ops
from xequinet_amd.data import NeighborTransform, XequiBatch
wl = sys.argv[1] if len(sys.argv) > 1 else "qm9"
dev = "cuda"
if wl == "qm9":
    pos, z, ptr = syn.synth_qm9_batch(1024, seed=1234); cell=None
else:
    pos, z, ptr, cell = syn.synth_water_box(8, seed=5)
b = XequiBatch(torch.tensor(pos, dtype=torch.float32), torch.tensor(z), torch.tensor(ptr),
               pbc=None if cell is None else torch.tensor([[True]*3]), cell=None if cell is None else torch.tensor(cell, dtype=torch.float32)).to(dev)
b = NeighborTransform(5.0)(b)
g = getattr(b, "_xeq_edge_graph")
N, E = g.n_nodes, g.n_edges
ei = b.edge_index
if cell is None:
    vec = (b.pos[ei[0]] - b.pos[ei[1]]).contiguous()
else:
    vec = (b.pos[ei[0]] - b.pos[ei[1]] - b.cell_offsets @ b.cell[0]).contiguous()
torch.manual_seed(0)
F_, mul = 128, (128, 64, 32); C, D, H, B = 224, 480, 576, 20
h = torch.randn(N, H, device=dev); xhat = torch.randn(N, D, device=dev); s = torch.randn(N, F_, device=dev); x = torch.randn(N, D, device=dev)
W = torch.randn(H, B, device=dev) / B**0.5; bias = torch.randn(H, device=dev)
p0 = (torch.pi * torch.arange(1, B + 1, device=dev) / 5.0).float()
gs = torch.randn(N, F_, device=dev); gx = torch.randn(N, D, device=dev)
cfg = ("bessel", "cosine", B, 5.0, F_, mul)
def run(impl):
    os.environ["XEQ_MESSAGE_IMPL"] = impl
    hh, xx, vv = h.clone().requires_grad_(), xhat.clone().requires_grad_(), vec.clone().requires_grad_()
    so, xo = ops.FusedMessage.apply(hh, xx, vv, s, x, W, bias, p0, None, g, cfg)
    ((so * gs).sum() + (xo * gx).sum()).backward()
    return so.detach(), xo.detach(), hh.grad, xx.grad, vv.grad
ref = run("valu"); got = run("sb")
for n, a, r in zip(["s_out", "x_out", "g_h", "g_xhat", "g_vec"], got, ref):
    print(f"{n:8s} max|diff| {float((a - r).abs().max()):.3e}  scale {float(r.abs().max()):.3e}")
def timeit(impl, reps=20):
    os.environ["XEQ_MESSAGE_IMPL"] = impl
    ops.KERNEL_TIMER.reset(True)
    for _ in range(reps):
        run(impl)
    r = ops.KERNEL_TIMER.summary(); ops.KERNEL_TIMER.reset(False)
    return {k: v["total_ms"] / v["launches"] * 1e3 for k, v in r.items()}
run("mfma"); run("valu"); run("sb")
print(f"N={N} E={E}")
for impl in ("sb", "sb"):
    print(impl, os.environ.get("XEQ_SB_FORM"), os.environ.get("XEQ_SB_GROUP"), {k: f"{v:.1f} us" for k, v in timeit(impl).items()})

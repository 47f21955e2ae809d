"""Micro-benchmark of the wm message kernels alone on the QM9-1024 workload (XEQ_WM_ABLATE ablations)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd import ops
from xequinet_amd.data import NeighborTransform, XequiBatch
dev = "cuda"
pos, z, ptr = syn.synth_qm9_batch(1024, seed=1234)
b = XequiBatch(torch.tensor(pos, dtype=torch.float32), torch.tensor(z), torch.tensor(ptr)).to(dev)
b = NeighborTransform(5.0)(b)
g = getattr(b, "_xeq_edge_graph")
N, E = g.n_nodes, g.n_edges
ei = b.edge_index
vec = (b.pos[ei[0]] - b.pos[ei[1]]).contiguous()
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
def timeit(impl, reps=10):
    ops.KERNEL_TIMER.reset(True)
    for _ in range(reps):
        run(impl)
    r = ops.KERNEL_TIMER.summary(); ops.KERNEL_TIMER.reset(False)
    return {k: v["total_ms"] / v["launches"] * 1e3 for k, v in r.items()}
ref = run("sb"); got = run("wm")
for n, a, r in zip(["s_out", "x_out", "g_h", "g_xhat", "g_vec"], got, ref):
    print(f"{n:8s} max|diff| {float((a - r).abs().max()):.3e}  scale {float(r.abs().max()):.3e}")
print(f"N={N} E={E}")
for ab in sys.argv[1:] or ["0"]:
    os.environ["XEQ_WM_ABLATE"] = ab
    run("wm")
    print("ablate", ab, "threads", os.environ.get("XEQ_WM_THREADS"), {k: f"{v:.1f} us" for k, v in timeit("wm").items()})
os.environ["XEQ_WM_ABLATE"] = "0"
print("sb", {k: f"{v:.1f} us" for k, v in timeit("sb").items()})
if os.environ.get("XEQ_WM_STAMPS"):
    import ctypes
    from xequinet_amd import lib
    L = lib.load()
    buf = (ctypes.c_ulonglong * 8)()
    L.xeq_wm_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    L.xeq_wm_debug_stamps(buf)            # clear
    run("wm"); torch.cuda.synchronize()
    L.xeq_wm_debug_stamps(buf)
    tot = sum(buf)
    names = ["tile top (issue loads)", "pass S MFMA issue", "pass S rows", "pass E MFMA issue", "pass E rows", "pass M MFMA issue", "pass M rows", "sums+table"]
    print("reverse kernel, l = 0 waves, cycles by phase (one backward launch):")
    for n, v in zip(names, buf): print(f"  {n:26s} {v/tot*100:5.1f} %   {v:.3e}")

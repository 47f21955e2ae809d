"""wm message kernels: launch time vs edges-per-stream for small and large systems."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd import ops
from xequinet_amd.data import NeighborTransform, XequiBatch
dev = "cuda"
F_, mul = 128, (128, 64, 32); C, D, H, B = 224, 480, 576, 20
def case(name, pos, z, ptr, cell=None):
    kw = {} if cell is None else dict(pbc=torch.tensor([[True, True, True]]), cell=torch.tensor(cell, dtype=torch.float32))
    b = XequiBatch(torch.tensor(pos, dtype=torch.float32), torch.tensor(z), torch.tensor(ptr), **kw).to(dev)
    b = NeighborTransform(5.0)(b)
    g = getattr(b, "_xeq_edge_graph"); N, E = g.n_nodes, g.n_edges
    torch.manual_seed(0)
    vec = torch.randn(E, 3, device=dev); vec = vec / vec.norm(dim=1, keepdim=True) * (1 + 3 * torch.rand(E, 1, device=dev))
    h = torch.randn(N, H, device=dev); xhat = torch.randn(N, D, device=dev); s = torch.randn(N, F_, device=dev); x = torch.randn(N, D, device=dev)
    W = torch.randn(H, B, device=dev) / B**0.5; bias = torch.randn(H, device=dev)
    p0 = (torch.pi * torch.arange(1, B + 1, device=dev) / 5.0).float()
    gs = torch.randn(N, F_, device=dev); gx = torch.randn(N, D, device=dev)
    cfg = ("bessel", "cosine", B, 5.0, F_, mul)
    def run():
        hh, xx, vv = h.clone().requires_grad_(), xhat.clone().requires_grad_(), vec.clone().requires_grad_()
        so, xo = ops.FusedMessage.apply(hh, xx, vv, s, x, W, bias, p0, None, g, cfg)
        ((so * gs).sum() + (xo * gx).sum()).backward()
    out = []
    for eps in sys.argv[1:] or ["16", "32", "64", "128"]:
        os.environ["XEQ_WM_EDGES_PER_STREAM"] = eps
        for _ in range(3): run()
        ops.KERNEL_TIMER.reset(True)
        for _ in range(10): run()
        r = ops.KERNEL_TIMER.summary(); ops.KERNEL_TIMER.reset(False)
        t = {k: v["total_ms"] / v["launches"] * 1e3 for k, v in r.items()}
        out.append(f"eps {eps}: fwd {t['xeq_message_fwd_wm']:.1f} bwd {t['xeq_message_bwd_wm']:.1f}")
    print(f"{name} N={N} E={E} | " + " | ".join(out), flush=True)
os.environ["XEQ_MESSAGE_IMPL"] = "wm"
if os.environ.get("SWEEP_ONLY_BIG"):
    pos, z, ptr = syn.synth_qm9_batch(1024, seed=1234); case("qm9-1024", pos, z, ptr)
    pos, z, ptr = syn.synth_qm9_batch(1024, seed=1235); case("qm9-1024b", pos, z, ptr)
    sys.exit(0)
pos, z, ptr = syn.synth_aspirin(); case("aspirin", pos, z, ptr)
pos, z, ptr, cell = syn.synth_water_box(4, seed=5); case("water-64", pos, z, ptr, cell)
pos, z, ptr = syn.synth_qm9_batch(64, seed=3); case("qm9-64", pos, z, ptr)
pos, z, ptr, cell = syn.synth_water_box(8, seed=5); case("water-512", pos, z, ptr, cell)
pos, z, ptr = syn.synth_qm9_batch(256, seed=3); case("qm9-256", pos, z, ptr)
pos, z, ptr = syn.synth_qm9_batch(1024, seed=1234); case("qm9-1024", pos, z, ptr)

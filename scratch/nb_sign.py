"""Is the node block's reverse launch odd in its cotangents bit for bit?  g -> -g should negate every output bit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_nodeblock as T
from xequinet_amd.nn import nodeblock
upd, msg = T._modules(21)
dev = T._dev()
upd, msg = upd.to(dev), msg.to(dev)
F, D, C = T.F, T.D, T.C
for n in (333, 7000):
    for mode in ("tail", "gx", "last"):
        tail = mode == "tail"
        torch.manual_seed(n)
        s = torch.randn(n, F, device=dev) * 1.5 + 0.2
        x = torch.randn(n, D, device=dev) * 0.8
        saved = nodeblock.node_block_fwd(s, x, upd, msg if tail else None, want_x=True)
        g_s_in = torch.randn(n, F, device=dev)
        g_x_in = torch.randn(n, D, device=dev) if mode != "last" else None
        g_h = torch.randn(n, F + 2 * C, device=dev) if tail else None
        g_xh = T._mulir_to_bt(torch.randn(n, D, device=dev)) if tail else None
        neg = lambda t: None if t is None else -t
        a = nodeblock.node_block_bwd(saved, s, x, upd, msg if tail else None, g_s_in, g_x_in, g_h, g_xh)
        b = nodeblock.node_block_bwd(saved, s, x, upd, msg if tail else None, neg(g_s_in), neg(g_x_in), neg(g_h), neg(g_xh))
        ds = (a[0] + b[0]).abs().max().item(); dx = (a[1] + b[1]).abs().max().item()
        print(n, mode, "g_s odd:", ds == 0.0, ds, " g_x odd:", dx == 0.0, dx, flush=True)

"""The node block at full size against itself on small slices (bitwise: a node's result does not depend on its batch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_nodeblock import _modules, F, D
from xequinet_amd.nn import nodeblock
dev = torch.device("cuda:0")
upd, msg = _modules(3); upd, msg = upd.to(dev), msg.to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 18609
torch.manual_seed(1)
s = torch.randn(n, F, device=dev); x = torch.randn(n, D, device=dev)
for rep in range(3):
    full = nodeblock.node_block_fwd(s, x, upd, msg)
    torch.cuda.synchronize()
    bad = 0
    for a in range(0, n, 256):
        b = min(n, a + 256)
        part = nodeblock.node_block_fwd(s[a:b].contiguous(), x[a:b].contiguous(), upd, msg)
        pairs = [(k, full[k][a:b], part[k]) for k in ("s_out", "x_out", "h2")]
        pairs += [(k, nodeblock.native_to_rows(full[k], n, wd)[a:b], nodeblock.native_to_rows(part[k], b - a, wd)) for k, wd in (("pre2", F), ("a", 480), ("ip", F), ("pre", F), ("uv", 960), ("p", 224))]
        for k, A_, B_ in pairs:
            if not torch.equal(A_, B_):
                d = (A_ - B_).abs()
                cols = torch.nonzero(d.amax(0) > 0).flatten()
                print(f"   {k}: cols {int(cols.min())}..{int(cols.max())} ({len(cols)} cols)")
                rows = torch.nonzero(d.amax(1) > 0).flatten()
                if bad < 5: print(f"rep {rep} slice {a}:{b} {k}: {len(rows)} rows differ, max {float(d.max()):.3e}, first rows {rows[:6].tolist()}, nan {bool(torch.isnan(full[k][a:b]).any())}")
                bad += 1
        if bad > 12: break
    print(f"rep {rep}: {bad} mismatching (slice, tensor) pairs of {3 * ((n + 255) // 256)}")

# ---- reverse launch, full size against slices
from tests.test_gpu_nodeblock import _mulir_to_bt, C
for mode in ("tail", "last"):
    tail = mode == "tail"
    m2 = msg if tail else None
    saved = nodeblock.node_block_fwd(s, x, upd, m2, want_x=True)
    g_s_in = torch.randn(n, F, device=dev)
    g_x_in = torch.randn(n, D, device=dev) if mode != "last" else None
    g_h = torch.randn(n, F + 2 * C, device=dev) if tail else None
    g_xh = torch.randn(n, D, device=dev) if tail else None
    g_s, g_x = nodeblock.node_block_bwd(saved, s, x, upd, m2, g_s_in, g_x_in, g_h, _mulir_to_bt(g_xh) if tail else None)
    torch.cuda.synchronize()
    bad = 0
    for a in range(0, n, 4096):
        b = min(n, a + 300)
        sp, xp = s[a:b].contiguous(), x[a:b].contiguous()
        sv = nodeblock.node_block_fwd(sp, xp, upd, m2, want_x=True)
        gs2, gx2 = nodeblock.node_block_bwd(sv, sp, xp, upd, m2, g_s_in[a:b].contiguous(), g_x_in[a:b].contiguous() if g_x_in is not None else None,
                                            g_h[a:b].contiguous() if tail else None, _mulir_to_bt(g_xh[a:b].contiguous()) if tail else None)
        for nm, A, B in (("g_s", g_s[a:b], gs2), ("g_x", g_x[a:b], gx2)):
            if not torch.equal(A, B):
                d = (A - B).abs(); rows = torch.nonzero(d.amax(1) > 0).flatten()
                if bad < 6: print(f"bwd {mode} slice {a}:{b} {nm}: {len(rows)} rows differ, max {float(d.max()):.3e}, first rows {rows[:6].tolist()} nan {bool(torch.isnan(A).any())}")
                bad += 1
    print(f"bwd {mode}: {bad} mismatching pairs")

"""Timing of xeq_update_uv_fwd against the kernel chain it replaces (through UpdateBlock's front half)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import lib
from xequinet_amd.lib import call, ptr, stream, mul3, dtype_code
from xequinet_amd.nn import fused
from xequinet_amd.nn.xpainn import XPainnUpdate

dev = "cuda"
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 18421
blk = XPainnUpdate().to(dev).eval().requires_grad_(False)
F, mul = 128, (128, 64, 32); C, D = 224, 480
s, x = torch.randn(n, F, device=dev), torch.randn(n, D, device=dev)
cat = torch.empty(n, F + C, device=dev); uv = torch.empty(2 * n * D, device=dev); p = torch.empty(n, C, device=dev); stats = torch.empty(n, 4, device=dev)
frag, has_bias, frag_t = fused._packed_uv_frag(blk)
packs, bias = fused._packed_uv(blk)
def fused_front():
    call("xeq_update_uv_fwd", ptr(s), ptr(x), ptr(blk.norm.weight), ptr(blk.norm.bias), ptr(blk.o3norm.affine_weight), ptr(blk.o3norm.affine_bias),
         n, F, mul3(mul), 1, ptr(frag[0]), ptr(frag[1]), ptr(frag[2]), int(has_bias), 1e-5, ptr(cat), F + C, ptr(p), ptr(uv), ptr(stats), stream())
def chain():
    _, xhat, st, _ = fused._norm_fwd(s, x, blk.norm, blk.o3norm, F, mul, shat_out=cat, ld=F + C)
    for (l, m, xb), (_, _, ub), W in zip(fused._bt_blocks(xhat, n, mul, 1), fused._bt_blocks(uv, n, mul, 2), packs):
        if l == 0 and bias is not None: torch.addmm(bias, xb, W, out=ub)
        else: torch.mm(xb, W, out=ub)
    call("xeq_uv_reduce_fwd", dtype_code(s), ptr(uv), n, mul3(mul), 1e-5, ptr(cat), F + C, F, ptr(p), stream())
def timeit(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(f"n={n}: fused front {timeit(fused_front):.1f} us   chain (norm + 3 GEMM + reduce) {timeit(chain):.1f} us")

# ---- reverse
g_p = torch.randn(n, C, device=dev); g_cat = torch.randn(n, F + C, device=dev); g_x_out = torch.randn(n, D, device=dev); g_s_out = torch.randn(n, F, device=dev)
a_t = torch.randn(n, C + 2 * F, device=dev)
fused_front(); torch.cuda.synchronize()
g_s, g_x = torch.empty_like(s), torch.empty_like(x); g_xhat = torch.empty(n * D, device=dev); g_uv = torch.empty_like(uv)
lw, ew = blk.norm.weight, blk.o3norm.affine_weight
def rev(fuse):
    call("xeq_update_uv_bwd", ptr(uv), ptr(g_p), ptr(g_cat), F + C, ptr(g_x_out), ptr(g_s_out), ptr(a_t), C + 2 * F, ptr(s), ptr(x), ptr(stats), ptr(lw), ptr(ew),
         n, F, mul3(mul), 1, ptr(frag_t[0]), ptr(frag_t[1]), ptr(frag_t[2]), 1e-5, ptr(g_s), ptr(g_x), None if fuse else ptr(g_xhat), stream())
    if not fuse:
        return fused._norm_bwd(s, x, blk.norm, blk.o3norm, stats, 1, F, mul, g_cat, F + C, g_xhat, g_s_out, g_x_out)
    return g_s, g_x
def rev_chain():
    call("xeq_uv_reduce_bwd", dtype_code(s), ptr(uv), ptr(g_p), ptr(g_cat), F + C, F, n, mul3(mul), 1e-5, ptr(g_x_out), ptr(a_t), ptr(g_uv), stream())
    for (l, m, gb), (_, _, gub), W in zip(fused._bt_blocks(g_xhat, n, mul, 1), fused._bt_blocks(g_uv, n, mul, 2), packs):
        torch.mm(gub, W.t(), out=gb)
    return fused._norm_bwd(s, x, blk.norm, blk.o3norm, stats, 1, F, mul, g_cat, F + C, g_xhat, g_s_out, g_x_out)
r_f = [t.clone() for t in rev(True)]; r_s = [t.clone() for t in rev(False)]; r_c = [t.clone() for t in rev_chain()]
print("   reverse: max |fused - chain|", (r_f[0] - r_c[0]).abs().max().item(), (r_f[1] - r_c[1]).abs().max().item(), " |split - chain|", (r_s[0] - r_c[0]).abs().max().item(), (r_s[1] - r_c[1]).abs().max().item())
print(f"   reverse: fused {timeit(lambda: rev(True)):.1f} us   split (+ norm_bwd) {timeit(lambda: rev(False)):.1f} us   chain {timeit(rev_chain):.1f} us")

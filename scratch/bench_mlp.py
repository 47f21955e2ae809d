"""Correctness + timing of xeq_mlp2_fwd / bwd against the library GEMM chain."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import lib
from xequinet_amd.lib import call, ptr, stream

dev = "cuda"
torch.manual_seed(0)

def pack(W, transposed, bias=None):
    if transposed:   # W viewed [k_in][n_out]
        k_in, n_out = W.shape
    else:
        n_out, k_in = W.shape
    out = torch.empty(lib.load().xeq_mlp_packed_floats(n_out, k_in), device=dev)
    call("xeq_mlp_pack", ptr(W), ptr(bias), n_out, k_in, int(transposed), ptr(out), stream())
    return out

def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

CASES = [(18421, 128, 576), (18421, 352, 480), (777, 128, 576), (31, 352, 480), (147456, 128, 576)]
if len(sys.argv) > 3:
    CASES = [tuple(int(v) for v in sys.argv[1:4])]
for (n, k1, n2) in CASES:
    X = torch.randn(n, k1, device=dev)
    W1 = torch.randn(128, k1, device=dev) / k1 ** 0.5
    b1 = torch.randn(128, device=dev)
    W2 = torch.randn(n2, 128, device=dev) / 128 ** 0.5
    b2 = torch.randn(n2, device=dev)
    W1p, W2p = pack(W1, False, b1), pack(W2, False, b2)
    W2tp = pack(W2, True)          # [k_in = n2][n_out = 128]
    W1tp = pack(W1, True)          # [k_in = 128][n_out = k1]
    pre = torch.empty(n, 128, device=dev); Y = torch.empty(n, n2, device=dev)
    def fused():
        call("xeq_mlp2_fwd", ptr(X), k1, n, k1, ptr(W1p), ptr(W2p), n2, ptr(pre), ptr(Y), n2, stream())
    def chain():
        p = torch.addmm(b1, X, W1.t()); return torch.addmm(b2, torch.nn.functional.silu(p), W2.t())
    fused(); torch.cuda.synchronize()
    Xd, W1d, W2d = X.double(), W1.double(), W2.double()
    pd = Xd @ W1d.t() + b1.double(); Yd = torch.nn.functional.silu(pd) @ W2d.t() + b2.double()
    Yc = chain()
    print(f"n={n} k1={k1} n2={n2}: fwd err fused {(Y.double()-Yd).abs().max():.2e} chain {(Yc.double()-Yd).abs().max():.2e}  pre err {(pre.double()-pd).abs().max():.2e}", end="  ")
    G = torch.randn(n, n2, device=dev); GX = torch.empty(n, k1, device=dev)
    def fused_b():
        call("xeq_mlp2_bwd", ptr(G), n2, n, n2, ptr(W2tp), ptr(pre), ptr(W1tp), k1, ptr(GX), k1, stream())
    def chain_b():
        return torch.mm(torch.ops.aten.silu_backward(torch.mm(G, W2), pre), W1)
    fused_b(); torch.cuda.synchronize()
    sig = torch.sigmoid(pd)
    GXd = ((G.double() @ W2d) * (sig * (1 + pd * (1 - sig)))) @ W1d
    GXc = chain_b()
    print(f"bwd err fused {(GX.double()-GXd).abs().max():.2e} chain {(GXc.double()-GXd).abs().max():.2e}")
    import ctypes
    L = lib.load()
    if os.environ.get("XEQ_LIB_PATH"):
        buf = (ctypes.c_ulonglong * 8)()
        L.xeq_mlp_debug_stamps(buf)
        for _ in range(20): fused()
        torch.cuda.synchronize()
        L.xeq_mlp_debug_stamps(buf)
        if buf[2]:
            print(f"    stamps: {buf[0]/buf[2]:.0f} core cycles / workgroup, {buf[1]/buf[2]*10:.0f} ns / workgroup, clock {buf[0]/max(buf[1],1)*0.1:.2f} GHz; wave 0: prologue {buf[3]/buf[2]:.0f}, stage-1 loop {buf[4]/buf[2]:.0f}, hidden epilogue {buf[5]/buf[2]:.0f}, stage 2 {buf[7]/buf[2]:.0f} (MFMA quarters {buf[6]/buf[2]:.0f})")
    if os.environ.get("XEQ_LIB_PATH") and os.environ.get("XEQ_MLP_WG"):
        import collections
        fused(); torch.cuda.synchronize()
        wg = (ctypes.c_ulonglong * (4096 * 4))()
        L.xeq_mlp_debug_wg(wg)
        nwg = min(4096, (n + 31) // 32)
        recs = [(wg[4*b], wg[4*b+1] & 0xf, wg[4*b+2], wg[4*b+3]) for b in range(nwg)]
        t0 = min(r[2] for r in recs)
        percu = collections.defaultdict(list)
        for hw, xcc, a_, b_ in recs:
            cu = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
            percu[cu].append(((a_ - t0) * 10, (b_ - t0) * 10))
        cnt = collections.Counter(len(v) for v in percu.values())
        print(f"    {nwg} workgroups on {len(percu)} CUs; workgroups per CU histogram {sorted(cnt.items())}; span {max(r[3] for r in recs) - t0} x10 ns")
        for cu, v in list(sorted(percu.items(), key=lambda kv: -len(kv[1])))[:3]:
            print("      ", cu, sorted(v))
    print(f"    fwd fused {timeit(fused):.1f} us  chain {timeit(chain):.1f} us   bwd fused {timeit(fused_b):.1f} us  chain {timeit(chain_b):.1f} us")

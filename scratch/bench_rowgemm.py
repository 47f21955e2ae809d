"""Row-parallel products of the training pass: the library (torch.mm / addmm) against xeq_mlp_pack + xeq_linear_fwd, device time."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import lib
from xequinet_amd.lib import call, ptr, stream
dev = "cuda"
N = 18609
def t_us(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
L = lib.load()
for rows, k, m in ((N, 128, 128), (N, 128, 576), (N, 352, 128), (N, 128, 480), (N, 224, 128), (N, 576, 128), (N, 480, 128), (N, 128, 352), (N, 128, 256), (3 * N, 64, 128), (5 * N, 32, 64), (3 * N, 128, 64), (5 * N, 64, 32)):
    x = torch.randn(rows, k, device=dev); W = torch.randn(m, k, device=dev)
    lib_us = t_us(lambda: torch.mm(x, W.t()))
    if L.xeq_linear_supported(lib.XEQ_F32, k, m):
        pack = torch.empty(L.xeq_mlp_packed_floats(m, k), dtype=torch.float32, device=dev)
        y = torch.empty(rows, m, device=dev)
        def own():
            call("xeq_mlp_pack", ptr(W), None, m, k, 0, ptr(pack), stream())
            call("xeq_linear_fwd", ptr(x), k, rows, k, None, ptr(pack), m, 0, 0, None, ptr(y), m, stream())
        own_us = t_us(own)
        err = float((y - x @ W.t()).abs().max() / (x @ W.t()).abs().max())
        print(f"[{rows} x {k}] x [{k} x {m}]: library {lib_us:7.1f} us ({2e-6 * rows * k * m / lib_us:5.1f} TF/s)   pack + xeq_linear_fwd {own_us:7.1f} us ({2e-6 * rows * k * m / own_us:5.1f} TF/s)  rel diff {err:.1e}")
    else:
        print(f"[{rows} x {k}] x [{k} x {m}]: library {lib_us:7.1f} us ({2e-6 * rows * k * m / lib_us:5.1f} TF/s)   xeq_linear_fwd does not take it")

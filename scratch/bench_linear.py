"""Per-launch time of xeq_linear_fwd on the model's five shapes against the library GEMM (QM9-1024 row count): python scratch/bench_linear.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.nn import fused
N = int(sys.argv[1]) if len(sys.argv) > 1 else 18609
dev = "cuda"
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, k, n, bias in (("dot_lin fwd", 224, 128, False), ("dot_lin bwd", 128, 224, False), ("embedding", 56, 128, True), ("head 1", 128, 64, True), ("head 1 bwd", 64, 128, False)):
    lin = torch.nn.Linear(k, n, bias=bias).to(dev).requires_grad_(False)
    x = torch.randn(N, k, device=dev)
    t_own = timeit(lambda: fused.linear_module_fwd(lin, x))
    t_lib = timeit(lambda: torch.nn.functional.linear(x, lin.weight, lin.bias))
    print(f"{name:12s} K={k:3d} n_out={n:3d}: xeq_linear {t_own:6.1f} us   library {t_lib:6.1f} us")

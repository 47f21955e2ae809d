"""Per-launch time of xeq_linear_fwd on the shapes of one evaluation (18 609 rows).  usage: [XEQ_LIB_PATH=...] python scratch/bench_linear.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.nn import fused
n = 18609
out = []
for k_in, n_out, act in ((224, 128, 0), (128, 224, 0), (128, 64, 1), (56, 128, 0)):
    lin = torch.nn.Linear(k_in, n_out).cuda()
    x = torch.randn(n, k_in, device="cuda")
    pack = fused._linear_pack(lin, lin.weight, lin.bias, False)
    f = lambda: fused._linear(x, pack, k_in, n_out, True, act=act)
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): f()
    torch.cuda.synchronize()
    out.append(f"{k_in}->{n_out}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
print(os.environ.get("XEQ_LIB_PATH", "in-tree"), " | ".join(out))

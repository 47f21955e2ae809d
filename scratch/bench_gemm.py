"""The model's dense contractions: the experimental xeq_gemm_f32 (scratch/xeq_gemm_lds_experiment.hip, per workgroup tile;
the earlier LDS-free form is scratch/xeq_gemm_experiment.hip) against the library GEMM (TunableOp picks).  Shelved: needs the
kernel in csrc/build.py, its prototype in lib.py / xeq.h and an ops.gemm wrapper; results of the last runs:
scratch/bench_gemm_lds_result.txt, scratch/bench_gemm_result.txt."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import ops
from xequinet_amd.tuning import enable_gemm_autotune
enable_gemm_autotune()
dev = "cuda"; N0 = int(sys.argv[1]) if len(sys.argv) > 1 else 18609
SHAPES = [("lin1", 1, 128, 128, "nk", True, "silu"), ("lin2", 1, 128, 576, "nk", True, None), ("lin2^T", 1, 576, 128, "kn", False, None),
          ("lin1^T", 1, 128, 128, "kn", False, None), ("UV0", 1, 128, 256, "kn", True, None), ("UV1", 3, 64, 128, "kn", False, None),
          ("UV2", 5, 32, 64, "kn", False, None), ("UV0^T", 1, 256, 128, "nk", False, None), ("UV1^T", 3, 128, 64, "nk", False, None),
          ("UV2^T", 5, 64, 32, "nk", False, None), ("lin3", 1, 352, 128, "nk", True, "silu"), ("lin4", 1, 128, 480, "nk", True, None),
          ("dot", 1, 224, 128, "nk", False, None), ("dot^T", 1, 128, 224, "kn", False, None), ("lin4^T", 1, 480, 128, "kn", False, None),
          ("lin3^T", 1, 128, 352, "kn", False, None)]
def t(fn, n=30):
    for _ in range(5): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
tot_lib = tot_best = 0.0
for name, mult, K, N, lay, has_b, act in SHAPES:
    M = N0 * mult
    x = torch.randn(M, K, device=dev); w = torch.randn((N, K) if lay == "nk" else (K, N), device=dev) / K**0.5
    b = torch.randn(N, device=dev) if has_b else None
    wt = w.t() if lay == "nk" else w
    def lib_fn():
        z = torch.mm(x, wt) if b is None else torch.addmm(b, x, wt)
        return torch.nn.functional.silu(z) if act else z
    ref = (x.double() @ wt.double() + (b.double() if has_b else 0))
    refo = torch.nn.functional.silu(ref) if act else ref
    tl = t(lib_fn)
    res = []
    for tile in (1, 2, 3, 4, 5, 6, 7, 0):
        fn = lambda: ops.gemm(x, w, b, lay, act, tile=tile)
        o = fn(); o = o[0] if act else o
        err = float((o.double() - refo).abs().max())
        res.append((t(fn), tile, err))
    eb = float((lib_fn().double() - refo).abs().max())
    best = min(res[:-1]); auto = res[-1]
    tot_lib += tl; tot_best += best[0]
    print(f"{name:7s} M={M:6d} K={K:3d} N={N:3d} {lay} | lib {tl:6.1f} us (err {eb:.1e}) | " + " ".join(f"{tl_:.0f}@{ti}" for tl_, ti, _ in res[:-1]) +
          f" | auto {auto[0]:.1f} | best {best[0]:.1f}@{best[1]} err {best[2]:.1e}", flush=True)
print(f"sum lib {tot_lib:.0f} us, sum best {tot_best:.0f} us")

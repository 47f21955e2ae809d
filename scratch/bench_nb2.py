"""Per-launch times of the fused node-block kernels, forward and reverse, tail / no tail (N = 18 609): python scratch/bench_nb2.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_nodeblock import _modules, F, D
from xequinet_amd.nn import nodeblock
dev = torch.device("cuda:0")
upd, msg = _modules(1)
upd, msg = upd.to(dev), msg.to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 18609
C = 224
s = torch.randn(n, F, device=dev); x = torch.randn(n, D, device=dev)
gs = torch.randn(n, F, device=dev); gx = torch.randn(n, D, device=dev); gh = torch.randn(n, F + 2 * C, device=dev); gxh = torch.randn(n * D, device=dev)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
sv_t = nodeblock.node_block_fwd(s, x, upd, msg)
sv_n = nodeblock.node_block_fwd(s, x, upd, None)
res = {
    "fwd<tail>": t(lambda: nodeblock.node_block_fwd(s, x, upd, msg)),
    "fwd<last>": t(lambda: nodeblock.node_block_fwd(s, x, upd, None)),
    "bwd<tail,gx>": t(lambda: nodeblock.node_block_bwd(sv_t, s, x, upd, msg, gs, gx, gh, gxh)),
    "bwd<last,nogx>": t(lambda: nodeblock.node_block_bwd(sv_n, s, x, upd, None, gs, None)),
}
print(f"n={n}", os.environ.get("XEQ_LIB_PATH", "in-tree").split("/")[-1], {k: f"{v:.1f}" for k, v in res.items()}, "us per call (incl. allocations)")

"""Per-launch time of the fused node-block kernels over node counts around one workgroup per CU (256 x 64 = 16 384 nodes):
python scratch/bench_nodeblock_sizes.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_nodeblock import _modules, F, D
from xequinet_amd.nn import nodeblock

dev = torch.device("cuda:0")
upd, msg = _modules(1)
upd, msg = upd.to(dev), msg.to(dev)
for n in (8192, 12288, 16384, 17408, 18609, 20480, 24576, 32768, 36864):
    s = torch.randn(n, F, device=dev); x = torch.randn(n, D, device=dev)
    out = []
    for tail in (True, False):
        for _ in range(3):
            o = nodeblock.node_block_fwd(s, x, upd, msg if tail else None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            o = nodeblock.node_block_fwd(s, x, upd, msg if tail else None)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"n = {n:6d} ({(n + 63) // 64:4d} workgroups): forward with tail {out[0]:7.1f} us, without {out[1]:7.1f} us   ({n / out[0]:.0f} nodes / us)")

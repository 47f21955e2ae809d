"""Time the fused node-block forward kernel at QM9-1024 size (N = 18 609) and a few other sizes."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_nodeblock import _modules, F, D
from xequinet_amd.nn import nodeblock

dev = torch.device("cuda:0")
upd, msg = _modules(1)
upd, msg = upd.to(dev), msg.to(dev)
import sys as _s
for n in ((1536,) if 'small' in _s.argv else (18609,) if 'quick' in _s.argv else (18609, 147410, 1536, 4096)):
    s = torch.randn(n, F, device=dev); x = torch.randn(n, D, device=dev)
    for tail in ((True,) if ('quick' in _s.argv or 'small' in _s.argv) else (True, False)):
        for _ in range(3):
            nodeblock.node_block_fwd(s, x, upd, msg if tail else None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            nodeblock.node_block_fwd(s, x, upd, msg if tail else None)
        e1.record(); torch.cuda.synchronize()
        print(f"n={n:7d} tail={tail}: {e0.elapsed_time(e1) / reps * 1e3:8.1f} us per launch (incl. allocs)")

import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.lib import call, ptr, stream
dev = torch.device("cuda:0")
form = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, n_ot = 18609, 18
x = torch.randn(n, 128, device=dev); w = torch.randn(32 * n_ot, 128, device=dev)
scratch = torch.empty((8 * n_ot + 32) * 3072, dtype=torch.uint8, device=dev); y = torch.empty(n, 32 * n_ot, device=dev)
for _ in range(3): call("xeq_node_block_linear_test", ptr(x), n, ptr(w), n_ot, form, ptr(scratch), ptr(y), stream())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): call("xeq_node_block_linear_test", ptr(x), n, ptr(w), n_ot, form, ptr(scratch), ptr(y), stream())
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"{us:.1f} us per (pack + 144-tile GEMM) launch pair -> ~{us * 2.2e3 / 144:.0f} cycles per tile at 2.2 GHz")
import ctypes, numpy as np
from xequinet_amd import lib
h = lib.load()
if hasattr(h, "xeq_node_block_debug_stamps"):
    h.xeq_node_block_debug_stamps.argtypes = [ctypes.c_void_p]
    buf = np.zeros(1024 * 4 * 24, dtype=np.uint64); h.xeq_node_block_debug_stamps(buf.ctypes.data)
    st = buf.reshape(1024, 4, 24)[: (n + 127) // 128].astype(np.int64)
    loop = st[:, :, 2] - st[:, :, 1]; tot = st[:, :, 2] - st[:, :, 0]; real = (st[:, :, 5] - st[:, :, 4]) * 10.0  # ns (100 MHz)
    print(f"loop cycles median {np.median(loop):.0f} ({np.median(loop)/144:.0f} per tile), total {np.median(tot):.0f} cycles in {np.median(real):.0f} ns -> clock {np.median(tot)/np.median(real):.2f} GHz")

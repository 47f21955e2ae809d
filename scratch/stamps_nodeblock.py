"""Phase timeline of the fused node-block forward kernel (XEQ_NB_STAMPS build): XEQ_LIB_PATH=scratch/variants/libxeq_nbst.so"""
import ctypes, sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_nodeblock import _modules, F, D
from xequinet_amd.nn import nodeblock
from xequinet_amd import lib
dev = torch.device("cuda:0")
upd, msg = _modules(1); upd, msg = upd.to(dev), msg.to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 18609
s = torch.randn(n, F, device=dev); x = torch.randn(n, D, device=dev)
for _ in range(3): nodeblock.node_block_fwd(s, x, upd, msg)
torch.cuda.synchronize()
h = lib.load(); h.xeq_node_block_debug_stamps.argtypes = [ctypes.c_void_p]
buf = np.zeros(1024 * 4 * 24, dtype=np.uint64)
h.xeq_node_block_debug_stamps(buf.ctypes.data)
st = buf.reshape(1024, 4, 24)[: (n + 63) // 64].astype(np.int64)
names = ["init", "LN+L1c0", "eq stats", "l=0", "l=1", "l=2", "hidden", "a_vv+dx", "scalar", "tailLN+sL1", "eqln2", "hid2", "sL2"]
d = np.diff(st[:, :, :14], axis=2)
print("workgroups", st.shape[0], "total cycles median", np.median(st[:, :, 13] - st[:, :, 0]))
for i, nm in enumerate(names):
    print(f"{nm:12s} median {np.median(d[:, :, i]):9.0f}  min {d[:, :, i].min():9.0f} max {d[:, :, i].max():9.0f}")
t0 = st[:, :, 0].min()
print("start spread", (st[:, :, 0].max() - t0), "end spread", st[:, :, 13].max() - t0, st[:, :, 13].min() - t0)
print("per wave: cycles waiting for the fetched stage + its LDS write", np.median(st[:, :, 14]), " at the stage barrier", np.median(st[:, :, 15]), "of", np.median(st[:, :, 13] - st[:, :, 0]))

"""Which operand of the l = 2 residual is wrong in the rows that differ?  (16-node form, XEQ_NB_DBG build)"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_nodeblock import _modules, F, D
from xequinet_amd.nn import nodeblock
from xequinet_amd import lib
dev = torch.device("cuda:0")
upd, msg = _modules(3); upd, msg = upd.to(dev), msg.to(dev)
n = 86016
torch.manual_seed(1)
s = torch.randn(n, F, device=dev); x = torch.randn(n, D, device=dev)
h = lib.load(); h.xeq_node_block_debug_buffer.argtypes = [ctypes.c_void_p]
nblk = (n + 63) // 64 * 4
dbg_full = torch.zeros(nblk, 11, 2, 64, 4, device=dev)
dbg_part = torch.zeros(32, 11, 2, 64, 4, device=dev)
names = [f"U[{m}]" for m in range(5)] + [f"x_in[{m}]" for m in range(5)] + ["a_vv"]
found = 0
for rep in range(6):
    h.xeq_node_block_debug_buffer(ctypes.c_void_p(dbg_full.data_ptr()))
    full = nodeblock.node_block_fwd(s, x, upd, msg)
    torch.cuda.synchronize()
    for a in range(0, n, 256):
        b = a + 256
        h.xeq_node_block_debug_buffer(ctypes.c_void_p(dbg_part.data_ptr()))
        part = nodeblock.node_block_fwd(s[a:b].contiguous(), x[a:b].contiguous(), upd, msg)
        torch.cuda.synchronize()
        if not torch.equal(full["x_out"][a:b], part["x_out"]):
            d = (full["x_out"][a:b] - part["x_out"]).abs()
            rows = torch.nonzero(d.amax(1) > 0).flatten(); cols = torch.nonzero(d.amax(0) > 0).flatten().tolist()
            blk = a // 16 + int(rows[0]) // 16
            diff = (dbg_full[blk] != dbg_part[(blk - a // 16)])
            which = [names[i] for i in range(11) if bool(diff[i].any())]
            where = {names[i]: (torch.nonzero(diff[i].any(-1).any(0)).flatten().tolist()[:20], torch.nonzero(diff[i].any(1).any(0)).flatten().tolist()) for i in range(11) if bool(diff[i].any())}
            print(f"rep {rep} slice {a}: x_out cols {cols}; block {blk}: operands that differ: {which}; (lanes, element) {where}")
            found += 1
            if found >= 8: sys.exit(0)
print("mismatches found:", found)

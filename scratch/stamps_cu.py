"""Do two workgroups on one CU overlap?  Per workgroup: duration and the CU it ran on (XEQ_NB_STAMPS build)."""
import ctypes, sys, os, numpy as np, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
nwg = min(1024, (n + 63) // 64)
st = buf.reshape(1024, 4, 24)[:nwg].astype(np.int64)
dur = (st[:, :, 13] - st[:, :, 0]).max(1)
start, end = st[:, :, 0].min(1), st[:, :, 13].max(1)
hw = st[:, 0, 16]
xcc, hwid = hw >> 32, hw & 0xffffffff
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
key = [(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(xcc, se, sh, cu)]
groups = collections.defaultdict(list)
for i, k in enumerate(key): groups[k].append(i)
print(f"{nwg} workgroups on {len(groups)} distinct (xcc, se, sh, cu); workgroups per CU: {collections.Counter(len(v) for v in groups.values())}")
alone = [dur[v[0]] for v in groups.values() if len(v) == 1]
print(f"alone on their CU: {len(alone)} workgroups, median {np.median(alone) if alone else 0:.0f} cycles")
for k, v in list(groups.items()):
    if len(v) >= 2:
        ov = [(i, j) for i in v for j in v if i < j and start[j] < end[i] and start[i] < end[j]]
        if ov:
            i, j = ov[0]
            print(f"CU {k}: workgroups {v}: durations {[int(dur[q]) for q in v]}, overlap of {i},{j}: {int(min(end[i], end[j]) - max(start[i], start[j]))} cycles")
            break
shared = [dur[i] for v in groups.values() if len(v) >= 2 for i in v]
print(f"on shared CUs: {len(shared)} workgroups, median duration {np.median(shared) if shared else 0:.0f} cycles")
names = ["init", "LN+L1c0", "eq stats", "l=0", "l=1", "l=2", "hidden", "a_vv+dx", "scalar", "tailLN+sL1", "eqln2", "hid2", "sL2"]
d = np.diff(st[:, :, :14], axis=2).max(1)
isal = np.array([len(groups[k]) == 1 for k in key])
for i, nm in enumerate(names):
    a = np.median(d[isal, i]) if isal.any() else 0; b = np.median(d[~isal, i]) if (~isal).any() else 0
    print(f"{nm:12s} alone {a:9.0f}   sharing {b:9.0f}   x {b / a if a else 0:.2f}")
rt = (st[:, :, 18] - st[:, :, 17]).max(1) * 10.0   # ns (100 MHz)
print(f"core clock seen by the workgroups: median {np.median(dur / rt):.3f} GHz (min {np.min(dur / rt):.3f}, max {np.max(dur / rt):.3f}); median duration {np.median(rt) / 1e3:.1f} us")

"""Micro-benchmark of the fused message kernels alone (QM9-1024 by default): python scratch/bench_wq.py [workload] impl [impl ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import ops
from xequinet_amd.data import NeighborTransform, XequiBatch, synthetic as syn
dev = "cuda"
args = sys.argv[1:]
wl = args.pop(0) if args and (args[0] in syn.WORKLOADS or args[0].startswith("qm9_")) else "qm9_1024"
pos, z, ptr, cell = syn.make_workload(wl, 1234)
kw = {} if cell is None else dict(pbc=torch.tensor([[True, True, True]], device=dev), cell=torch.tensor(cell, dtype=torch.float32, device=dev))
b = XequiBatch(torch.tensor(pos, dtype=torch.float32), torch.tensor(z), torch.tensor(ptr)).to(dev)
for k, v in kw.items(): setattr(b, k, v.reshape(-1, 3) if k == "pbc" else v.reshape(-1, 3, 3))
b = NeighborTransform(5.0)(b)
g = getattr(b, "_xeq_edge_graph")
N, E = g.n_nodes, g.n_edges
ei = b.edge_index
vec = (b.pos[ei[0]] - b.pos[ei[1]]).contiguous()
if cell is not None:
    vec = vec - b.cell_offsets @ b.cell[0]
torch.manual_seed(0)
F_, mul = 128, (128, 64, 32); C, D, H, B = 224, 480, 576, 20
h = torch.randn(N, H, device=dev); xhat = torch.randn(N, D, device=dev); s = torch.randn(N, F_, device=dev); x = torch.randn(N, D, device=dev)
W = torch.randn(H, B, device=dev) / B**0.5; bias = torch.randn(H, device=dev)
p0 = (torch.pi * torch.arange(1, B + 1, device=dev) / 5.0).float()
gs = torch.randn(N, F_, device=dev); gx = torch.randn(N, D, device=dev)
cfg = ("bessel", "cosine", B, 5.0, F_, mul)
def run(impl):
    os.environ["XEQ_MESSAGE_IMPL"] = impl
    g._basis = None
    for p in (g._wq or {}).values(): p["records"] = None
    hh, xx, vv = h.clone().requires_grad_(), xhat.clone().requires_grad_(), vec.clone().requires_grad_()
    so, xo = ops.FusedMessage.apply(hh, xx, vv, s, x, W, bias, p0, None, g, cfg)
    ((so * gs).sum() + (xo * gx).sum()).backward()
    return so.detach(), xo.detach(), hh.grad, xx.grad, vv.grad
def timeit(impl, reps=10):
    ops.KERNEL_TIMER.reset(True)
    for _ in range(reps):
        run(impl)
    r = ops.KERNEL_TIMER.summary(); ops.KERNEL_TIMER.reset(False)
    return {k: v["total_ms"] / v["launches"] * 1e3 for k, v in r.items()}
print(f"{wl}: N={N} E={E} lib={os.environ.get('XEQ_LIB_PATH', 'in-tree')}")
ref = run("sb")
for impl in args or ["wq"]:
    got = run(impl)
    again = run(impl)
    errs = []
    for n, a, r, a2 in zip(["s_out", "x_out", "g_h", "g_xhat", "g_vec"], got, ref, again):
        errs.append(f"{n} {float((a - r).abs().max()) / max(1.0, float(r.abs().max())):.1e}{'' if torch.equal(a, a2) else ' NOT-REPRODUCIBLE'}")
    run(impl)
    print(impl, {k: f"{v:.1f} us" for k, v in timeit(impl).items()}, "| rel err vs sb:", ", ".join(errs))
if os.environ.get("XEQ_WQ_STAMPS"):
    import ctypes
    from xequinet_amd import lib
    L = lib.load()
    buf = (ctypes.c_ulonglong * 32)(); buf2 = (ctypes.c_ulonglong * 32)()
    L.xeq_wq_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    L.xeq_wq_debug_stamps_bwd.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]   # (each half of the source has its own counter array)
    L.xeq_wq_debug_stamps(buf); L.xeq_wq_debug_stamps_bwd(buf2)            # clear
    run("wq"); torch.cuda.synchronize()
    L.xeq_wq_debug_stamps(buf); L.xeq_wq_debug_stamps_bwd(buf2)
    for i in range(16, 32): buf[i] = buf2[i]
    fn = ["step head", "window staging", "barrier after staging", "body prologue", "wait for record", "phase A issue", "phase B issue",
          "MFMA issue", "phase D rows+stores", "publish next table", "barrier after step"]
    bn = ["step head", "window staging", "barrier after staging", "body prologue", "wait for records", "tile top (gathers issued)",
          "MFMA issue (3 passes)", "rows pass S", "rows pass E (+dY sums)", "rows pass M", "barrier after step", "dL/dd channel sums",
          "publish next table"]
    for title, names, off, cnt in (("forward kernel, l = 0 waves", fn, 0, buf[15]), ("reverse kernel, waves of one l", bn, 16, buf[31])):
        vals = [buf[off + i] for i in range(len(names))]
        tot = sum(vals)
        print(f"{title} ({cnt}): cycles by phase, one launch; {tot / max(cnt, 1):.0f} cycles per wave")
        for n, v in zip(names, vals): print(f"  {n:28s} {v / max(tot, 1) * 100:5.1f} %   {v / max(cnt, 1):9.0f} per wave")
        if off == 0 and buf[14]: print(f"  tiles {buf[14]}: {buf[14] / max(cnt, 1):.1f} per wave; LITE build: slot 'phase D' = whole tile loops = {buf[8] / buf[14]:.0f} cycles per tile, 'body prologue' = range prologue")

if os.environ.get("XEQ_WQ_ROLE_TIME"):   # workgroup timeline of the reverse kernel (a -DXEQ_WQ_ROLE_TIME build)
    import collections, ctypes
    from xequinet_amd import lib as _lib
    L = _lib.load()
    wg = (ctypes.c_ulonglong * (8192 * 4))()
    dbg = getattr(L, "xeq_wq_debug_wg" if os.environ["XEQ_WQ_ROLE_TIME"] == "fwd" else "xeq_wq_debug_wg_bwd")   # (each half of the file has its own array)
    dbg(wg)
    run("wq"); torch.cuda.synchronize()
    dbg(wg)
    recs = [(wg[4*b] & 0xffffffff, wg[4*b] >> 32, wg[4*b+1] & 0xf, wg[4*b+2], wg[4*b+3]) for b in range(8192) if wg[4*b+3]]
    t0 = min(r[3] for r in recs); t1 = max(r[4] for r in recs)
    percu = collections.defaultdict(list)
    for hw, l, xcc, a_, b_ in recs:
        percu[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)].append(((a_ - t0) / 100, (b_ - t0) / 100, int(l)))
    busy = sum(b_ - a_ for v in percu.values() for a_, b_, _ in v)
    print(f"reverse kernel (last launch): {len(recs)} workgroups on {len(percu)} CUs, span {(t1 - t0) / 100:.1f} us, workgroup-time / (span x 2 slots x CUs) = {busy / ((t1 - t0) / 100 * 2 * len(percu)):.2f}")
    ends = sorted(max(b_ for _, b_, _ in v) for v in percu.values())
    print(f"   CU finish times (us): min {ends[0]:.0f}, 10 % {ends[len(ends)//10]:.0f}, median {ends[len(ends)//2]:.0f}, 90 % {ends[9*len(ends)//10]:.0f}, max {ends[-1]:.0f}; workgroups per CU: {sorted(collections.Counter(len(v) for v in percu.values()).items())}")
    dur = collections.defaultdict(list)
    for v in percu.values():
        for a_, b_, l in v: dur[l].append(b_ - a_)
    print("   workgroup durations (us) by l:", {l: (round(min(d)), round(sorted(d)[len(d)//2]), round(max(d))) for l, d in sorted(dur.items())})
    byx = collections.defaultdict(list)
    for cu, v in percu.items(): byx[cu[0]].append(max(b_ for _, b_, _ in v))
    print("   latest finish per XCD:", {x: round(max(v)) for x, v in sorted(byx.items())})
    for cu, v in list(sorted(percu.items(), key=lambda kv: -max(b_ for _, b_, _ in kv[1])))[:2]:
        print("   latest CU", cu, [(round(a_), round(b_), l) for a_, b_, l in sorted(v)])

if os.environ.get("XEQ_WQ_STEP_TIME"):   # per-step timeline of the workgroups (a -DXEQ_WQ_STEP_TIME (forward) or -DXEQ_WQ_STEP_TIME_BWD build)
    import ctypes
    from xequinet_amd import lib as _lib
    L = _lib.load()
    buf = (ctypes.c_uint * (8192 * 32))()
    L.xeq_wq_debug_steps(buf)
    run("wq"); torch.cuda.synchronize()
    L.xeq_wq_debug_steps(buf)
    raw = np.frombuffer(buf, dtype=np.uint32).reshape(8192, 4, 8)
    lval = (raw[:, :, 7] >> 30).astype(int)
    a = raw.astype(np.float64)
    a[:, :, 7] = (raw[:, :, 7] & 0x3fffffff)
    a /= 100.0   # us
    ok = a[:, :, 7] > 0
    med = lambda x: float(np.median(x))
    for l in range(3):
        for st in range(3):
            m = ok[:, st] & (lval[:, st] == l)
            if not m.any(): continue
            r = a[m, st]
            stage, bar1, body, bar2 = r[:, 1] - r[:, 0], r[:, 2] - r[:, 1], r[:, 3:7].max(1) - r[:, 2], r[:, 7] - r[:, 3:7].max(1)
            skew = r[:, 3:7].max(1) - r[:, 3:7].min(1)
            tot = r[:, 7] - r[:, 0]
            print(f"l={l} step {st} ({int(m.sum())} wgs): total {med(tot):.2f} us = staging {med(stage):.2f} + barrier {med(bar1):.2f} + slowest range {med(body):.2f} + end barrier {med(bar2):.2f}; "
                  f"skew {med(skew):.2f}; mean range {med((r[:, 3:7] - r[:, 2:3]).mean(1)):.2f}")
if os.environ.get("XEQ_WQ_LOOP_TIME"):   # where a range's time goes: prologue (before the tile loop) against the tile loop
    import ctypes
    from xequinet_amd import lib as _lib
    L = _lib.load()
    run("wq"); torch.cuda.synchronize()
    lp = (ctypes.c_ulonglong * (8192 * 4))(); lp2 = (ctypes.c_ulonglong * (8192 * 8))()
    L.xeq_wq_debug_loop(lp, lp2)
    lp = np.frombuffer(lp, dtype=np.uint64).reshape(8192, 4).astype(np.float64); lp2 = np.frombuffer(lp2, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
    m = (lp > 0) & (lp2[:, :4] > 0) & (lp2[:, 4:] > lp)
    pro = (lp - lp2[:, :4])[m] / 100.0; loop = (lp2[:, 4:] - lp)[m] / 100.0
    print(f"last range of {int(m.sum())} waves: prologue (barrier -> tile loop) median {np.median(pro):.2f} us (p90 {np.quantile(pro, .9):.2f}), tile loop median {np.median(loop):.2f} us (p90 {np.quantile(loop, .9):.2f})")

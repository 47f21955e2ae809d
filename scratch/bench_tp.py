"""The reference's two general tensor products on the fused kernel (all paths, one launch) against the round-2 form (one launch per
path): SelfMixTP's 'uuu' product (nn/xe3net.py:118-146, shared internal weights) and the Cartesian-tensor head's 'uuw' product with one
weight set per sample (nn/output.py:411-421).  python scratch/bench_tp.py [n] -> table + gpurun_out/r04_tp_bench.csv"""
import os, sys, time, csv, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import tp
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 18609

def selfmix(ch=64, lmax=2):
    hid = tp.Irreps([(ch, (l, (-1) ** l)) for l in range(lmax + 1)])
    mix = [(ch, (0, 1))]
    for l in range(2, 2 * lmax):
        mix += [(ch, (l, -1)), (ch, (l, 1))]
    mix.append((ch, (2 * lmax, 1)))
    out, ins = tp.get_feasible_tp(hid, hid, tp.Irreps(mix), "uuu")
    return tp.TensorProduct(hid, hid, out, ins, internal_weights=True, shared_weights=True), hid, hid

def cartesian(ch=64, lmax=2):
    m, _, _ = selfmix(ch, lmax)
    mixed = m.irreps_out
    rtp = tp.Irreps("1x0e+1x2e")      # symmetric rank-2 tensor (Sph2Cart 'ij=ji')
    out, ins = tp.get_feasible_tp(mixed, mixed, rtp, "uuw")
    return tp.TensorProduct(mixed, mixed, out, ins, internal_weights=False, shared_weights=False), mixed, mixed

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6

rows = []
for name, build in (("SelfMixTP uuu 64 channels lmax 2", selfmix), ("Cartesian head uuw per-sample weights", cartesian)):
    mod, i1, i2 = build()
    mod = mod.to(dev).float()
    x = torch.randn(n, i1.dim, device=dev, requires_grad=True); y = torch.randn(n, i2.dim, device=dev, requires_grad=True)
    w = None if mod.internal_weights else torch.randn(n, mod.weight_numel, device=dev, requires_grad=True)
    res = {}
    g = None
    for fused in (False, True):
        mod.fused = fused
        args = (x, y) if w is None else (x, y, w)
        out = mod(*args)
        g = torch.randn_like(out) if g is None else g
        res[fused] = (out.detach().clone(), [t.clone() for t in torch.autograd.grad(out, [x, y] + ([w] if w is not None else [mod.weight]), g)])
        with torch.no_grad():
            t_f = timeit(lambda: mod(*args))
        def fb():
            o = mod(*args)
            torch.autograd.grad(o, [x, y] + ([w] if w is not None else [mod.weight]), g)
        t_fb = timeit(fb)
        rows.append(dict(product=name, n=n, paths=len(mod.instructions), dim_in=i1.dim, dim_out=mod.irreps_out.dim, weights=mod.weight_numel,
                         form="all paths, one launch" if fused else "one launch per path", forward_us=round(t_f, 1), forward_backward_us=round(t_fb, 1),
                         bytes_in_out=4 * n * (i1.dim + i2.dim + mod.irreps_out.dim + (mod.weight_numel if w is not None else 0))))
    d_out = float((res[True][0] - res[False][0]).abs().max())
    d_g = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(res[True][1], res[False][1]))
    print(f"{name}: {len(mod.instructions)} paths, dims {i1.dim} x {i2.dim} -> {mod.irreps_out.dim}, {mod.weight_numel} weights; fused vs per-path: max |d out| {d_out:.2e}, gradients rel {d_g:.2e}")
for r in rows:
    print(f"  {r['product'][:36]:36s} {r['form']:22s} forward {r['forward_us']:9.1f} us   forward + backward {r['forward_backward_us']:9.1f} us   ({r['bytes_in_out'] / 1e6:.1f} MB in + out)")
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/r04_tp_bench.csv", "w", newline="") as fh:
    wr = csv.DictWriter(fh, fieldnames=list(rows[0])); wr.writeheader(); wr.writerows(rows)

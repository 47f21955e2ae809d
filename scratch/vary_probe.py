"""Whole-step graph (runtime.GraphedStep): what a capacity larger than the batch costs, and what each draw costs on its own.
usage (GPU box): python scratch/vary_probe.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import runtime
from xequinet_amd.data import synthetic as syn, XequiBatch
from xequinet_amd.nn import resolve_model
dev = "cuda"
torch.manual_seed(0)
model = resolve_model("xpainn").to(dev).eval().requires_grad_(False)
draws = []
for k in range(4):
    p, z, ptr, _ = syn.make_workload("qm9_1024", seed=1234 if k == 0 else 4321 + 97 * k)
    b = XequiBatch(torch.tensor(p, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev))
    draws.append((b.pos, b.atomic_numbers, b.ptr, ptr, b.batch))
def run(cap, which, steps=30):
    g = runtime.GraphedStep(model, cap, compute_forces=True)
    tot = torch.zeros(1, dtype=torch.int64, device=dev)
    for i in range(6):
        d = draws[which[i % len(which)]]; g(d[0], d[1], d[2], batch=d[4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        d = draws[which[i % len(which)]]; tot.add_(g(d[0], d[1], d[2], batch=d[4])["n_edges"])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    return dt * 1e3, tot.item() / steps / dt / 1e6
G = len(draws[0][3]) - 1
for k, d in enumerate(draws):
    n, e = d[0].shape[0], runtime.pair_capacity(d[3])
    ms, me = run((n + 64, G, e), [k])
    print(f"draw {k}: N = {n}, pair capacity {e}: tight capacity {ms:.3f} ms, {me:.1f} M edges/s", flush=True)
nmax, emax = max(d[0].shape[0] for d in draws), max(runtime.pair_capacity(d[3]) for d in draws)
ms, me = run((nmax + 64, G, emax), [0]); print(f"draw 0 under the common capacity ({nmax + 64}, {emax}): {ms:.3f} ms, {me:.1f} M edges/s")
ms, me = run((nmax + 64, G, emax), [0, 1, 2, 3]); print(f"four draws in turn under the common capacity: {ms:.3f} ms, {me:.1f} M edges/s")
ms, me = run((nmax + 2000, G, int(emax * 1.2)), [0]); print(f"draw 0 under a capacity 10 % / 20 % too large: {ms:.3f} ms, {me:.1f} M edges/s")

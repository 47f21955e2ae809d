"""Weight-gradient contractions of the training pass, dW[M, K] = G[n, M]^T X[n, K] with n = 18 609 nodes: library forms against each other.
usage (GPU box): python scratch/bench_wgrad.py"""
import time, torch
dev = "cuda"
n = 18609


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def split(g, x, chunk):
    m = (n // chunk) * chunk
    out = torch.bmm(g[:m].view(-1, chunk, g.shape[1]).transpose(1, 2), x[:m].view(-1, chunk, x.shape[1])).sum(0)
    if m < n:
        out = out + torch.mm(g[m:].t(), x[m:])
    return out


for M, K in ((576, 128), (128, 128), (480, 128), (128, 352), (128, 224), (64, 128)):
    g, x = torch.randn(n, M, device=dev), torch.randn(n, K, device=dev)
    ref = torch.mm(g.double().t(), x.double())
    forms = {"mm(g.t, x)": lambda: torch.mm(g.t(), x), "mm(x.t, g).t": lambda: torch.mm(x.t(), g).t(),
             "mm(g.t.contig, x)": lambda: torch.mm(g.t().contiguous(), x),
             "split 512": lambda: split(g, x, 512), "split 1024": lambda: split(g, x, 1024), "split 2048": lambda: split(g, x, 2048)}
    line = [f"[{M} x {n}] x [{n} x {K}]:"]
    for name, fn in forms.items():
        err = (fn().double() - ref).abs().max().item() / ref.abs().max().item()
        line.append(f"{name} {t(fn):.0f} us (err {err:.1e})")
    print("  ".join(line), flush=True)

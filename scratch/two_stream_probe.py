"""Probe: one 1024-molecule evaluation vs two 512-molecule halves replayed concurrently on two streams."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd.data import NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
from xequinet_amd.runtime import GraphedModel
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(dev)
pos, z, ptr = syn.synth_qm9_batch(1024, seed=1234)
tr = NeighborTransform(5.0)
def mk(g0, g1):
    a, b = int(ptr[g0]), int(ptr[g1])
    d = tr(XequiBatch(torch.tensor(pos[a:b], dtype=torch.float32, device=dev), torch.tensor(z[a:b], device=dev),
                      torch.tensor(ptr[g0:g1 + 1] - ptr[g0], device=dev))).to_dict()
    return d
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
full = mk(0, 1024)
cuts = [1024 * i // nsplit for i in range(nsplit + 1)]
parts = [mk(cuts[i], cuts[i + 1]) for i in range(nsplit)]
gm_full = GraphedModel(model)
gms = [GraphedModel(model) for _ in parts]
streams = [torch.cuda.Stream() for _ in parts]
gm_full(full)
for g, d in zip(gms, parts): g(d)
torch.cuda.synchronize()
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def seq():
    for g, d in zip(gms, parts): g(d)
def conc():
    cur = torch.cuda.current_stream()
    for s, g, d in zip(streams, gms, parts):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            g(d)
    for s in streams: cur.wait_stream(s)
print(f"full batch replay {t(lambda: gm_full(full)):.3f} ms | {nsplit} parts sequential {t(seq):.3f} ms | concurrent on {nsplit} streams {t(conc):.3f} ms")

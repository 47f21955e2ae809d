"""Where a chunked, replayed evaluation spends its wall time (bench.py --workload qm9_65536 path) on a smaller batch.
usage (GPU box): python scratch/debug_chunks.py [n_mol] [max_chunk_edges]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import runtime, dist as xdist, keys
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model

n_mol = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
max_edges = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
dev = "cuda"
pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=1234)
pos_d, z_d, ptr_d = torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)
torch.manual_seed(0)
model = resolve_model("xpainn").to(dev).eval().requires_grad_(False)
chunks = xdist.plan_chunks(ptr, max_edges)
print("chunks", len(chunks), [int(ptr[b] - ptr[a]) for a, b in chunks])
graphed = runtime.GraphedModel(model, compute_forces=True, compute_virial=False, tune_gemms=False, max_graphs=8)
transform = NeighborTransform(5.0)
sync = torch.cuda.synchronize


def step(detail=False):
    t = {"list": 0.0, "run": 0.0, "clone": 0.0}
    for g0, g1 in chunks:
        a, b = int(ptr[g0]), int(ptr[g1])
        if detail: sync(); t0 = time.perf_counter()
        batch = transform(XequiBatch(pos_d[a:b].detach(), z_d[a:b], ptr_d[g0 : g1 + 1] - ptr_d[g0]))
        if detail: sync(); t1 = time.perf_counter()
        out = graphed(batch.to_dict())
        if detail: sync(); t2 = time.perf_counter()
        out = {k: v.clone() for k, v in out.items()}
        if detail:
            sync(); t3 = time.perf_counter()
            t["list"] += t1 - t0; t["run"] += t2 - t1; t["clone"] += t3 - t2
    return t


for _ in range(4):
    step()
sync()
print("captures after warm-up", graphed.captures)
t0 = time.perf_counter()
for _ in range(5):
    step()
sync()
print(f"step {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms; captures {graphed.captures}")
acc = None
for _ in range(5):
    t = step(detail=True)
    acc = t if acc is None else {k: acc[k] + v for k, v in t.items()}
print({k: f"{v / 5 * 1e3:.1f} ms" for k, v in acc.items()})
# host-side profile of one replayed chunk
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
step()
sync()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)

"""Throughput of whole steps with D batches in flight: D GraphedStep contexts, each replayed on its own stream, round-robin.
python scratch/two_in_flight.py [workload] [depths...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.nn import resolve_model
from xequinet_amd.data import synthetic as syn
from xequinet_amd import runtime

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(torch.float32).to(dev)
wl = sys.argv[1] if len(sys.argv) > 1 else "qm9_1024"
pos, z, ptr, _ = syn.make_workload(wl, seed=1234)
t = lambda a, dt=None: torch.as_tensor(a, device=dev).to(dt) if dt is not None else torch.as_tensor(a, device=dev)
a = (t(pos, torch.float32), t(z), t(ptr))
cap = (len(pos) + 64, len(ptr) - 1, runtime.pair_capacity(ptr))


def run(depth, n=60):
    ctx = [runtime.GraphedStep(model, cap) for _ in range(depth)]
    outs = [c(*a) for c in ctx]           # captures
    torch.cuda.synchronize()
    ref = outs[0]["forces"].clone()
    streams = [torch.cuda.Stream() for _ in range(depth)]

    def loop(k):
        for i in range(k):
            c = ctx[i % depth]
            with torch.cuda.stream(streams[i % depth]):
                c._load(*a, None)
                c.graph.replay()
    loop(2 * depth); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(n); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    same = all(torch.equal(c.outputs["forces"][: len(pos)], ref) for c in ctx)
    return ms, same


for d in [int(x) for x in sys.argv[2:]] or [1, 2, 3]:
    ms, same = run(d)
    print(f"{wl}: {d} in flight: {ms:.3f} ms per step   bitwise equal forces: {same}", flush=True)

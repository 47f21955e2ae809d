"""Does it pay to run a batch as TWO half-batches on two streams inside one captured graph (the node-block launches of one half under
the message launches of the other)?  python scratch/two_lane_probe.py [lanes...]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.nn import resolve_model
from xequinet_amd.data import synthetic as syn
from xequinet_amd import runtime, dist as xdist

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(torch.float32).to(dev)
pos, z, ptr, _ = syn.make_workload("qm9_1024", seed=1234)
t = lambda a, dt=None: torch.as_tensor(a, device=dev).to(dt) if dt is not None else torch.as_tensor(a, device=dev)


def time_it(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def lanes_run(L):
    cuts = xdist.shard_by_edges(ptr, L)
    steps, args = [], []
    for g0, g1 in cuts:
        p, zz, pp = xdist.take_shard(pos, z, ptr, g0, g1)
        cap = (len(p) + 64, g1 - g0, int(runtime.pair_capacity(pp) ))
        st = runtime.GraphedStep(model, cap)
        a = (t(p, torch.float32), t(zz), t(pp))
        st._load(*a, None)
        steps.append(st); args.append(a)
    streams = [torch.cuda.Stream() for _ in range(L)]
    main = torch.cuda.current_stream()
    # warm-up
    for st in steps:
        for _ in range(2): st._step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    outs = [None] * L
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream()
        for i, st in enumerate(steps):
            if i == 0: continue
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs[i] = st._step()
        outs[0] = steps[0]._step()
        for i in range(1, L): cur.wait_stream(streams[i])
    ms = time_it(g.replay)
    E = torch.cat([o["energy"][: c[1] - c[0]] for o, c in zip(outs, cuts)])
    F = torch.cat([o["forces"][: len(a[0])] for o, a in zip(outs, args)])
    return ms, E.clone(), F.clone()

full = runtime.GraphedStep(model, (len(pos) + 64, len(ptr) - 1, runtime.pair_capacity(ptr)))
a = (t(pos, torch.float32), t(z), t(ptr))
o = full(*a)
E0, F0 = o["energy"].clone(), o["forces"].clone()
print(f"one lane (GraphedStep): {time_it(lambda: full(*a)):.3f} ms   XEQ_NODE_BLOCK_MIN_NODES={os.environ.get('XEQ_NODE_BLOCK_MIN_NODES')}")
for L in (2, 4):     # one shard of L alone: what a lane costs when nothing runs beside it
    g0, g1 = xdist.shard_by_edges(ptr, L)[0]
    p_, zz_, pp_ = xdist.take_shard(pos, z, ptr, g0, g1)
    st_ = runtime.GraphedStep(model, (len(p_) + 64, g1 - g0, int(runtime.pair_capacity(pp_))))
    a_ = (t(p_, torch.float32), t(zz_), t(pp_))
    print(f"shard 1/{L} alone ({len(p_)} atoms): {time_it(lambda: st_(*a_)):.3f} ms")
for L in [int(x) for x in sys.argv[1:]] or [2, 3]:
    ms, E, F = lanes_run(L)
    print(f"{L} lanes in one graph: {ms:.3f} ms   max |dE| {float((E - E0).abs().max()):.2e}  max |dF| {float((F - F0).abs().max()):.2e}")

#!/usr/bin/env python
"""Headline benchmark: directed edges/sec of one XPaiNN energy+force evaluation
(neighbour list + 3 message/update blocks forward + force backward), fp32, random-init weights,
synthetic inputs (xequinet_amd/data/synthetic.py).

    python bench.py [--gpus N --steps K --warmup W] [--workload qm9_1024 | qm9_65536 | ...]     (N > 1: starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; molecules are independent, so ranks never exchange data.
  * default (qm9_1024, BASELINE.json configs[1]): every rank evaluates its own 1024-molecule batch -- weak scaling.
  * --workload qm9_65536 (configs[4]): ONE 65536-molecule batch, cut into contiguous molecule ranges balanced on edge
    count (dist.shard_by_edges); rank r evaluates range r, in chunks that fit the kernels' 32-bit offsets
    (runtime.evaluate_in_chunks) -- strong scaling.
Rank 0 prints ONE JSON line.  See DESIGN.md "Measurement" for the roofline bookkeeping.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

METRIC = "edges/sec + achieved HBM GB/s, energy+force inference, QM9-shape batch"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact f32 (= the f32 vector rate)
SHARDED = ("qm9_65536", "qm9_8192_sharded")   # workloads that are ONE batch sharded over the ranks (strong scaling); the second: the
                                               # 8192-molecule batch cut over the ranks, the size a one-card rehearsal of the sharded path can afford


def _oracle_eval(orc, oracle, p, zz, pp):
    n_mol = len(pp) - 1
    ei = orc.radius_graph_canonical(p, pp, 5.0)
    batch = np.repeat(np.arange(n_mol), np.diff(pp))
    out = oracle({"pos": torch.tensor(p), "atomic_numbers": torch.tensor(zz.astype(np.int64)), "edge_index": torch.tensor(ei),
                  "batch": torch.tensor(batch), "ptr": torch.tensor(pp)})
    return ei.shape[1], out


def cpu_baseline(pos, z, ptr, sd, budget_s=25.0):
    """The oracle (a CPU restatement of the reference's eager op graph: index_select -> Linear -> elementwise ->
    index_add -> autograd.grad) timed on this host, fp32, brute-force neighbour list included.  The thread count is
    tuned first on a 64-molecule probe (the default, one thread per logical core, oversubscribes these small tensors);
    the sample is then the largest leading slice of the SAME batch that the probe's rate predicts to fit the time
    budget and the host's free memory (the oracle materialises every [E, .] tensor: ~0.12 MB per edge)."""
    from oracle import xpainn_oracle as orc

    sd32 = {k: (v.float().cpu() if v.is_floating_point() else v.cpu()) for k, v in sd.items()}
    oracle = orc.XPaiNNOracle(sd32)
    n_all = len(ptr) - 1
    default_threads = torch.get_num_threads()

    def timed(n_mol, reps):
        a = int(ptr[n_mol])
        p, zz, pp = pos[:a].astype(np.float32), z[:a], ptr[: n_mol + 1]
        n_edges, _ = _oracle_eval(orc, oracle, p, zz, pp)   # warm-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            _oracle_eval(orc, oracle, p, zz, pp)
            ts.append(time.perf_counter() - t0)
        return n_edges, float(np.median(ts))

    probe_mol = min(64, n_all)
    cores = os.cpu_count() or default_threads
    candidates = sorted({t for t in (8, 16, 32, 64, default_threads) if 1 <= t <= max(cores, default_threads)})
    rates, e_probe = {}, 1
    for t in candidates:
        torch.set_num_threads(t)
        e_probe, dt = timed(probe_mol, 2)
        rates[t] = e_probe / dt
    best = max(rates, key=rates.get)
    torch.set_num_threads(best)
    try:
        import psutil
        free = psutil.virtual_memory().available
    except Exception:
        free = 32 << 30
    edges_per_mol = e_probe / probe_mol
    by_time = budget_s / 4.0 * rates[best] / edges_per_mol * 1.6   # ~4 evaluations (warm-up + 3 timed); the whole batch when it is within 1.6 x the budget
    by_mem = 0.5 * free / (0.12e6 * edges_per_mol)
    n_mol = int(max(probe_mol, min(n_all, by_time, by_mem)))
    n_edges, med = timed(n_mol, 3)
    a = int(ptr[n_mol])
    torch.set_num_threads(default_threads)
    return {"value": n_edges / med, "unit": "edges/s", "cores": best, "kind": "port",
            "sample": f"first {n_mol} of {n_all} molecules of the batch ({a} atoms, {n_edges} edges), fp32, one warm-up + median of 3 timed runs, "
                      f"{med * 1e3:.1f} ms/eval, brute-force neighbour list included, {best} threads (tuned on {probe_mol} molecules: "
                      + ", ".join(f"{t} thr {r:.0f} e/s" for t, r in sorted(rates.items())) + f"; host has {cores} logical cores)",
            "default_threads": {"threads": default_threads, "value": rates.get(default_threads), "sample": f"{probe_mol}-molecule probe"}}


def _load_sharded_batch(name, seed):
    """The whole batch of a sharded workload (every rank needs all molecule sizes to cut the same ranges); cached
    under TMPDIR because the pure-numpy generator takes ~26 s for 65536 molecules."""
    from xequinet_amd.data import synthetic as syn

    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"xeq_{name}_{seed}.npz")
    if os.path.exists(path):
        try:
            d = np.load(path)
            return d["pos"], d["z"], d["ptr"]
        except Exception:
            pass
    pos, z, ptr, _ = syn.make_workload(name.replace("_sharded", ""), seed)
    try:
        tmp = f"{path}.{os.getpid()}.npz"
        np.savez(tmp, pos=pos, z=z, ptr=ptr)
        os.replace(tmp, path)
    except OSError:
        pass
    return pos, z, ptr


def run_train(args, rank, world, dev, dtype, xdist):
    """bench.py --train [--forces]: K optimisation steps of the reference's inner loop (utils/trainer.py:290-308) on this rank's own batch
    of the workload, barrier + synchronize on both sides, MAX over ranks; with --gpus > 1 the model is DistributedDataParallel
    (run/train.py:185-190), so every timed step carries the bucketed gradient all-reduce over RCCL."""
    from xequinet_amd import keys, train
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.data import synthetic as syn
    from xequinet_amd.nn import resolve_model

    pos, z, ptr, cell = syn.make_workload(args.workload, seed=1234 + rank)
    assert cell is None, "--train takes the open-boundary workloads"
    torch.manual_seed(0)                                        # the same initial weights on every rank
    model = resolve_model("xpainn").to(dtype).to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    ddp = train.wrap_ddp(model, local_rank=dev.index)
    opt = torch.optim.Adam(ddp.parameters(), lr=1e-4)
    b = NeighborTransform(5.0)(XequiBatch(torch.tensor(pos, dtype=dtype, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)))
    data = b.to_dict()
    n_edges = int(data["edge_index"].shape[1])
    g = torch.Generator().manual_seed(rank)
    target = {keys.TOTAL_ENERGY: torch.randn(len(ptr) - 1, generator=g).to(dtype).to(dev), keys.BATCH_PTR: data["ptr"]}
    weights = {keys.TOTAL_ENERGY: 1.0}
    if args.forces:
        target[keys.FORCES] = torch.randn(len(pos), 3, generator=g).to(dtype).to(dev)
        weights[keys.FORCES] = 10.0

    def step():
        d = {k: v for k, v in data.items() if not k.startswith("_")}
        d["pos"] = d["pos"].detach().clone()
        return train.train_step(ddp, d, target, opt, weights)[0]

    for _ in range(max(args.warmup, 15 if args.forces else 5)):     # (the caching allocator of a new step shape settles in ~15 steps)
        step()
    xdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    xdist.barrier()
    elapsed, edges_all = xdist.reduce_timing(time.perf_counter() - t0, float(n_edges), device=dev)
    if rank == 0:
        print(json.dumps({
            "metric": "training step: edges/sec, " + ("energy + forces loss (twice-differentiated pass)" if args.forces else "energy loss (native pass)"),
            "value": edges_all * args.steps / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if dtype == torch.float32 else "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: one optimisation step (forward, l2 loss, backward, Adam) per step on {len(pos)} atoms / {n_edges} edges per rank",
                       "parallelism": f"DistributedDataParallel x{world}: one gradient bucket of {n_params * 4 / 1e6:.2f} MB all-reduced per step" if world > 1 else "one rank, no collective",
                       "loss": float(loss), "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30}}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    from xequinet_amd.data import synthetic as syn

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)    # (0.2 s of timed region at 2 ms per step: the fill and drain of two steps in flight are 1 % of it)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="qm9_1024", choices=list(syn.WORKLOADS))
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true",
                    help="launch every kernel of the model part from the host in the timed region (default: the model part is "
                         "a captured HIP graph, runtime.GraphedModel; the neighbour list is eager either way)")
    ap.add_argument("--no-gemm-autotune", action="store_true",
                    help="keep the libraries' default GEMM heuristics (xequinet_amd/tuning.py)")
    ap.add_argument("--gemm-results", default=None,
                    help="file of library-GEMM selections: written by a run that times them, replayed (no timing launches) "
                         "when it already exists -- used by profiles/collect_stats.sh so that the trace holds no tuning kernels")
    ap.add_argument("--max-chunk-edges", type=int, default=None, help="edge cap per chunk of a sharded workload")
    ap.add_argument("--replay-model-only", action="store_true",
                    help="round 2's launch mode: the model part as a captured graph per (atoms, edges) signature, the neighbour list "
                         "launched from the host with its edge count read back (default since round 3 for open-boundary workloads: "
                         "neighbour list + model as ONE graph over capacity-sized arrays, runtime.GraphedStep)")
    ap.add_argument("--lanes", type=int, default=-1, metavar="L",
                    help="whole-step graph of an open-boundary batch: evaluate the batch as L contiguous molecule ranges in parallel branches of "
                         "the one captured graph (runtime.GraphedLanes; results bit for bit those of L = 1).  Default: runtime.auto_lanes(atoms)")
    ap.add_argument("--in-flight", type=int, default=2, metavar="D",
                    help="whole-step graph of an open-boundary batch: D steps in flight (runtime.GraphedStepsInFlight: D contexts with their own "
                         "buffers and captured graphs, each replayed on its own stream, in turn; every step's results are bit for bit those of "
                         "D = 1).  The K timed steps are still K whole steps, all finished before the clock stops; what overlaps is the "
                         "front / the tails of one step with the body of the next.  D = 1: one step at a time (also reported by the default "
                         "run as ms_per_step_one_in_flight)")
    ap.add_argument("--train", action="store_true",
                    help="time ONE OPTIMISATION STEP per step instead of an inference step (SURVEY 8f-4; utils/trainer.py:290-308): forward in train "
                         "mode, weighted l2 loss, backward, Adam, the model wrapped in DistributedDataParallel when --gpus > 1 so that the timed "
                         "step includes the gradient all-reduce (RCCL; 865 k fp32 gradients, one bucket).  Prints its own JSON line "
                         "(metric 'training step'); the headline metric is the default mode's")
    ap.add_argument("--forces", action="store_true", help="with --train: forces in the loss (weight 10), i.e. the twice-differentiated pass")
    ap.add_argument("--vary-batch", type=int, default=-1, metavar="K",
                    help="feed K different draws of the workload in turn (different atom and edge counts every step) through the "
                         "one captured graph.  Default: as many draws as steps in flight (two contexts evaluate two different "
                         "batches: a stream of batches, not one batch twice), one draw with --in-flight 1")
    args = ap.parse_args()
    if args.vary_batch < 0:   # (a sharded workload is ONE batch by definition: every step evaluates this rank's share of it)
        args.vary_batch = 1 if args.workload in SHARDED else max(1, args.in_flight)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here, BEFORE anything in this process touches the GPU
        # (a child process per rank under torch.distributed.run -- never a re-exec of a process that has initialised HIP);
        # rank 0's JSON line passes through, this process exits with the launcher's code
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    from xequinet_amd import dist as xdist
    from xequinet_amd import ops, runtime
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.nn import resolve_model

    # rehearsal of the multi-rank path on a one-GPU box: XEQ_BENCH_BACKEND=gloo XEQ_BENCH_DEVICE=0 (every rank on that device)
    rank, local_rank, world = xdist.init_from_env(os.environ.get("XEQ_BENCH_BACKEND"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    dev = torch.device("cuda", int(os.environ.get("XEQ_BENCH_DEVICE", local_rank)))
    torch.cuda.set_device(dev)
    dtype = torch.float32 if args.dtype == "f32" else torch.float64

    torch.manual_seed(0)
    model = resolve_model("xpainn").eval().requires_grad_(False)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dtype).to(dev)

    if args.train:
        return run_train(args, rank, world, dev, dtype, xdist)

    sharded = args.workload in SHARDED
    if sharded:
        # ONE batch for the whole job; this rank's share is a contiguous molecule range balanced on edge count
        pos_all, z_all, ptr_all = _load_sharded_batch(args.workload, 1234)
        g0, g1 = xdist.shard_by_edges(ptr_all, world)[rank]
        pos, z, ptr = xdist.take_shard(pos_all, z_all, ptr_all, g0, g1)
        cell = None
    else:
        # every rank gets its own batch of the same shape (weak scaling)
        pos, z, ptr, cell = syn.make_workload(args.workload, seed=1234 + rank)
    pos_d = torch.tensor(pos, dtype=dtype, device=dev)      # inputs resident in HBM before the timed region
    z_d = torch.tensor(z, device=dev)
    ptr_d = torch.tensor(ptr, device=dev)
    cell_d = None if cell is None else torch.tensor(cell, dtype=dtype, device=dev)
    pbc_d = None if cell is None else torch.tensor([[True, True, True]], device=dev)
    transform = NeighborTransform(model.cutoff_radius)
    if not args.no_gemm_autotune:
        from xequinet_amd.tuning import enable_gemm_autotune
        enable_gemm_autotune(results_file=args.gemm_results)   # every GEMM shape is timed once, during the warm-up steps

    # the collated batch (positions, atomic numbers, graph pointer and the per-atom graph index: what the reference's
    # DataLoader hands over with `data.to(device)`, run/inference.py:39) is resident before the timed region; a step takes a
    # shallow copy of it (no device work), builds the neighbour list and evaluates the model
    import copy
    collated = XequiBatch(pos_d.detach(), z_d, ptr_d, pbc=pbc_d, cell=cell_d)

    def new_batch():
        return copy.copy(collated)

    def step_eager():
        batch = transform(new_batch())                 # HIP radius graph
        with torch.enable_grad():
            out = model(batch.to_dict(), compute_forces=True, compute_virial=False)
        return batch.edge_index.shape[1], out

    n_chunks = 1
    n_lanes = 1
    n_flight = 1
    if sharded:
        max_edges = args.max_chunk_edges or runtime.WM_MAX_EDGES_PER_CHUNK
        n_chunks = len(xdist.plan_chunks(ptr, max_edges))
    # a shard that is ONE chunk (the 65k batch over 8 ranks) is a batch like any other: it takes the whole-step path below, two steps
    # in flight; the chunk walkers are for shards beyond one step's buffers
    chunked = sharded and (n_chunks > 1 or args.eager or args.replay_model_only)
    if chunked:
        graphed = None if args.eager else runtime.GraphedModel(model, compute_forces=True, compute_virial=False, tune_gemms=False,
                                                               max_graphs=max(8, 2 * n_chunks))

        def step_chunked(runner):
            out = runtime.evaluate_in_chunks(model, pos_d, z_d, ptr_d, ptr_host=ptr, max_edges=max_edges, runner=runner)
            return out.pop("n_edges"), out

        def step_eager():                              # noqa: F811  (the chunked form of the same step)
            return step_chunked(None)

        gchunks = None
        if graphed is not None and not args.replay_model_only:
            # default since round 5: every chunk a whole captured step over capacity-sized arrays (no edge count read back, no list built
            # from the host), --in-flight of them side by side (runtime.GraphedChunks); results bit for bit those of the forms above
            gchunks = runtime.GraphedChunks(model, ptr, max_edges=max_edges, depth=max(1, args.in_flight), compute_forces=True)
            batch_d = collated.batch
            n_flight = gchunks.depth

            def step():
                return None, gchunks(pos_d, z_d, ptr_d, batch_d)
        else:
            step = step_eager if graphed is None else (lambda: step_chunked(graphed))
    elif args.eager:
        step = step_eager
    elif cell is None and not args.replay_model_only:
        # neighbour list + model as ONE captured graph (runtime.GraphedStep): arrays sized by a capacity, the edge count stays on
        # the device, nothing is read back inside a step.  --vary-batch K: K different draws of the workload (their own atom and
        # edge counts) take turns through the same graph
        draws = [(pos_d, z_d, ptr_d, ptr, collated.batch)]     # the collated batch: per-atom graph index included, as before
        for k in range(1, max(1, args.vary_batch)):
            p_k, z_k, ptr_k, _ = syn.make_workload(args.workload, seed=4321 + 97 * k + rank)
            b_k = XequiBatch(torch.tensor(p_k, dtype=dtype, device=dev), torch.tensor(z_k, device=dev), torch.tensor(ptr_k, device=dev))
            draws.append((b_k.pos, b_k.atomic_numbers, b_k.ptr, ptr_k, b_k.batch))
        cap = (max(d[0].shape[0] for d in draws) + 64, len(ptr) - 1, max(runtime.pair_capacity(d[3]) for d in draws))
        n_lanes = args.lanes if args.lanes >= 1 else runtime.auto_lanes(cap[0])
        n_flight = max(1, args.in_flight) if n_lanes == 1 else 1
        if n_lanes > 1:
            gstep = runtime.GraphedLanes(model, cap, lanes=n_lanes, compute_forces=True)
        elif n_flight > 1:
            gstep = runtime.GraphedStepsInFlight(model, cap, depth=n_flight, compute_forces=True)
        else:
            gstep = runtime.GraphedStep(model, cap, compute_forces=True)
        turn = [0]
        ticket = [None]

        def step():
            p_k, z_k, ptr_k, ph_k, b_k = draws[turn[0] % len(draws)]
            turn[0] += 1
            # every replay adds its true edge count to the step's device-side counter (inside the neighbour-list launch); read once
            # behind the timed region
            if n_flight > 1:                               # enqueue and go on: the step's results are fetched (and waited for) by ticket
                ticket[0] = gstep.submit(p_k, z_k, ptr_k, batch=b_k)
                return None, None
            out = gstep(p_k, z_k, ptr_k, b_k, ph_k) if n_lanes > 1 else gstep(p_k, z_k, ptr_k, batch=b_k)
            return None, out
    elif cell is not None and len(ptr) == 2 and not args.replay_model_only:
        # ONE periodic system: search + model as one captured graph over capacity-sized edge arrays (runtime.GraphedStepPBC); the
        # capacity comes from one sized search in front (a quarter more room than that list needs), checked behind the timed region
        n0 = int(transform(new_batch()).edge_index.shape[1])
        gpbc = runtime.GraphedStepPBC(model, pos_d.shape[0], int(1.25 * n0) + 1024, compute_forces=True)
        pbc_list = [bool(v) for v in pbc_d.reshape(-1, 3)[0].tolist()]
        edge_total = torch.zeros(1, dtype=torch.int64, device=dev)
        turn = [0]

        def step():
            out = gpbc(pos_d, z_d, cell_d.reshape(-1, 3, 3)[0], pbc_list, check=False)
            edge_total.add_(out["n_edges"])
            return None, out
    else:
        # the same kernels in the same order as one HIP-graph launch (results bitwise those of the eager path); the
        # neighbour list stays eager: its edge count has to reach the host to size the edge arrays
        graphed = runtime.GraphedModel(model, compute_forces=True, compute_virial=False, tune_gemms=False)

        def step():
            batch = transform(new_batch())
            return batch.edge_index.shape[1], graphed(batch.to_dict())

    whole_step = not chunked and not args.eager and (cell is None or len(ptr) == 2) and not args.replay_model_only
    chunk_graphs = chunked and not args.eager and not args.replay_model_only
    try:
        # set-up, not a timed or counted step: the first evaluation times the library GEMM candidates (TunableOp) and
        # captures the HIP graph; the W warm-up steps and the K timed steps that follow are all plain steps
        n_edges, out = step()
        for _ in range(args.warmup):
            n_edges, out = step()
        torch.cuda.synchronize()
        if chunk_graphs:
            n_edges = int(gchunks.edge_total.item()) // (args.warmup + 1)
        elif n_flight > 1:
            out = gstep.result(ticket[0])
        if whole_step:
            n_edges = (sum(int(st.outputs["n_edges"].item()) for st in gstep.steps) if (cell is None and n_lanes > 1)
                       else int(out["n_edges"].item()))
    except RuntimeError as err:      # a failed capture must not cost the measurement: fall back to host launches
        if args.eager:
            raise
        print(f"[bench] HIP-graph capture failed on rank {rank} ({err}); continuing with host launches", file=sys.stderr, flush=True)
        args.eager, step, chunk_graphs, whole_step, n_flight = True, step_eager, False, False, 1
        for _ in range(args.warmup + 1):
            n_edges, out = step()
        torch.cuda.synchronize()
    if world > 1:                    # every rank must time the same mode
        flag = torch.tensor([1 if args.eager else 0], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if bool(flag.item()) and not args.eager:
            args.eager, step, chunk_graphs, whole_step, n_flight = True, step_eager, False, False, 1
    n_atoms = pos_d.shape[0]

    # ---- timed region: exactly K steps between barrier + synchronize
    ops.KERNEL_TIMER.reset(enabled=args.eager)
    xdist.barrier()
    torch.cuda.synchronize()
    if chunk_graphs:
        gchunks.zero_edge_total()
        torch.cuda.synchronize()
    if whole_step:
        if cell is None:
            for st in (gstep.steps if (n_lanes > 1 or n_flight > 1) else [gstep]):
                st.edge_total.zero_()
        else:
            edge_total.zero_()
        turn[0] = 0
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges_done = 0
    for _ in range(args.steps):
        n_step, out = step()
        edges_done += n_step if n_step is not None else 0
    torch.cuda.synchronize()                              # every step in flight has finished here
    xdist.barrier()
    elapsed = time.perf_counter() - t0
    one_ms = None
    if chunk_graphs:
        edges_done = int(gchunks.edge_total.item())
        assert not gchunks.overflowed(), "a chunk's neighbour list outgrew its capacity"
        if gchunks.depth > 1:                              # the same chunks one at a time
            one = runtime.GraphedChunks(model, ptr, max_edges=max_edges, depth=1, compute_forces=True)
            for _ in range(2):
                one(pos_d, z_d, ptr_d, batch_d)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one(pos_d, z_d, ptr_d, batch_d)
            torch.cuda.synchronize()
            one_ms = (time.perf_counter() - t1) / args.steps * 1e3
            del one
    elif n_flight > 1:
        out = gstep.result(ticket[0])
        edges_timed = int(gstep.edge_total.item())
        one = runtime.GraphedStep(model, cap, compute_forces=True)   # the same step, ONE at a time (its latency): a lone step's own capture
        p_k, z_k, ptr_k, _, b_k = draws[0]
        for _ in range(3):
            one(p_k, z_k, ptr_k, batch=b_k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one(p_k, z_k, ptr_k, batch=b_k)
        torch.cuda.synchronize()
        one_ms = (time.perf_counter() - t1) / args.steps * 1e3
        out = one(p_k, z_k, ptr_k, batch=b_k)             # draw 0 again: what the host-launched and native-operator runs below are checked against
        out_is_draw0 = True
    if whole_step:
        edges_done = edges_timed if n_flight > 1 else int((gstep.edge_total if cell is None else edge_total).item())   # the device-side counts of the K timed steps
        if cell is not None:
            assert not gpbc.overflowed(), "the periodic list outgrew its capacity inside the timed region"
    eager_ms = native_ms = None
    out_is_draw0 = locals().get("out_is_draw0", False) or max(1, args.vary_batch) == 1
    if args.eager:
        kernel_ms = ops.KERNEL_TIMER.summary()       # HIP events recorded on the launch stream, in the timed region
        kernel_timing = "HIP events around every launch of the timed region, on the launch stream"
        cal = args.steps
    else:
        # HIP offers no per-node event timing inside a graph launch on this stack (torch: "External events are disallowed
        # in rocm"), so the message kernels' launch durations are read right after the timed region: the same kernels on
        # the same inputs, launched from the host between HIP events on the launch stream
        out_keep = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
        cal = max(1, min(args.steps, 10))
        for _ in range(3):                           # untimed: the allocator's pool outside the captured graph fills up here
            step_eager()
        torch.cuda.synchronize()
        te = time.perf_counter()                     # the same steps with host launches, before any event is recorded
        for _ in range(cal):
            step_eager()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - te) / cal * 1e3
        if not chunked and cell is None and dtype == torch.float32:
            # the same step through ONE registered operator (xeq::xpainn_eval, csrc/xeq_torch.cpp): every kernel enqueued
            # from C++, nothing captured -- what a stream of batches with ever-new topologies pays
            try:
                from xequinet_amd.interface.scripted import XPaiNNNative
                native = XPaiNNNative(model)

                def step_native():
                    batch = transform(new_batch())
                    return native(batch.pos, batch.atomic_numbers, batch.edge_index, batch.ptr, None, None, True, True, True, False)

                for _ in range(3):
                    step_native()
                torch.cuda.synchronize()
                tn = time.perf_counter()
                for _ in range(cal):
                    got = step_native()
                torch.cuda.synchronize()
                native_ms = (time.perf_counter() - tn) / cal * 1e3
                if whole_step:   # the captured step works on the padded batch (another row count for the remaining library GEMMs)
                    if out_is_draw0:
                        assert float((got[2] - out_keep["forces"]).abs().max()) <= 2e-3, "native operator and graph replay disagree"
                else:
                    assert torch.equal(got[2], out_keep["forces"]), "native operator and graph replay disagree"
            except ImportError as err:
                print(f"[bench] native operator not timed: {err}", file=sys.stderr, flush=True)
        ops.KERNEL_TIMER.reset(enabled=True)
        for _ in range(cal):
            step_eager()
        kernel_ms = ops.KERNEL_TIMER.summary()
        kernel_timing = (f"HIP events around the kernel's launches in {cal} host-launched evaluations of the same batch right after the "
                         "timed region (inside it the launches are nodes of one HIP graph, which HIP cannot time one by one here)")
        out = out_keep
    ops.KERNEL_TIMER.reset(enabled=False)

    t_max, edges_total = xdist.reduce_timing(elapsed, edges_done, device=dev)
    assert bool(torch.isfinite(out["energy"]).all()) and bool(torch.isfinite(out["forces"]).all())

    if rank == 0:
        ms_per_step = t_max / args.steps * 1e3
        value = edges_total / t_max
        # dominant kernel: fused message reverse pass.  Algorithmic bytes per launch:
        #   B_bwd = 10 880 N + 40 E   (fp32 features, int64 indices), one launch per layer (and per chunk)
        # The model's first message block runs the kernels' first-block forms (x = 0 behind the embedding, no node gradients in a
        # force evaluation): fewer bytes and flops per launch, booked under their own label so that the figure of the general
        # form is not flattered by their shorter launches (their own numbers: "first_block_form" below).
        esz = 4 if dtype == torch.float32 else 8
        general = {k: v for k, v in kernel_ms.items() if not k.endswith("_first")}
        dom = max(general, key=lambda k: general[k]["total_ms"]) if general else None
        roofline = None
        if dom is not None:
            layers = 3.0 - (1.0 if dom + "_first" in kernel_ms else 0.0)   # message blocks one evaluation runs in this form
            launches_per_eval = kernel_ms[dom]["launches"] / cal          # layers x chunks
            n_launch_nodes = layers * n_atoms / launches_per_eval          # nodes / edges one launch covers (average)
            n_launch_edges = layers * n_edges / launches_per_eval
            alg_fwd = (2272 * esz) * n_launch_nodes + (16 + 3 * esz) * n_launch_edges   # B_fwd
            alg_bwd = (2720 * esz) * n_launch_nodes + (16 + 6 * esz) * n_launch_edges   # B_bwd
            alg = alg_bwd if "bwd" in dom else alg_fwd
            avg_ms = kernel_ms[dom]["total_ms"] / kernel_ms[dom]["launches"]
            achieved = alg / (avg_ms * 1e-3) / 1e9
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tfile):
                traffic = json.load(open(tfile)).get(args.workload, {}).get(dom)
            # the same launch against the f32 matrix pipe (~26 kFLOP per edge-layer forward, filter
            # [576 x 21] + gating; the reverse pass evaluates the filter and its d/dd): what the kernel is nearer to.  The flops
            # are the ALGORITHM's (f32); the kernel spends fewer pipe cycles on them than the exact-f32 instruction would
            # (split-bf16 filter), the peak stays the exact-f32 one
            flops = (52.0e3 if "bwd" in dom else 26.0e3) * n_launch_edges
            tfl = flops / (avg_ms * 1e-3) / 1e12
            # which roof binds: the launch's arithmetic intensity against the ridge of the exact-f32 matrix pipe over HBM
            # (157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B, MI355X_MICROARCH.md).  The message kernels sit at ~75 FLOP/B: matrix-pipe bound.
            intensity = flops / alg
            ridge = F32_MATRIX_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
            mfma_bound = intensity > ridge
            hbm_line = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
            head = ({"bound": "mfma", "achieved": tfl, "peak": F32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / F32_MATRIX_PEAK_TFLOPS}
                    if mfma_bound else dict(bound="hbm", **hbm_line))
            traffic_source = (None if traffic is None else
                              "profiles/traffic.json: HBM bytes per launch from the separate FETCH_SIZE / WRITE_SIZE rocprofv3 --pmc passes of "
                              "profiles/collect_traffic.sh on this workload (FETCH doubled per MI355X_MICROARCH.md), recorded when the "
                              "profiles were taken -- NOT measured inside this run")
            roofline = {**head, "kernel": dom, "mfma_dtype": "f32 (exact, v_mfma_f32_32x32x2_f32: 157.3 TFLOP/s dense)",
                        "traffic_source": traffic_source,
                        "arithmetic_intensity_flop_per_byte": intensity, "ridge_flop_per_byte": ridge,
                        "traffic": traffic, "avg_launch_ms": avg_ms, "timing": kernel_timing,
                        "algorithmic_bytes_per_launch": alg, "algorithmic_flops_per_launch": flops, "hbm": hbm_line,
                        "matrix_pipe": {"algorithmic_flops_per_launch": flops, "achieved": tfl, "peak": F32_MATRIX_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": tfl / F32_MATRIX_PEAK_TFLOPS,
                                        "dtype": "algorithmic f32 flops against the exact-f32 rate (v_mfma_f32_32x32x2_f32); the filter runs as three-way bf16 "
                                                 "splits (nine v_mfma_f32_32x32x16_bf16 per chain since round 6: 288 pipe cycles per filter tile, round 3-5: 384, "
                                                 "exact f32: 704), so the fraction can exceed what the exact-f32 pipe alone allows; against the bf16-split-"
                                                 "equivalent rate (2.5 PFLOP/s / 6) it is 0.377 x this fraction"},
                        "kernels_ms_per_step": {k: v["total_ms"] / cal for k, v in kernel_ms.items()}}
            first = kernel_ms.get(dom + "_first")
            if first is not None:
                # first-block form of the same kernel: reads h without the l > 0 gate_state columns (480 of 576 floats), xhat on
                # the 0e columns only (128 of 480), the centers' grad_s / grad_x (608); writes no node gradients (reverse), or
                # s_in, x_in and the outputs as before (forward)
                f_ms = first["total_ms"] / first["launches"]
                per_eval = first["launches"] / cal
                nodes1, edges1 = n_atoms / per_eval, n_edges / per_eval
                alg1 = (1216 * esz) * nodes1 + (16 + 6 * esz) * edges1 if "bwd" in dom else (1824 * esz) * nodes1 + (16 + 3 * esz) * edges1
                traffic1 = json.load(open(tfile)).get(args.workload, {}).get(dom + "_first") if os.path.exists(tfile) else None
                roofline["first_block_form"] = {"kernel": dom + " (first-block form)", "avg_launch_ms": f_ms, "algorithmic_bytes_per_launch": alg1,
                                                "achieved": alg1 / (f_ms * 1e-3) / 1e9, "frac": alg1 / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "traffic": traffic1}

        what = "ONE batch sharded by molecule over the GPUs" if sharded else "per GPU"
        line = {
            "metric": METRIC, "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_eager": eager_ms if eager_ms is not None else ms_per_step,
            "ms_per_step_native_op": native_ms, "ms_per_step_one_in_flight": one_ms if one_ms is not None else ms_per_step,
            # the headline `value` is THROUGHPUT (edges/s of K whole steps between the two barriers) and, since round 5, is taken with
            # `in_flight` steps overlapped; the one-step-at-a-time figure (rounds 1-4's definition, a step's latency) stays next to it
            "value_one_in_flight": (edges_total / args.steps) / ((one_ms if one_ms is not None else ms_per_step) * 1e-3) if world == 1 else None,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {syn.WORKLOADS[args.workload]} ({what}), 5 A cutoff, default XPaiNN (865141 params, "
                                   "random init), neighbour list + energy + forces",
                       "atoms_rank0": int(n_atoms), "edges_rank0": int(n_edges), "edges_all_ranks_per_step": edges_total / args.steps,
                       "parallelism": f"molecule shards x{world}, no collectives", "chunks_rank0": n_chunks,
                       "resident_inputs": "the collated batch (positions, atomic numbers, graph pointer, per-atom graph index); every step builds its neighbour list and evaluates the model",
                       "library_gemm_selection": ("no library GEMM on the f32 path since round 3 (every contraction is an xeq kernel)" if dtype == torch.float32 else
                                                  "default heuristics" if args.no_gemm_autotune else "timed once per shape in warm-up (TunableOp)"),
                       "launch": ("host launch per kernel" if args.eager else
                                  (f"neighbour list + model as ONE captured HIP graph over capacity-sized arrays, edge count on the device (runtime.{'GraphedLanes: ' + str(n_lanes) + ' contiguous molecule ranges as parallel branches of the graph, results bit for bit those of one range' if n_lanes > 1 else ('GraphedStepsInFlight: ' + str(n_flight) + ' steps in flight, each on its own stream with its own buffers, all K finished inside the timed region; results bit for bit those of one at a time, whose time is ms_per_step_one_in_flight') if n_flight > 1 else 'GraphedStep'}; {max(1, args.vary_batch)} different draw(s) of the workload in turn)"
                                   if cell is None else "periodic neighbour search + model as ONE captured HIP graph over capacity-sized edge arrays, edge count on the device (runtime.GraphedStepPBC)")
                                  if whole_step else
                                  (f"every chunk a whole captured step (neighbour list + model + forces over capacity-sized arrays, edge count on the device), {n_flight} chunk(s) in flight on their own streams (runtime.GraphedChunks); results bit for bit those of one chunk at a time, whose time is ms_per_step_one_in_flight"
                                   if chunk_graphs else "model part (forward + force backward) as one captured HIP graph per step (per chunk); neighbour list launched from the host")),
                       "ms_per_step_eager": "the same step with every kernel launched from the host through the Python modules, measured on rank 0 right after the timed region",
                       "in_flight": n_flight,
                       "ms_per_step_native_op": "the same step as ONE registered operator (xeq::xpainn_eval: kernels enqueued from C++, no capture): what a stream of batches with ever-new edge counts pays"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pos, z, ptr, sd)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

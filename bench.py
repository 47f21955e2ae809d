#!/usr/bin/env python
"""Headline benchmark: directed edges/sec of one XPaiNN energy+force evaluation
(neighbour list + 3 message/update blocks forward + force backward) on a 1024-molecule
QM9-shape synthetic batch per GPU (BASELINE.json configs[1]), fp32, random-init weights.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; molecules are independent, so ranks never exchange data
(weak scaling: every rank evaluates its own 1024-molecule batch).  Rank 0 prints
ONE JSON line.  See DESIGN.md "Measurement" for the roofline bookkeeping.
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


WORKLOADS = {"qm9_1024": "1024 QM9-shape synthetic molecules", "qm9_64": "64 QM9-shape synthetic molecules",
             "qm9_8192": "8192 QM9-shape synthetic molecules (the per-GPU share of QM9-65k on 8 GPUs, SURVEY 8d-5)",
             "md17_4096": "4096 perturbed aspirin frames (MD17 shape)", "water_512": "one periodic box of 512 water molecules"}


def make_workload(name: str, seed: int):
    from oracle import xpainn_oracle as orc  # synthetic-input generators only (pure numpy)

    if name == "qm9_1024":
        pos, z, ptr = orc.synth_qm9_batch(1024, seed=seed)
        return pos, z, ptr, None
    if name in ("qm9_64", "qm9_8192"):
        pos, z, ptr = orc.synth_qm9_batch(int(name.split("_")[1]), seed=seed)
        return pos, z, ptr, None
    if name == "md17_4096":
        p0, z0, _ = orc.synth_aspirin()
        rng = np.random.default_rng(11 + seed)
        pos = (p0[None] + rng.normal(0, 0.05, size=(4096, 21, 3))).reshape(-1, 3)
        return pos, np.tile(z0, 4096), np.arange(0, 4097 * 21, 21, dtype=np.int64)[:4097], None
    if name == "water_512":
        pos, z, ptr, cell = orc.synth_water_box(8, seed=5 + seed)
        return pos, z, ptr, cell
    raise ValueError(name)


def cpu_baseline(pos, z, ptr, sd, budget_s=20.0, n_mol=96):
    """The oracle (a CPU restatement of the reference's eager op graph: index_select ->
    Linear -> elementwise -> index_add -> autograd.grad) timed on this host, fp32, on the
    first `n_mol` molecules of the SAME batch, brute-force neighbour list included."""
    from oracle import xpainn_oracle as orc

    n_mol = min(n_mol, len(ptr) - 1)
    a = int(ptr[n_mol])
    p, zz, pp = pos[:a].astype(np.float32), z[:a], ptr[: n_mol + 1]
    sd32 = {k: (v.float().cpu() if v.is_floating_point() else v.cpu()) for k, v in sd.items()}
    oracle = orc.XPaiNNOracle(sd32)
    threads = torch.get_num_threads()

    def one():
        ei = orc.radius_graph_canonical(p, pp, 5.0)
        batch = np.repeat(np.arange(n_mol), np.diff(pp))
        out = oracle({"pos": torch.tensor(p), "atomic_numbers": torch.tensor(zz.astype(np.int64)), "edge_index": torch.tensor(ei),
                      "batch": torch.tensor(batch), "ptr": torch.tensor(pp)})
        return ei.shape[1], out

    n_edges, _ = one()  # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 10 and (time.perf_counter() < t_end or len(times) < 2):
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": n_edges / med, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"first {n_mol} molecules of the batch ({a} atoms, {n_edges} edges), fp32, median of {len(times)} runs, "
                      f"{med * 1e3:.1f} ms/eval, brute-force neighbour list included"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="qm9_1024", choices=list(WORKLOADS))
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
    args = ap.parse_args()

    from xequinet_amd import dist as xdist
    from xequinet_amd import ops
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.nn import resolve_model

    rank, local_rank, world = xdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dtype = torch.float32 if args.dtype == "f32" else torch.float64

    torch.manual_seed(0)
    model = resolve_model("xpainn").eval().requires_grad_(False)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dtype).to(dev)

    # every rank gets its own batch of the same shape (weak scaling), resident in HBM
    pos, z, ptr, cell = make_workload(args.workload, seed=1234 + rank)
    pos_d = torch.tensor(pos, dtype=dtype, device=dev)
    z_d = torch.tensor(z, device=dev)
    ptr_d = torch.tensor(ptr, device=dev)
    cell_d = None if cell is None else torch.tensor(cell, dtype=dtype, device=dev)
    pbc_d = None if cell is None else torch.tensor([[True, True, True]], device=dev)
    transform = NeighborTransform(model.cutoff_radius)
    if not args.no_gemm_autotune:
        from xequinet_amd.tuning import enable_gemm_autotune
        enable_gemm_autotune(results_file=args.gemm_results)   # every GEMM shape is timed once, during the warm-up steps

    def step_eager():
        batch = XequiBatch(pos_d.detach(), z_d, ptr_d, pbc=pbc_d, cell=cell_d)
        batch = transform(batch)                       # HIP radius graph
        with torch.enable_grad():
            out = model(batch.to_dict(), compute_forces=True, compute_virial=False)
        return batch.edge_index.shape[1], out

    if args.eager:
        step = step_eager
    else:
        # the same kernels in the same order as one HIP-graph launch (results bitwise those of the eager path); the
        # neighbour list stays eager: its edge count has to reach the host to size the edge arrays
        from xequinet_amd.runtime import GraphedModel
        graphed = GraphedModel(model, compute_forces=True, compute_virial=False, tune_gemms=False)

        def step():
            batch = XequiBatch(pos_d.detach(), z_d, ptr_d, pbc=pbc_d, cell=cell_d)
            batch = transform(batch)
            return batch.edge_index.shape[1], graphed(batch.to_dict())

    try:
        # set-up, not a timed or counted step: the first evaluation times the library GEMM candidates (TunableOp) and
        # captures the HIP graph; the W warm-up steps and the K timed steps that follow are all plain steps
        n_edges, out = step()
        for _ in range(args.warmup):
            n_edges, out = step()
        torch.cuda.synchronize()
    except RuntimeError as err:      # a failed capture must not cost the measurement: fall back to host launches
        if args.eager:
            raise
        print(f"[bench] HIP-graph capture failed on rank {rank} ({err}); continuing with host launches", file=sys.stderr, flush=True)
        args.eager, step = True, step_eager
        for _ in range(args.warmup + 1):
            n_edges, out = step()
        torch.cuda.synchronize()
    if world > 1:                    # every rank must time the same mode
        flag = torch.tensor([1 if args.eager else 0], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if bool(flag.item()) and not args.eager:
            args.eager, step = True, step_eager
    n_atoms = pos_d.shape[0]

    # ---- timed region: exactly K steps between barrier + synchronize
    ops.KERNEL_TIMER.reset(enabled=args.eager)
    xdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges_done = 0
    for _ in range(args.steps):
        n_edges, out = step()
        edges_done += n_edges
    torch.cuda.synchronize()
    xdist.barrier()
    elapsed = time.perf_counter() - t0
    if args.eager:
        kernel_ms = ops.KERNEL_TIMER.summary()       # HIP events recorded on the launch stream, in the timed region
        kernel_timing = "HIP events around every launch of the timed region, on the launch stream"
    else:
        # HIP offers no per-node event timing inside a graph launch on this stack (torch: "External events are disallowed
        # in rocm"), so the message kernels' launch durations are read right after the timed region: the same kernels on
        # the same inputs, launched from the host between HIP events on the launch stream
        out_keep = {k: v.clone() for k, v in out.items()}
        ops.KERNEL_TIMER.reset(enabled=True)
        cal = max(1, min(args.steps, 10))
        for _ in range(cal):
            step_eager()
        kernel_ms = ops.KERNEL_TIMER.summary()
        kernel_ms = {k: {"launches": v["launches"], "total_ms": v["total_ms"] * args.steps / cal} for k, v in kernel_ms.items()}
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
        #   B_bwd = 10 880 N + 40 E   (fp32 features, int64 indices), one launch per layer
        esz = 4 if dtype == torch.float32 else 8
        dom = max(kernel_ms, key=lambda k: kernel_ms[k]["total_ms"]) if kernel_ms else None
        alg_fwd = (2272 * esz) * n_atoms + (16 + 3 * esz) * n_edges   # B_fwd
        alg_bwd = (2720 * esz) * n_atoms + (16 + 6 * esz) * n_edges   # B_bwd
        roofline = None
        if dom is not None:
            alg = {dom: alg_bwd if "bwd" in dom else alg_fwd}
            avg_ms = kernel_ms[dom]["total_ms"] / kernel_ms[dom]["launches"]
            if not args.eager:
                avg_ms = avg_ms * cal / args.steps      # undo the per-step rescaling above: a plain average over the launches
            achieved = alg[dom] / (avg_ms * 1e-3) / 1e9
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tfile):
                traffic = json.load(open(tfile)).get(args.workload, {}).get(dom)
            # the same launch against the f32 matrix pipe (~26 kFLOP per edge-layer forward, filter
            # [576 x 21] + gating; the reverse pass evaluates the filter and its d/dd): what the kernel is nearer to
            flops = (52.0e3 if "bwd" in dom else 26.0e3) * n_edges
            tfl = flops / (avg_ms * 1e-3) / 1e12
            roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": avg_ms, "timing": kernel_timing,
                        "algorithmic_bytes_per_launch": alg[dom],
                        "matrix_pipe": {"algorithmic_flops_per_launch": flops, "achieved": tfl, "peak": F32_MATRIX_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": tfl / F32_MATRIX_PEAK_TFLOPS, "dtype": "f32 (exact, v_mfma_f32_32x32x2_f32)"},
                        "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in kernel_ms.items()}}
        line = {
            "metric": METRIC, "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload]} per GPU, 5 A cutoff, default XPaiNN (865141 params, random init), "
                                   "neighbour list + energy + forces", "atoms_per_gpu": int(n_atoms), "edges_per_gpu": int(n_edges),
                       "parallelism": f"molecule shards x{world}, no collectives",
                       "library_gemm_selection": "default heuristics" if args.no_gemm_autotune else "timed once per shape in warm-up (TunableOp)",
                       "launch": "host launch per kernel" if args.eager else "model part (forward + force backward) as one captured HIP graph per step; neighbour list launched from the host"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pos, z, ptr, sd)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

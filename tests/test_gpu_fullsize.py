"""Full-size parity of the BASELINE.json configurations against the fp64 oracle (pytest -m gpu, on the MI355X box).

The HIP path runs each configuration AT FULL SIZE in fp32; the fp64 CPU oracle is then run on a random subset of the
molecules / frames of the very same batch (the whole box for the periodic configuration) and explicit bounds are
asserted on exactly those: |dE| <= 1e-5 |E| + 1e-4; |dF| within max(1e-4, 1.5 x the error of the fp32 oracle on the same
molecules), at the maximum and at the 99th percentile (tests/test_gpu_parity.py::f32_force_bounds: the HIP path is no worse
than the reference's own arithmetic at the reference's own precision).  Molecules of a batch do not interact, so a
molecule's oracle result does not depend on which other molecules the oracle sees.

Every comparison appends its achieved maxima to PARITY; tests/conftest.py writes them to
``gpurun_out/parity_r06.json`` at the end of the session (copied to ``profiles/``).
"""
import numpy as np
import pytest
import torch

from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn

from tests import parity_record
from tests.test_gpu_parity import DEV, _build, _t, f32_force_bounds

pytestmark = pytest.mark.gpu

E_RTOL, E_ATOL = 1e-5, 1e-4
F32_PLAIN_FORCE_TOL = 1e-4     # max-abs force tolerance in fp32 (model units), no envelope
N_SAMPLE = 160   # molecules / frames of the property checks further down (sharding, replay, other initialisations)
# test_full_size_against_oracle (round-5 review, item 3): the fp64 oracle evaluates EVERY molecule of the headline configuration and
# 1 024 molecules / frames of the larger ones, in chunks of ORACLE_CHUNK molecules (molecules do not interact); the error distribution
# is heavy-tailed, the maximum over a 16 % sample is not the maximum
N_COMPARE = {"qm9_1024": 1024, "md17_4096": 1024, "qm9_8192": 1024, "qm9_8192_chunked": 1024}
ORACLE_CHUNK = 256
# forces (model units): tests/test_gpu_parity.py::f32_force_bounds


def _hip_eval(model, pos, z, ptr, cell=None, chunked=False):
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.runtime import evaluate_in_chunks

    if chunked:
        out = evaluate_in_chunks(model, _t(pos, torch.float32), _t(z), _t(ptr), ptr_host=ptr, max_edges=chunked)
        return (out["energy"].cpu().double().numpy(), out["forces"].cpu().double().numpy(), out["n_edges"], None)
    kw = {} if cell is None else dict(pbc=torch.tensor([[True, True, True]], device=DEV), cell=_t(cell, torch.float32))
    b = NeighborTransform(5.0)(XequiBatch(_t(pos, torch.float32), _t(z), _t(ptr), **kw))
    with torch.enable_grad():
        out = model(b.to_dict(), compute_forces=True)
    return out["energy"].detach().cpu().double().numpy(), out["forces"].cpu().double().numpy(), b.edge_index.shape[1], b


def _oracle_subset(oracle, pos, z, ptr, mols):
    """fp64 oracle on the molecules `mols` of the batch (positions rounded to fp32 first: the values the GPU saw)."""
    idx = np.concatenate([np.arange(ptr[g], ptr[g + 1]) for g in mols])
    p = pos[idx].astype(np.float32)
    pp = np.concatenate([[0], np.cumsum(np.diff(ptr)[mols])]).astype(np.int64)
    ei = orc.radius_graph_canonical(p, pp, 5.0)
    batch = np.repeat(np.arange(len(mols)), np.diff(pp))
    ref_in = {"pos": torch.tensor(p.astype(np.float64)), "atomic_numbers": torch.tensor(z[idx].astype(np.int64)),
              "edge_index": torch.tensor(ei), "batch": torch.tensor(batch), "ptr": torch.tensor(pp)}
    want = oracle(ref_in, compute_forces=True)
    return idx, want["energy"].numpy(), want["forces"].numpy(), ei.shape[1], ref_in


def _compare(name, E, F, Eref, Fref, extra, oracle, ref_in):
    dE, dF = np.abs(E - Eref), np.abs(F - Fref)
    bounds = f32_force_bounds(oracle, ref_in, Fref)
    b_max, b_p99, e32_max, e32_p99 = bounds
    rec = dict(config=name, max_abs_dE=float(dE.max()), max_dE_over_bound=float((dE / (E_RTOL * np.abs(Eref) + E_ATOL)).max()),
               max_abs_dF=float(dF.max()), max_abs_F=float(np.abs(Fref).max()), max_abs_E=float(np.abs(Eref).max()),
               p99_abs_dF=float(np.quantile(dF, 0.99)), p999_abs_dF=float(np.quantile(dF, 0.999)),
               bound_dE=f"{E_RTOL}*|E|+{E_ATOL}", **bounds.record(),
               dtype="f32 HIP vs f64 oracle", **extra)
    parity_record.add(rec)
    assert np.all(dE <= E_RTOL * np.abs(Eref) + E_ATOL), rec
    assert dF.max() <= b_max and np.quantile(dF, 0.99) <= b_p99, rec


def _compare_chunked(name, E, F, pos, z, ptr, mols, oracle, extra):
    """The comparison of ``_compare`` over MANY molecules: the fp64 oracle and the fp32 envelope (two CPU-oracle edge orders and sixteen
    ATen-only GPU orders per chunk) walk the molecules in chunks, the per-atom errors and envelopes are gathered, and ONE maximum and
    ONE 99th percentile are asserted over all of them -- the HIP maximum over the whole compared set against the envelope of the whole
    compared set."""
    from tests.test_gpu_parity import bounds_from_envelopes

    dE_all, Eref_all, dF_all, Fref_all, ecpu, egpu, atoms, edges, members = [], [], [], [], [], [], 0, 0, None
    for c0 in range(0, len(mols), ORACLE_CHUNK):
        part = mols[c0 : c0 + ORACLE_CHUNK]
        idx, Eref, Fref, e_sub, ref_in = _oracle_subset(oracle, pos, z, ptr, part)
        b = f32_force_bounds(oracle, ref_in, Fref, cpu_members=2)
        dE_all.append(np.abs(E[part] - Eref)); Eref_all.append(Eref)
        dF_all.append(np.abs(F[idx] - Fref)); Fref_all.append(Fref)
        ecpu.append(b.err_cpu); egpu.append(b.err_gpu)
        atoms += len(idx); edges += e_sub; members = b.members
    dE, Eref, dF, Fref = (np.concatenate(a) for a in (dE_all, Eref_all, dF_all, Fref_all))
    bounds = bounds_from_envelopes(np.concatenate(ecpu), np.concatenate(egpu), members)
    b_max, b_p99, e32_max, e32_p99 = bounds
    rec = dict(config=name, max_abs_dE=float(dE.max()), max_dE_over_bound=float((dE / (E_RTOL * np.abs(Eref) + E_ATOL)).max()),
               max_abs_dF=float(dF.max()), max_abs_F=float(np.abs(Fref).max()), max_abs_E=float(np.abs(Eref).max()),
               p99_abs_dF=float(np.quantile(dF, 0.99)), p999_abs_dF=float(np.quantile(dF, 0.999)),
               frac_atoms_within_plain_tol=float((dF.max(axis=1) <= F32_PLAIN_FORCE_TOL).mean()),
               bound_dE=f"{E_RTOL}*|E|+{E_ATOL}", **bounds.record(), dtype="f32 HIP vs f64 oracle", compared_atoms=int(atoms),
               compared_edges=int(edges), oracle_chunk_molecules=ORACLE_CHUNK, **extra)
    parity_record.add(rec)
    assert np.all(dE <= E_RTOL * np.abs(Eref) + E_ATOL), rec
    assert dF.max() <= b_max and np.quantile(dF, 0.99) <= b_p99, rec


@pytest.mark.parametrize("name,n_mol,n_atoms,n_edges,chunked", [
    ("qm9_1024", 1024, 18609, 311994, False),
    ("md17_4096", 4096, 86016, None, False),
    ("qm9_8192", 8192, None, None, False),            # the per-GPU share of the 65k batch on 8 GPUs
    ("qm9_8192_chunked", 8192, None, None, 600_000),  # the same batch through evaluate_in_chunks (5 chunks)
])
def test_full_size_against_oracle(name, n_mol, n_atoms, n_edges, chunked):
    model, oracle = _build(torch.float32)
    pos, z, ptr, _ = syn.make_workload(name.replace("_chunked", ""), seed=1234)
    assert len(ptr) - 1 == n_mol and (n_atoms is None or len(pos) == n_atoms)
    E, F, e_hip, _ = _hip_eval(model, pos, z, ptr, chunked=chunked)
    assert n_edges is None or e_hip == n_edges
    assert E.shape == (n_mol,) and F.shape == (len(pos), 3) and np.isfinite(E).all() and np.isfinite(F).all()
    n_cmp = min(n_mol, N_COMPARE[name])
    mols = np.arange(n_mol) if n_cmp == n_mol else np.sort(np.random.default_rng(7).choice(n_mol, size=n_cmp, replace=False))
    _compare_chunked(name, E, F, pos, z, ptr, mols, oracle,
                     dict(atoms=int(len(pos)), edges=int(e_hip), graphs=int(n_mol), compared_graphs=int(len(mols))))


@pytest.mark.parametrize("depth", [1, 2])
def test_graphed_chunks_equal_one_evaluation_bit_for_bit(depth, monkeypatch):
    """runtime.GraphedChunks: a 1536-molecule batch as three chunks of ~9 200 atoms, every chunk a whole captured step (its neighbour
    list inside, nothing read back), one at a time and two in flight: energies, forces and the device-side edge count are those of ONE
    evaluation of the batch, bit for bit, call after call; chunks on both sides of the node-block threshold are refused."""
    from xequinet_amd import runtime

    model, _ = _build(torch.float32)
    pos, z, ptr = syn.synth_qm9_batch(1536, seed=99)
    E, F, e1, _ = _hip_eval(model, pos, z, ptr)
    batch = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    gc = runtime.GraphedChunks(model, ptr, max_edges=200_000, depth=depth)
    assert gc.n_chunks == 3 and gc.depth == depth
    args = (_t(pos, torch.float32), _t(z), _t(ptr), _t(batch))
    for rep in range(3):
        out = gc(*args)
        assert np.array_equal(out["energy"].cpu().double().numpy(), E) and np.array_equal(out["forces"].cpu().double().numpy(), F)
    assert int(gc.edge_total) == 3 * e1 and not gc.overflowed()
    assert all(st.captures == 1 for st in gc.steps)
    monkeypatch.setenv("XEQ_NODE_BLOCK_MIN_NODES", "9200")       # chunks of 9 220 / 9 123 / 9 257 atoms: one would take the chain of small kernels
    with pytest.raises(ValueError, match="threshold"):
        runtime.GraphedChunks(model, ptr, max_edges=200_000)


@pytest.mark.parametrize("node_block,n_mol,max_edges", [("by size", 1024, 200_000), ("always", 512, 40_000), ("never", 512, 40_000)])
def test_chunked_equals_unchunked(node_block, n_mol, max_edges, monkeypatch):
    """runtime.evaluate_in_chunks (what a rank does with a shard above the kernels' 32-bit bound) against ONE evaluation
    of the same batch: identical edge count, and identical BITS.  Every kernel of the f32 path is this library's own since
    round 3 (dot_lin, the embedding and the energy head were library GEMMs, which pick another kernel -- another summation
    order -- for another row count: 1.2e-3 between two batchings of an ill-conditioned molecule then) and gives a node /
    graph the same sums in the same order in any batch.  As for shards (test_sharded_equals_unsharded): the fused node-block
    launches and the chain they replace round differently, so the bits agree while every chunk sits on the batch's side of
    xeq_node_block_auto's threshold ("by size": 1024 molecules in two chunks of ~9 300 atoms, like the multi-million-edge chunks of real use) or
    with the choice pinned for the job."""
    if node_block == "always":
        monkeypatch.setenv("XEQ_NODE_BLOCK_MIN_NODES", "0")
    elif node_block == "never":
        monkeypatch.setenv("XEQ_NODE_BLOCK", "0")
    model, _ = _build(torch.float32)
    pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=99)
    E, F, e1, _ = _hip_eval(model, pos, z, ptr)
    Ec, Fc, e2, _ = _hip_eval(model, pos, z, ptr, chunked=max_edges)
    assert e1 == e2
    parity_record.add(dict(config=f"qm9_{n_mol} chunked (max {max_edges} edges, node block {node_block}) vs one evaluation", max_abs_dE=float(np.abs(E - Ec).max()),
                           max_abs_dF=float(np.abs(F - Fc).max()), bitwise=bool(np.array_equal(E, Ec) and np.array_equal(F, Fc))))
    assert np.array_equal(Ec, E) and np.array_equal(Fc, F), (np.abs(E - Ec).max(), np.abs(F - Fc).max())


@pytest.mark.parametrize("node_block", ["by size", "always", "never"])
def test_sharded_equals_unsharded(node_block, monkeypatch):
    """BASELINE config 5 in small: the batch cut by dist.shard_by_edges for 1/2/4/8 ranks, every shard evaluated on
    this one GPU, results concatenated in rank order == the unsharded evaluation BIT FOR BIT (no collective: ranks are
    independent; no library GEMM left whose pick depends on the shard's row count).

    The fused node-block launches (round 4) and the chain of small kernels they replace round differently, and
    `xeq_node_block_auto` chooses between them by the node count of an evaluation: the bits are those of the unsharded run as long
    as every shard sits on the same side of that threshold ("by size": all below it here), or when the choice is pinned for the
    job -- XEQ_NODE_BLOCK_MIN_NODES=0 ("always": every shard through the fused launches, which give a node the same sums in the
    same order wherever it sits) or XEQ_NODE_BLOCK=0 ("never")."""
    from xequinet_amd import dist as xdist

    if node_block == "always":
        monkeypatch.setenv("XEQ_NODE_BLOCK_MIN_NODES", "0")
    elif node_block == "never":
        monkeypatch.setenv("XEQ_NODE_BLOCK", "0")
    model, _ = _build(torch.float32)
    pos, z, ptr = syn.synth_qm9_batch(256, seed=5)
    E, F, e_all, _ = _hip_eval(model, pos, z, ptr)
    for world in (2, 4, 8):
        Es, Fs, edges = [], [], 0
        for g0, g1 in xdist.shard_by_edges(ptr, world):
            p, zz, pp = xdist.take_shard(pos, z, ptr, g0, g1)
            e, f, ne, _ = _hip_eval(model, p, zz, pp)
            Es.append(e), Fs.append(f)
            edges += ne
        Es, Fs = np.concatenate(Es), np.concatenate(Fs)
        assert edges == e_all and Es.shape == E.shape and Fs.shape == F.shape
        parity_record.add(dict(config=f"qm9_256 sharded x{world} vs unsharded (node block {node_block})", max_abs_dE=float(np.abs(E - Es).max()),
                               max_abs_dF=float(np.abs(F - Fs).max()), bitwise=bool(np.array_equal(E, Es) and np.array_equal(F, Fs))))
        assert np.array_equal(Es, E) and np.array_equal(Fs, F), (world, np.abs(E - Es).max(), np.abs(F - Fs).max())


def test_qm9_65536_on_one_gpu_in_chunks():
    """BASELINE config 5 at FULL size on one GPU (what a single rank does with the whole 65 536-molecule batch: 1.18 M atoms, 19.6 M
    edges, above the message kernels' 32-bit offsets, so runtime.evaluate_in_chunks walks it in three ranges).  The oracle cannot
    finish this size, so the checks are the size-independent properties of the path: every number finite, the edge count of the
    batch's own generator (the figure bench.py reports for the same seed), no net force on any molecule (translation invariance of
    E(pos): sum_i F_i = 0 per graph), and the first molecules' results equal to those of the same molecules evaluated alone."""
    from xequinet_amd.runtime import evaluate_in_chunks

    model, _ = _build(torch.float32)
    pos, z, ptr, _ = syn.make_workload("qm9_65536", seed=1234)
    assert len(ptr) - 1 == 65536 and len(pos) == 1178952
    out = evaluate_in_chunks(model, _t(pos, torch.float32), _t(z), _t(ptr), ptr_host=ptr)
    E, F = out["energy"], out["forces"]
    assert out["n_chunks"] >= 3 and out["n_edges"] == 19617584, (out["n_chunks"], out["n_edges"])
    assert E.shape == (65536,) and F.shape == (len(pos), 3)
    assert bool(torch.isfinite(E).all()) and bool(torch.isfinite(F).all())
    graph = torch.repeat_interleave(torch.arange(65536, device=DEV), _t(np.diff(ptr)))
    net = torch.zeros(65536, 3, device=DEV, dtype=torch.float64).index_add_(0, graph, F.double())
    scale = float(F.abs().max())
    parity_record.add(dict(config="qm9_65536 on one GPU in chunks (properties)", atoms=int(len(pos)), edges=int(out["n_edges"]), chunks=int(out["n_chunks"]),
                           max_abs_net_force_per_molecule=float(net.abs().max()), max_abs_F=scale))
    assert float(net.abs().max()) <= 2e-4 * max(1.0, scale), float(net.abs().max())
    # the first 48 molecules alone (another batch, another kernel policy by size: fp32 agreement, not bits)
    g = 48
    a = int(ptr[g])
    E1, F1, _, _ = _hip_eval(model, pos[:a], z[:a], ptr[: g + 1])
    dE, dF = np.abs(E[:g].cpu().double().numpy() - E1), np.abs(F[:a].cpu().double().numpy() - F1)
    assert np.all(dE <= E_RTOL * np.abs(E1) + E_ATOL) and dF.max() <= 2e-3, (dE.max(), dF.max())


def test_water_512_whole_box_against_oracle():
    """BASELINE config 4 (periodic, ~51 neighbours per atom) at full size: the whole box through the fp64 oracle, on the
    edge list the HIP neighbour search produced (itself bit-exact against the reference's order on the golden boxes)."""
    model, oracle = _build(torch.float32)
    pos, z, ptr, cell = syn.make_workload("water_512", seed=0)
    E, F, n_edges, b = _hip_eval(model, pos, z, ptr, cell=cell)
    assert len(pos) == 1536 and 45 * 1536 < n_edges < 57 * 1536
    ei = b.edge_index.cpu().numpy()
    co = b.cell_offsets.cpu().double().numpy()
    p32 = pos.astype(np.float32).astype(np.float64)
    c32 = cell.astype(np.float32).astype(np.float64)
    ref_in = {"pos": torch.tensor(p32), "atomic_numbers": torch.tensor(z.astype(np.int64)), "edge_index": torch.tensor(ei),
              "batch": torch.zeros(len(pos), dtype=torch.long), "ptr": torch.tensor(ptr), "cell": torch.tensor(c32),
              "cell_offsets": torch.tensor(co)}
    want = oracle(ref_in, compute_forces=True)
    _compare("water_512", E, F, want["energy"].numpy(), want["forces"].numpy(),
             dict(atoms=1536, edges=int(n_edges), graphs=1, compared_graphs=1, compared_atoms=1536, compared_edges=int(n_edges)),
             oracle, ref_in)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver starts N = 1) must start its two ranks itself,
    before anything touches the GPU, and rank 0 prints the one JSON line.  Rehearsed on this one card: both ranks on
    device 0, gloo for the two reductions (XEQ_BENCH_BACKEND / XEQ_BENCH_DEVICE are bench.py's rehearsal switches)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(XEQ_BENCH_BACKEND="gloo", XEQ_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-gemm-autotune"], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["edges_all_ranks_per_step"] > 1.9 * line["config"]["edges_rank0"]


def test_bench_shards_one_batch_over_four_ranks_on_one_card():
    """`bench.py --workload qm9_8192 --gpus 4` (one 8192-molecule batch -- the per-GPU share of BASELINE config 5 -- cut by molecule over
    four ranks, strong scaling) rehearsed on this one card over gloo: every rank on device 0, the two reductions of the contract
    (max time, sum of edges) across four processes.  (The pool allows six GPU processes at once: the eight-rank run is the driver's.)"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(XEQ_BENCH_BACKEND="gloo", XEQ_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "qm9_8192_sharded", "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-gemm-autotune"], capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["scaling"] == "strong" and line["value"] > 0
    # the four shards together hold the whole batch; rank 0's share is a quarter to a few percent (balanced on planned edges)
    total = line["config"]["edges_all_ranks_per_step"]
    assert 0.23 * total <= line["config"]["edges_rank0"] <= 0.27 * total, line["config"]


def test_whole_step_graph_replays_batches_of_changing_sizes():
    """runtime.GraphedStep: neighbour list + model as ONE captured graph over capacity-sized arrays, the edge count on the device.
    Four batches with different atom / graph / edge counts go through one capture; each result is, bit for bit, the eager
    evaluation of the same (padded) batch and of the un-padded batch through the ordinary path (no kernel of the f32 path
    depends on the row count), and the device-side edge count is the list's true length."""
    from xequinet_amd import keys, ops
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.runtime import GraphedStep, pair_capacity

    model, _ = _build(torch.float32)
    draws = [syn.synth_qm9_batch(n, seed=s) for n, s in ((40, 1), (33, 2), (40, 3), (37, 4))]
    n_cap = max(len(p) for p, _, _ in draws) + 7
    g_cap = max(len(t) - 1 for _, _, t in draws)
    e_cap = max(pair_capacity(t) for _, _, t in draws)
    step = GraphedStep(model, (n_cap, g_cap, e_cap))
    seen = set()
    for pos, z, ptr in draws:
        n, g = len(pos), len(ptr) - 1
        out = step(_t(pos, torch.float32), _t(z), _t(ptr), ptr_host=ptr)
        E, F, ne = out["energy"].clone(), out["forces"].clone(), int(out["n_edges"].item())
        b = NeighborTransform(5.0)(XequiBatch(_t(pos, torch.float32), _t(z), _t(ptr)))
        assert ne == b.edge_index.shape[1] and E.shape == (g,) and F.shape == (n, 3)
        assert torch.equal(step.edge_index[:, :ne], b.edge_index)
        seen.add((n, ne))
        # eager evaluation of the padded batch (the step's own static buffers): bit for bit
        rowptr, _ = ops.radius_graph_capacity(step.pos, step.ptr, 5.0, step.edge_index)
        eg = ops.EdgeGraph(step.edge_index, step.n_atoms, center_sorted=True, ptr=step.ptr, c_rowptr=rowptr, symmetric=True)
        with torch.enable_grad():
            want = model({keys.POSITIONS: step.pos.detach().clone(), keys.ATOMIC_NUMBERS: step.z, keys.EDGE_INDEX: step.edge_index,
                          keys.BATCH: step.batch, keys.BATCH_PTR: step.ptr, keys.EDGE_GRAPH: eg}, compute_forces=True)
        assert torch.equal(E, want["energy"].detach()[:g]) and torch.equal(F, want["forces"][:n])
        # the un-padded batch through the ordinary path
        with torch.enable_grad():
            plain = model(b.to_dict(), compute_forces=True)
        assert torch.equal(E, plain["energy"].detach()) and torch.equal(F, plain["forces"])
    assert step.captures == 1 and len(seen) == 4
    with pytest.raises(ValueError):
        pos, z, ptr = syn.synth_qm9_batch(g_cap + 1, seed=9)
        step(_t(pos, torch.float32), _t(z), _t(ptr))


def test_capacity_list_that_outgrows_its_arrays_becomes_empty_and_reports_the_true_count():
    """The open-boundary list in its capacity form (ops.radius_graph_capacity, runtime.GraphedStep): a list that outgrows the edge
    arrays is replaced by an EMPTY list -- the row pointer never points behind the buffers and no symmetric shortcut sees a cut list,
    so no downstream kernel walks out of bounds -- and the true count comes back next to it (advisor, round 3: the row pointer used
    to be left unguarded)."""
    from xequinet_amd import ops
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.nn import resolve_model
    from xequinet_amd.runtime import GraphedStep, pair_capacity

    pos, z, ptr = syn.synth_qm9_batch(24, seed=3)
    full = pair_capacity(ptr)
    b = NeighborTransform(5.0)(XequiBatch(_t(pos, torch.float32), _t(z), _t(ptr)))
    true_edges = b.edge_index.shape[1]
    cap = true_edges // 2
    edge_index = torch.zeros((2, cap), dtype=torch.int64, device=DEV)
    rowptr, count = ops.radius_graph_capacity(_t(pos, torch.float32), _t(ptr), 5.0, edge_index)
    assert int(count.item()) == true_edges and int(rowptr.abs().max().item()) == 0
    big = torch.zeros((2, true_edges + 7), dtype=torch.int64, device=DEV)
    rowptr2, count2 = ops.radius_graph_capacity(_t(pos, torch.float32), _t(ptr), 5.0, big)
    assert int(count2.item()) == true_edges == int(rowptr2[-1].item()) and torch.equal(big[:, :true_edges], b.edge_index)
    # the whole step on such a capacity: finite results of the empty list, nothing faults, and the step says so
    model = resolve_model("xpainn", action_blocks=2).to(DEV).eval().requires_grad_(False)
    step = GraphedStep(model, (len(pos), len(ptr) - 1, cap))
    out = step(_t(pos, torch.float32), _t(z), _t(ptr))
    assert torch.isfinite(out["energy"]).all() and torch.isfinite(out["forces"]).all() and step.overflowed()
    step_ok = GraphedStep(model, (len(pos), len(ptr) - 1, full))
    step_ok(_t(pos, torch.float32), _t(z), _t(ptr))
    assert not step_ok.overflowed()


def test_whole_step_graph_follows_a_weight_update():
    """runtime.GraphedStep captures behind warm-up runs that already filled the packed-weight caches: the pack kernels are not in
    the graph.  It now tracks the parameters' version counters (and the pack epoch) and re-captures when they moved (advisor,
    round 3: replays used to mix live weights with stale packed copies)."""
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.nn import resolve_model
    from xequinet_amd.runtime import GraphedStep, pair_capacity

    torch.manual_seed(5)
    model = resolve_model("xpainn", action_blocks=2).to(DEV).eval().requires_grad_(False)
    pos, z, ptr = syn.synth_qm9_batch(16, seed=4)
    step = GraphedStep(model, (len(pos), len(ptr) - 1, pair_capacity(ptr)))
    args = (_t(pos, torch.float32), _t(z), _t(ptr))
    before = step(*args)["energy"].clone()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.02)
    after = step(*args)["energy"].clone()
    assert step.captures == 2 and not torch.equal(before, after)
    b = NeighborTransform(5.0)(XequiBatch(*args))
    with torch.enable_grad():
        want = model(b.to_dict(), compute_forces=True)
    assert torch.equal(after, want["energy"].detach())


def test_qm9_1024_with_the_reference_initialisation():
    """BASELINE.md section 2 states |dF| <= 1e-4 (fp32).  The parity model of the other checks randomises every affine weight and bias
    (so that a wrong index shows up); this one carries the REFERENCE'S OWN INITIALISATION (nn.Linear / LayerNorm defaults, o3.Linear
    ~ N(0, 1), zero o3 biases, affine weights 1: SURVEY 8d).  What round 4 found (profiles/r04_fp32_tail.txt): a random-weight
    network of this family is ill-conditioned on a few molecules whatever the initialisation -- there ANY fp32 evaluation (the CPU
    oracle in another edge order, ATen on the GPU, these kernels) draws errors 10-50 x apart, and a change of one ulp in one
    envelope value moves a force by 1e-5 -- so the plain tolerance is asserted where it is a property of the implementation, at the
    99.9th percentile of the components (measured: 99 % below 2e-5, rms 6e-6), and the single worst component against the
    reference's own fp32 error on the same molecules, as everywhere else."""
    from xequinet_amd.nn import resolve_model

    torch.manual_seed(0)
    model = resolve_model("xpainn").eval().requires_grad_(False)
    oracle = orc.XPaiNNOracle({k: v.detach().double().clone() for k, v in model.state_dict().items()})
    model = model.to(DEV)
    pos, z, ptr, _ = syn.make_workload("qm9_1024", seed=1234)
    E, F, n_edges, _ = _hip_eval(model, pos, z, ptr)
    assert n_edges == 311994
    mols = np.sort(np.random.default_rng(11).choice(len(ptr) - 1, size=N_SAMPLE, replace=False))
    idx, Eref, Fref, _, ref_in = _oracle_subset(oracle, pos, z, ptr, mols)
    dE, dF = np.abs(E[mols] - Eref), np.abs(F[idx] - Fref)
    bounds = f32_force_bounds(oracle, ref_in, Fref)
    b_max = bounds[0]
    parity_record.add(dict(config="qm9_1024, reference initialisation", max_abs_dE=float(dE.max()), max_abs_dF=float(dF.max()),
                           p99_abs_dF=float(np.quantile(dF, 0.99)), p999_abs_dF=float(np.quantile(dF, 0.999)), rms_dF=float(np.sqrt((dF ** 2).mean())),
                           max_abs_F=float(np.abs(Fref).max()), bound_dF_p999=F32_PLAIN_FORCE_TOL, **bounds.record(), compared_graphs=int(len(mols)),
                           compared_atoms=int(len(idx)), dtype="f32 HIP vs f64 oracle"))
    assert np.all(dE <= E_RTOL * np.abs(Eref) + E_ATOL)
    assert np.quantile(dF, 0.999) <= F32_PLAIN_FORCE_TOL, float(np.quantile(dF, 0.999))
    assert dF.max() <= b_max, (float(dF.max()), b_max)


def test_qm9_1024_well_conditioned_model_meets_the_plain_tolerance():
    """BASELINE.md section 2's plain figure, max |dF| <= 1e-4 in fp32, with NO envelope, at full size.

    The fp32 tail of the other checks sits at one spot of the reference's function (profiles/r04_fp32_tail.txt): ``Invariant``'s
    sqrt(sum_m V^2 + eps^2) - eps with eps = 1e-5 (nn/o3layer.py:39-44) on the 0e channels of V = update_V(xhat) (nn/xpainn.py:213-216),
    where V is a SCALAR that crosses zero: among 128 channels x 18 609 atoms x 3 blocks a few hundred land within 1e-4 of zero, and
    there the derivative V / sqrt(V^2 + eps^2) is decided by the low bits of V in any fp32 evaluation.  This model has the same
    architecture and random weights everywhere else, but the 0e block of every ``update_V`` is bounded away from zero BY CONSTRUCTION
    (weights scaled by 0.02, biases of magnitude 1 .. 1.5 with random signs: |V_0e| >= ~0.8), so the function itself is well
    conditioned and the stated tolerance is a property of the implementation: asserted on every force component of the compared
    molecules."""
    model, oracle = _build(torch.float32)
    g = torch.Generator().manual_seed(5)
    sd = model.state_dict()
    for name in sd:
        if name.endswith("update_V.weight"):
            sd[name][: 128 * 128] *= 0.02                        # flat o3.Linear weight: the l = 0 block comes first
        elif name.endswith("update_V.bias"):
            sign = torch.where(torch.rand(128, generator=g) < 0.5, -1.0, 1.0)
            sd[name].copy_((sign * (1.0 + 0.5 * torch.rand(128, generator=g))).to(sd[name]))
    model.load_state_dict(sd)
    oracle = orc.XPaiNNOracle({k: v.detach().double().cpu().clone() for k, v in model.state_dict().items()})
    pos, z, ptr, _ = syn.make_workload("qm9_1024", seed=1234)
    E, F, n_edges, _ = _hip_eval(model, pos, z, ptr)
    assert n_edges == 311994
    mols = np.sort(np.random.default_rng(13).choice(len(ptr) - 1, size=N_SAMPLE, replace=False))
    idx, Eref, Fref, _, ref_in = _oracle_subset(oracle, pos, z, ptr, mols)
    dE, dF = np.abs(E[mols] - Eref), np.abs(F[idx] - Fref)
    parity_record.add(dict(config="qm9_1024, well-conditioned model (0e block of update_V bounded away from 0), plain tolerance", max_abs_dE=float(dE.max()),
                           max_abs_dF=float(dF.max()), p99_abs_dF=float(np.quantile(dF, 0.99)), rms_dF=float(np.sqrt((dF ** 2).mean())),
                           max_abs_F=float(np.abs(Fref).max()), bound_dF_max=F32_PLAIN_FORCE_TOL, compared_graphs=int(len(mols)),
                           compared_atoms=int(len(idx)), dtype="f32 HIP vs f64 oracle, no envelope"))
    assert np.all(dE <= E_RTOL * np.abs(Eref) + E_ATOL)
    assert dF.max() <= F32_PLAIN_FORCE_TOL, float(dF.max())


@pytest.mark.parametrize("name,repeats", [("qm9_1024", 5), ("md17_4096", 3)])
def test_full_size_evaluations_repeat_bit_for_bit(name, repeats):
    """A whole evaluation run again gives the same bits, at sizes that put two or more waves of every kernel on a SIMD and several
    rounds of workgroups on a CU: every sum has a fixed order (no atomics), so anything else is a hazard.  (This is the check that
    shows a sporadic error at once; round 4's packed-fp32 finding, profiles/r04_nodeblock.txt item 9, was of this kind.)"""
    model, _ = _build(torch.float32)
    pos, z, ptr, _ = syn.make_workload(name, seed=1234)
    first = None
    for _ in range(repeats):
        E, F, _, _ = _hip_eval(model, pos, z, ptr)
        if first is None:
            first = (E, F)
        else:
            assert np.array_equal(E, first[0]) and np.array_equal(F, first[1]), (name, np.abs(F - first[1]).max())


def test_two_lanes_in_one_graph_equal_the_unsplit_step_bit_for_bit():
    """runtime.GraphedLanes: the QM9-1024 batch as two contiguous molecule ranges in parallel branches of one captured graph gives the
    energies and forces of the one-range step (runtime.GraphedStep) bit for bit, for two different batches through the same capture,
    and counts the same edges on the device."""
    from xequinet_amd import runtime

    model, _ = _build(torch.float32)
    cap = None
    draws = []
    for seed in (1234, 77):
        pos, z, ptr, _ = syn.make_workload("qm9_1024", seed=seed)
        batch = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
        draws.append((pos, z, ptr, batch))
    cap = (max(len(d[0]) for d in draws) + 64, 1024, max(runtime.pair_capacity(d[2]) for d in draws))
    one = runtime.GraphedStep(model, cap)
    two = runtime.GraphedLanes(model, cap, lanes=2)
    for pos, z, ptr, batch in draws:
        args = (_t(pos, torch.float32), _t(z), _t(ptr), _t(batch))
        a = one(*args[:3], batch=args[3])
        Ea, Fa, na = a["energy"].clone(), a["forces"].clone(), int(a["n_edges"])
        b = two(*args, ptr)
        assert torch.equal(b["energy"], Ea) and torch.equal(b["forces"], Fa)
        assert sum(int(st.outputs["n_edges"]) for st in two.steps) == na
    assert two.captures == 1 and not two.overflowed()
    assert int(two.edge_total) == int(one.edge_total)


def test_steps_in_flight_equal_one_at_a_time_bit_for_bit():
    """runtime.GraphedStepsInFlight: a stream of DIFFERENT batches with two (and three) steps in flight -- each on its own stream with
    its own buffers -- gives every batch the energies / forces / edge count of a lone runtime.GraphedStep bit for bit, in submit order;
    a ticket whose buffers were reused is refused; the device-side edge total is the sum over all submits."""
    from xequinet_amd import runtime

    model, _ = _build(torch.float32)
    draws = []
    for seed in (1234, 77, 5, 901, 42):
        pos, z, ptr, _ = syn.make_workload("qm9_256", seed=seed)
        draws.append((_t(pos, torch.float32), _t(z), _t(ptr)))
    cap = (max(d[0].shape[0] for d in draws) + 64, 256, max(runtime.pair_capacity(d[2].cpu().numpy()) for d in draws))
    one = runtime.GraphedStep(model, cap)
    want = []
    for d in draws:
        o = one(*d)
        want.append((o["energy"].clone(), o["forces"].clone(), int(o["n_edges"])))
    for depth in (2, 3):
        fl = runtime.GraphedStepsInFlight(model, cap, depth=depth)
        order = [0, 1, 2, 3, 4, 2, 0, 4, 1, 3, 3]
        tickets = []
        for k, i in enumerate(order):
            tickets.append(fl.submit(*draws[i]))
            if k >= depth - 1:                               # the oldest step still in flight: fetch it while the newer ones run
                t = tickets[k - depth + 1]
                o = fl.result(t)
                E, F, n = want[order[t]]
                assert torch.equal(o["energy"], E) and torch.equal(o["forces"], F) and int(o["n_edges"]) == n
        with pytest.raises(ValueError, match="reused"):
            fl.result(tickets[0])
        assert int(fl.edge_total) == sum(want[i][2] for i in order)
        assert all(st.captures == 1 for st in fl.steps) and not fl.overflowed()
        o = fl(*draws[1])                                     # the call form: submit + result
        assert torch.equal(o["forces"], want[1][1])
    # the loop over a stream of batches (run/inference.py:39-75): results in order, own copies
    got = list(runtime.evaluate_batches(model, (draws[i] for i in (3, 0, 4, 1, 2)), cap))
    assert len(got) == 5
    for o, i in zip(got, (3, 0, 4, 1, 2)):
        n, g = draws[i][0].shape[0], draws[i][2].numel() - 1
        assert o["energy"].shape == (g,) and o["forces"].shape == (n, 3)
        assert torch.equal(o["energy"], want[i][0]) and torch.equal(o["forces"], want[i][1]) and int(o["n_edges"]) == want[i][2]


@pytest.mark.parametrize("forces", [False, True])
def test_bench_train_mode_times_a_ddp_step_over_two_ranks_on_one_card(forces):
    """`bench.py --train [--forces] --gpus 2`: one optimisation step per timed step, the model in DistributedDataParallel, so the step
    carries the gradient all-reduce (gloo here, every rank on device 0; RCCL on a multi-GPU node).  One JSON line from rank 0."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(XEQ_BENCH_BACKEND="gloo", XEQ_BENCH_DEVICE="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--train", "--workload", "qm9_64", "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd + (["--forces"] if forces else []), capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["metric"].startswith("training step") and line["value"] > 0 and line["ms_per_step"] > 0
    assert "DistributedDataParallel x2" in line["config"]["parallelism"] and np.isfinite(line["config"]["loss"])

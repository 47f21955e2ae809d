"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the
committed golden fixtures.  Run on the MI355X box:  pytest tests -m gpu

Tolerances:
  fp64: energies/forces/features 1e-10 relative-ish (pure rounding differences)
  fp32: energies |dE| <= 1e-5 |E| + 1e-4; forces: THE tolerance is f32_force_bounds() below -- the HIP path may be no
        worse than the reference's own arithmetic run in fp32 on the same inputs (the oracle in fp32 against the oracle
        in fp64): max |dF| <= max(1e-4, 1.5 max |F_oracle32 - F_oracle64|), and the same at the 99th percentile
  integer outputs (edge_index, cell_offsets): bit-exact
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn

from tests import parity_record

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"
F32_FORCE_FLOOR = 1e-4      # the stated fp32 force tolerance (model units)
F32_ORACLE_FACTOR = 1.5     # ... widened only to 1.5 x the error of the reference's own arithmetic in fp32 on the same inputs (99th percentile)
F32_ORACLE_FACTOR_MAX = 1.5 # ... and at the single worst component (round 3 needed 2.5 here against an envelope of CPU evaluations only; see f32_force_bounds)


def f32_twin(oracle):
    """The same oracle with fp32 weights: the reference's op sequence (nn/xpainn.py:128-231, nn/basic.py:143-159) at the
    reference's default precision (utils/config.py:58)."""
    twin = getattr(oracle, "_f32_twin", None)
    if twin is None:
        sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in oracle.sd.items()}
        twin = oracle._f32_twin = orc.XPaiNNOracle(sd32, **getattr(oracle, "_kw", {}))
    return twin


F32_ORACLE_ORDERS = 4       # members of the fp32 oracle's ensemble (edge orders), see f32_force_bounds; 16 below 2 000 atoms


_F32_BOUNDS_CACHE = {}


class _aten_only:
    """Context of the envelope's GPU member: the differentiable form of the blocks (xequinet_amd/nn/training.py) with EVERY kernel
    switch of this package off, so that what runs is the reference's op sequence (index_select, nn.Linear, layer norms, index_add,
    autograd's own reverse pass) on the vendor's ATen / library kernels.  On exit it asserts that libxeq_hip.so launched NOTHING
    while the block ran (xeq_launch_count counts every launch of the library, whichever front -- ctypes or the registered torch
    operators -- made it): round 4's twin had silently become this package's own training kernels, i.e. the code under test helped
    define its own tolerance."""

    def __enter__(self):
        from xequinet_amd import lib
        from xequinet_amd.nn import training

        self._training = training
        self._saved = (training.NATIVE_MESSAGE, training.NATIVE_NODE, training.NATIVE_LINEAR)
        training.NATIVE_MESSAGE = training.NATIVE_NODE = training.NATIVE_LINEAR = False
        self._lib = lib
        self._count = lib.launch_count()
        return self

    def __exit__(self, *exc):
        t = self._training
        t.NATIVE_MESSAGE, t.NATIVE_NODE, t.NATIVE_LINEAR = self._saved
        launched = self._lib.launch_count() - self._count
        if exc[0] is None:
            assert launched == 0, f"the ATen-only member of the fp32 envelope launched {launched} kernel group(s) of libxeq_hip.so"
        return False


def _aten_gpu_twin(oracle):
    """The model whose training form -- evaluated inside ``_aten_only()`` -- is the reference's op sequence in fp32 on the GPU
    through ATen: what the reference itself computes when it runs on a GPU (its ``index_add`` is then an atomic scatter)."""
    twin = getattr(oracle, "_aten_gpu_twin", None)
    if twin is None:
        from xequinet_amd.nn import resolve_model

        twin = resolve_model("xpainn", **getattr(oracle, "_kw", {}))
        twin.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in oracle.sd.items()})
        twin = oracle._aten_gpu_twin = twin.to(DEV).train().requires_grad_(True)
    return twin


class ForceBounds(tuple):
    """(bound_max, bound_p99, err32_max, err32_p99) as before -- err32 = the envelope over both kinds of member -- plus the two
    envelopes apart: ``cpu`` = (max, p99) of the CPU oracle in fp32, ``aten_gpu`` = (max, p99) of the ATen-only GPU evaluation."""
    cpu = (0.0, 0.0)
    aten_gpu = (0.0, 0.0)

    def record(self):
        return dict(oracle32_max_abs_dF=self[2], oracle32_p99_abs_dF=self[3], bound_dF_max=self[0], bound_dF_p99=self[1],
                    cpu_oracle32_max_abs_dF=self.cpu[0], cpu_oracle32_p99_abs_dF=self.cpu[1],
                    aten_gpu32_max_abs_dF=self.aten_gpu[0], aten_gpu32_p99_abs_dF=self.aten_gpu[1],
                    envelope_members_cpu=getattr(self, "members", (None, None))[0], envelope_members_aten_gpu=getattr(self, "members", (None, None))[1])


def bounds_from_envelopes(err_cpu, err_gpu, members):
    """ForceBounds over per-atom envelopes gathered chunk by chunk (same factors and floor as f32_force_bounds)."""
    err = np.maximum(err_cpu, err_gpu)
    e_max, e_p99 = float(err.max()), float(np.quantile(err, 0.99))
    out = ForceBounds((max(F32_FORCE_FLOOR, F32_ORACLE_FACTOR_MAX * e_max), max(F32_FORCE_FLOOR, F32_ORACLE_FACTOR * e_p99), e_max, e_p99))
    out.cpu = (float(err_cpu.max()), float(np.quantile(err_cpu, 0.99)))
    out.aten_gpu = (float(err_gpu.max()), float(np.quantile(err_gpu, 0.99)))
    out.members = members
    out.err, out.err_cpu, out.err_gpu = err, err_cpu, err_gpu
    return out


def f32_force_bounds(oracle, ref_in, Fref, cpu_members=None):
    """(bound_max, bound_p99, err32_max, err32_p99): the reference's arithmetic in fp32 evaluated on the very inputs of the fp64 oracle.
    ``cpu_members`` overrides the number of CPU-oracle edge orders (the whole-batch comparisons of tests/test_gpu_fullsize.py take two
    per chunk: the CPU oracle is the suite's wall time); the per-atom envelopes stay on the result (``err`` / ``err_cpu`` / ``err_gpu``)
    so that a caller that walks a batch chunk by chunk can form ONE maximum and ONE 99th percentile over all of it.

    err32 = |F_32 - F_oracle64| is what fp32 rounding does to the reference's own arithmetic on these atoms.  This
    random-init model is ill-conditioned on a few molecules (on 384 QM9-shape molecules the fp32 oracle's error is 3e-7 at
    the median, 1e-6 at the 90th percentile, 2e-4 at the 99.9th and 8e-4 at the worst atom; the HIP path: 3e-7, 1e-6, 1.8e-4,
    4.5e-4 -- profiles/parity_r03.json), and which atom is worst moves with the summation order: the reference's fp32
    result is a SET (its index_add, nn/xpainn.py:156-159, is an atomic scatter on a GPU: SURVEY a13), so err32 is taken as
    the envelope over F32_ORACLE_ORDERS legitimate edge orders (as given, reversed, fixed permutations; four times
    as many for small systems (under 30 k edges), where one ill-conditioned atom is the whole tail: its error is that atom's conditioning times
    one draw of the rounding noise, and a single HIP draw exceeds 1.5 x the largest of four reference draws one time in
    fifty -- with sixteen, one time in five hundred), each order evaluated twice: by the CPU oracle in fp32 and by the same op sequence
    on the GPU through ATen only (``_aten_only``: no kernel of this package runs, asserted per evaluation).  The HIP path
    has to stay within 1.5 x that at the maximum and at the 99th percentile, or within BASELINE.md's 1e-4 where the fp32
    evaluations are better than that.  The two kinds of member are also recorded apart (ForceBounds.cpu / .aten_gpu)."""
    in32 = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in ref_in.items()}
    ei = in32["edge_index"]
    n_e = ei.shape[1]
    # the same reference inputs come back several times in a session (one check per kernel family, replay and eager forms): the
    # ensemble is a property of the inputs and the weights, evaluated once (the CPU oracle is what the GPU suite's wall time is)
    key = (hash(np.ascontiguousarray(Fref).tobytes()), n_e, tuple(Fref.shape), cpu_members)
    if key in _F32_BOUNDS_CACHE:
        return _F32_BOUNDS_CACHE[key]
    twin = f32_twin(oracle)
    gpu = _aten_gpu_twin(oracle)
    rng = np.random.default_rng(20261004)
    err_cpu = err_gpu = None
    n_members = F32_ORACLE_ORDERS * (4 if n_e < 30000 else 1)   # small systems: one ill-conditioned atom is the whole tail (see above)
    # large systems: the CPU oracle is what the suite's wall time is, so it keeps four orders there -- the GPU member costs a tenth of
    # a second and takes sixteen orders at every size (its maximum over four draws moved between 7.0e-4 and 1.2e-3 from run to run on
    # QM9-1024 -- its index_add is an atomic scatter -- against this package's 9.8e-4, which repeats bit for bit: a bound of 1.5 x the
    # largest of FOUR draws is one unlucky run away; profiles/parity_r05.json)
    if cpu_members is not None:
        n_members = int(cpu_members)
    n_gpu_members = max(n_members, 4 * F32_ORACLE_ORDERS)
    for member in range(n_gpu_members):
        perm = (torch.arange(n_e) if member == 0 else torch.arange(n_e - 1, -1, -1) if member == 1
                else torch.as_tensor(rng.permutation(n_e)))
        run = dict(in32)
        run["edge_index"] = ei[:, perm]
        if "cell_offsets" in run:
            run["cell_offsets"] = run["cell_offsets"][perm]
        if member < n_members:
            e = np.abs(twin(run, compute_forces=True)["forces"].double().numpy() - Fref)
            err_cpu = e if err_cpu is None else np.maximum(err_cpu, e)
        # ... and the same member through ATen's fp32 kernels on the GPU (autograd's reverse pass, atomic index_add): the reference's
        # arithmetic as a GPU runs it.  The tail is the conditioning of the random-weight network on a few molecules
        # (profiles/r04_fp32_tail.txt), where ANY fp32 evaluation order draws errors 10-50 x apart, so the envelope holds GPU members too
        dev_in = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in run.items()}
        with _aten_only(), torch.enable_grad():
            fg = gpu(dev_in, compute_forces=True)["forces"].detach().double().cpu().numpy()
        e = np.abs(fg - Fref)
        err_gpu = e if err_gpu is None else np.maximum(err_gpu, e)
    err = np.maximum(err_cpu, err_gpu)
    e_max, e_p99 = float(err.max()), float(np.quantile(err, 0.99))
    out = ForceBounds((max(F32_FORCE_FLOOR, F32_ORACLE_FACTOR_MAX * e_max), max(F32_FORCE_FLOOR, F32_ORACLE_FACTOR * e_p99), e_max, e_p99))
    out.cpu = (float(err_cpu.max()), float(np.quantile(err_cpu, 0.99)))
    out.aten_gpu = (float(err_gpu.max()), float(np.quantile(err_gpu, 0.99)))
    out.members = (n_members, n_gpu_members)
    out.err, out.err_cpu, out.err_gpu = err, err_cpu, err_gpu
    _F32_BOUNDS_CACHE[key] = out
    return out


def _load(name):
    return np.load(os.path.join(G, name))


def _t(a, dtype=None):
    t = torch.as_tensor(np.asarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


# ----------------------------------------------------------------------------- graph
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_radius_graph_nonpbc_bit_exact(dtype):
    from xequinet_amd.cluster import radius_graph

    pos, z, ptr = syn.synth_qm9_batch(64, seed=3)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    pos = pos.astype(npdt)
    want = orc.radius_graph_canonical(pos, ptr, 5.0)
    got = radius_graph(_t(pos), 5.0, ptr=_t(ptr)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    # batch-vector form (sorted batch), smaller cutoff => ragged degrees
    batch = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    want = orc.radius_graph_canonical(pos, ptr, 2.0)
    got = radius_graph(_t(pos), 2.0, batch=_t(batch)).cpu().numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_radius_graph_cell_list_equals_pair_sweep(dtype, monkeypatch):
    """Open-boundary cell list (graphs of many atoms) against the O(n_g^2) sweep: a 3000-atom blob, a flat sheet (one
    bin along z), a batch with small and empty graphs -- identical edge_index."""
    from xequinet_amd.cluster import radius_graph

    rng = np.random.default_rng(12)
    blob = rng.normal(0, 9.0, size=(3000, 3))
    sheet = np.concatenate([rng.uniform(0, 60, size=(1500, 2)), rng.uniform(0, 0.5, size=(1500, 1))], axis=1)
    small = rng.normal(0, 2.0, size=(9, 3))
    pos = np.concatenate([blob, small, sheet]).astype(np.float32 if dtype == torch.float32 else np.float64)
    ptr = np.array([0, 3000, 3009, 3009, 4509])
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("XEQ_CELL_LIST", flag)
        outs[flag] = radius_graph(_t(pos), 5.0, ptr=_t(ptr))
    assert outs["0"].shape[1] > 20000
    assert torch.equal(outs["0"], outs["1"])
    monkeypatch.delenv("XEQ_CELL_LIST")
    want = orc.radius_graph_canonical(pos[3000:3009], np.array([0, 9]), 5.0)
    sub = outs["1"][:, (outs["1"][0] >= 3000) & (outs["1"][0] < 3009)].cpu().numpy() - 3000
    np.testing.assert_array_equal(sub, want)


def test_radius_graph_edge_cases():
    from xequinet_amd.cluster import radius_graph

    # single atoms, an empty graph in the middle, two coincident-free atoms out of range
    pos = np.array([[0, 0, 0], [10, 0, 0], [10.5, 0, 0], [30, 0, 0], [36, 0, 0]], dtype=np.float32)
    ptr = np.array([0, 1, 1, 3, 5])
    want = orc.radius_graph_canonical(pos, ptr, 5.0)
    got = radius_graph(_t(pos), 5.0, ptr=_t(ptr)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    assert got.shape == (2, 2)
    # no atoms at all
    got = radius_graph(torch.zeros((0, 3), device=DEV), 5.0, ptr=_t(np.array([0])))
    assert got.shape == (2, 0)


@pytest.mark.parametrize("name", ["water192", "two_graphs_unwrapped", "triclinic40", "slab30"])
def test_radius_graph_pbc_golden_bit_exact(name):
    from xequinet_amd.data import radius_graph_pbc

    f = _load(f"radius_graph_pbc_{name}.npz")
    npg = f["n_per_graph"]
    pbc = torch.tensor(np.tile(f["pbc"], (len(npg), 1)))
    ei, co = radius_graph_pbc(_t(f["pos"]), _t(npg), pbc.to(DEV), _t(f["cell"]), float(f["cutoff"]))
    np.testing.assert_array_equal(ei.cpu().numpy(), f["edge_index"])
    np.testing.assert_array_equal(co.cpu().numpy(), f["cell_offsets"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_radius_graph_pbc_pruned_equals_exhaustive(dtype):
    """The image-pruned search (default) against the exhaustive sweep over all images on random triclinic cells,
    including cells smaller than the cutoff (2 images per axis), a slab (one open axis) and un-wrapped positions:
    identical edge_index, cell_offsets and order."""
    from xequinet_amd import ops
    from xequinet_amd.data.radius_graph import _image_counts, wrap_positions

    rng = np.random.default_rng(5)
    for trial, (n_atoms, L, pbc) in enumerate([((40, 25), 9.0, [True, True, True]), ((30,), 4.2, [True, True, True]),
                                                ((35, 20, 50), 7.5, [True, True, False]), ((64,), 12.0, [True, False, True]),
                                                ((900, 300), 27.0, [True, True, True]), ((700,), 31.0, [True, True, False])]):
        G = len(n_atoms)
        cell = np.stack([np.eye(3) * L * rng.uniform(0.8, 1.3) + rng.normal(0, 0.12 * L, size=(3, 3)) for _ in range(G)])
        frac = [rng.uniform(-0.4, 1.6, size=(n, 3)) for n in n_atoms]
        pos = np.concatenate([f @ c for f, c in zip(frac, cell)])
        pos_t, cell_t = _t(pos, dtype), _t(cell, dtype)
        nn = _t(np.array(n_atoms))
        reps, prune = _image_counts(cell_t, pbc, 5.0, with_prune=True)
        cells_per_dim = [torch.arange(-r, r + 1, device=DEV, dtype=dtype) for r in reps]
        cell_offsets = torch.cartesian_prod(*cells_per_dim).reshape(-1, 3)
        img = torch.bmm(cell_offsets.view(1, -1, 3).expand(G, -1, -1).contiguous(), cell_t)
        pw, shift = wrap_positions(pos_t, cell_t, nn, pbc)
        ptr = _t(np.concatenate([[0], np.cumsum(n_atoms)]))
        from xequinet_amd.data.radius_graph import _with_bins

        a = ops.radius_graph_pbc_raw(pw, ptr, img, cell_offsets, shift, 5.0)
        b = ops.radius_graph_pbc_raw(pw, ptr, img, cell_offsets, shift, 5.0, prune=prune)
        c = ops.radius_graph_pbc_raw(pw, ptr, img, cell_offsets, shift, 5.0, prune=_with_bins(prune, pbc))   # cell list
        assert a[0].shape[1] > 0, trial
        for u, v, w in zip(a, b, c):
            assert torch.equal(u, v), trial
            assert torch.equal(u, w), trial


@pytest.mark.parametrize("dtype,npdt", [(torch.float32, np.float32), (torch.float64, np.float64)])
def test_radius_graph_pbc_random_cells_equal_oracle(dtype, npdt):
    """Beyond the four reference-generated fixtures: random triclinic cells (some smaller than the cutoff: two images per
    axis), mixed periodic flags, several graphs per call, atoms outside the cell -- the public `radius_graph_pbc` against
    the oracle restatement (itself pinned bit for bit by the fixtures): identical edges, offsets and order."""
    from xequinet_amd.data import radius_graph_pbc

    rng = np.random.default_rng(11)
    cases = [((17, 9), 6.5, [True, True, True]), ((12,), 3.9, [True, True, True]), ((20, 5, 14), 7.0, [True, True, False]),
             ((25,), 8.0, [True, False, False]), ((16, 16), 5.5, [False, True, True]), ((30,), 11.0, [False, False, False])]
    for trial, (n_atoms, L, pbc) in enumerate(cases):
        G = len(n_atoms)
        cell = np.stack([np.eye(3) * L * rng.uniform(0.85, 1.25) + rng.normal(0, 0.1 * L, size=(3, 3)) for _ in range(G)]).astype(npdt)
        pos = np.concatenate([rng.uniform(-0.3, 1.4, size=(n, 3)) @ c for n, c in zip(n_atoms, cell)]).astype(npdt)
        want_ei, want_co = orc.radius_graph_pbc_oracle(pos, np.array(n_atoms), pbc, cell, 5.0)
        if not any(pbc):
            continue   # the reference routes fully open graphs to radius_graph (data/transform.py:40-64), not here
        ei, co = radius_graph_pbc(_t(pos), _t(np.array(n_atoms)), torch.tensor([pbc] * G, device=DEV), _t(cell), 5.0)
        assert ei.dtype == torch.int64 and co.dtype == dtype, trial
        np.testing.assert_array_equal(ei.cpu().numpy(), want_ei, err_msg=f"trial {trial}")
        np.testing.assert_array_equal(co.cpu().numpy(), want_co, err_msg=f"trial {trial}")
        assert want_ei.shape[1] > 0, trial


def test_single_radius_graph_golden():
    from xequinet_amd.data import single_radius_graph

    a = _load("radius_graph_pbc_water192.npz")
    b = _load("single_radius_graph_water192.npz")
    ei, co = single_radius_graph(_t(a["pos"]), torch.tensor([True, True, True]), _t(a["cell"][0]), 5.0)
    np.testing.assert_array_equal(ei.cpu().numpy(), b["edge_index"])
    np.testing.assert_array_equal(co.cpu().numpy(), b["cell_offsets"])


def test_edge_graph_csr_views():
    from xequinet_amd.ops import EdgeGraph

    rng = np.random.default_rng(0)
    N, E = 50, 400
    ei = rng.integers(0, N, size=(2, E))
    g = EdgeGraph(_t(ei), N)
    c_rowptr = g.c_rowptr.cpu().numpy()
    n_rowptr = g.n_rowptr.cpu().numpy()
    np.testing.assert_array_equal(c_rowptr, np.searchsorted(np.sort(ei[0]), np.arange(N + 1)))
    np.testing.assert_array_equal(n_rowptr, np.searchsorted(np.sort(ei[1]), np.arange(N + 1)))
    np.testing.assert_array_equal(g.c_perm.cpu().numpy(), np.argsort(ei[0], kind="stable"))
    np.testing.assert_array_equal(g.n_perm.cpu().numpy(), np.argsort(ei[1], kind="stable"))
    # exclusive scan helper
    from xequinet_amd import lib
    from xequinet_amd.lib import call, ptr, stream
    for n in (0, 1, 7, 1023, 1024, 1025, 5000, 8191, 8192, 8193, 40000):
        cnt = torch.randint(0, 7, (n,), dtype=torch.int32, device=DEV)
        out = torch.empty(n + 1, dtype=torch.int32, device=DEV)
        call("xeq_exclusive_scan_i32", ptr(cnt), n, ptr(out), stream())
        want = np.concatenate([[0], np.cumsum(cnt.cpu().numpy())])
        np.testing.assert_array_equal(out.cpu().numpy(), want)


# --------------------------------------------------------------------- edge geometry
@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 1e-6), ("f64", torch.float64, 1e-13)])
def test_edge_vectors_golden(tag, dtype, tol):
    from xequinet_amd.nn.basic import compute_edge_data

    f = _load(f"edge_data_aspirin_{tag}.npz")
    data = {"pos": _t(f["pos"], dtype), "edge_index": _t(f["edge_index"])}
    out = compute_edge_data(data, compute_forces=False)
    np.testing.assert_allclose(out["edge_vector"].cpu().numpy(), f["edge_vector"], rtol=0, atol=tol)
    np.testing.assert_allclose(out["edge_length"].cpu().numpy(), f["edge_length"], rtol=0, atol=tol)
    g = _load("radius_graph_pbc_two_graphs_unwrapped.npz")
    e = _load(f"edge_data_pbc2_{tag}.npz")
    npg = g["n_per_graph"]
    data = {
        "pos": _t(g["pos"], dtype), "edge_index": _t(g["edge_index"]), "cell": _t(g["cell"], dtype),
        "cell_offsets": _t(g["cell_offsets"], dtype), "batch": _t(np.repeat(np.arange(len(npg)), npg)),
        "ptr": _t(np.concatenate([[0], np.cumsum(npg)])),
    }
    out = compute_edge_data(data, compute_forces=False)
    np.testing.assert_allclose(out["edge_vector"].cpu().numpy(), e["edge_vector"], rtol=0, atol=10 * tol)
    np.testing.assert_allclose(out["edge_length"].cpu().numpy(), e["edge_length"], rtol=0, atol=10 * tol)


def test_edge_vectors_backward_matches_autograd():
    from xequinet_amd.nn.basic import compute_edge_data

    pos, z, ptr = syn.synth_qm9_batch(5, seed=1)
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, 4.0)
    perm = np.random.default_rng(0).permutation(ei.shape[1])  # unsorted edges: exercises both perms
    ei = ei[:, perm]
    w = np.random.default_rng(1).normal(size=(ei.shape[1], 3))
    wd = np.random.default_rng(2).normal(size=(ei.shape[1],))
    p_ref = torch.tensor(pos, requires_grad=True)
    d = orc.compute_edge_data({"pos": p_ref, "edge_index": torch.tensor(ei)}, compute_forces=False)
    (d["edge_vector"] * torch.tensor(w)).sum().add((d["edge_length"] * torch.tensor(wd)).sum()).backward()
    p = _t(pos, torch.float64).requires_grad_()
    out = compute_edge_data({"pos": p, "edge_index": _t(ei)}, compute_forces=False)
    ((out["edge_vector"] * _t(w)).sum() + (out["edge_length"] * _t(wd)).sum()).backward()
    np.testing.assert_allclose(p.grad.cpu().numpy(), p_ref.grad.numpy(), rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------ operator level
IRREPS = ["128x0e + 64x1o + 32x2e", "16x1o", "8x0e+4x1o+2x2e", "5x0e+3x2e"]


@pytest.mark.parametrize("irreps", IRREPS)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float64, 1e-13)])
def test_spherical_harmonics_fwd_bwd(irreps, dtype, tol):
    from xequinet_amd import o3

    torch.manual_seed(0)
    v = torch.randn(37, 3, dtype=torch.float64) * 2.0
    v[0] = torch.tensor([0.0, 0.0, 1.5])  # axis-aligned known answers ride along
    v[1] = torch.tensor([2.0, 0.0, 0.0])
    w = torch.randn(37, orc.irreps_dim(irreps), dtype=torch.float64)
    vr = v.clone().requires_grad_()
    ref = orc.spherical_harmonics(irreps, vr)
    (ref * w).sum().backward()
    sh = o3.SphericalHarmonics(irreps, normalize=True, normalization="component")
    vg = v.to(dtype).to(DEV).requires_grad_()
    out = sh(vg)
    (out * w.to(dtype).to(DEV)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().double().numpy(), ref.detach().numpy(), rtol=0, atol=tol * 5)
    np.testing.assert_allclose(vg.grad.cpu().double().numpy(), vr.grad.numpy(), rtol=0, atol=tol * 200)


@pytest.mark.parametrize("irreps", IRREPS)
def test_elementwise_tp_dot_invariant(irreps):
    from xequinet_amd import o3
    from xequinet_amd.nn.o3layer import EquivariantDot, Invariant

    torch.manual_seed(1)
    n, D, C = 23, orc.irreps_dim(irreps), orc.irreps_num(irreps)
    x, y, g = torch.randn(n, D, dtype=torch.float64), torch.randn(n, D, dtype=torch.float64), torch.randn(n, C, dtype=torch.float64)
    xr, yr, gr = (t.clone().requires_grad_() for t in (x, y, g))
    ref = orc.elementwise_tp(irreps, xr, gr).square().sum() + (orc.equivariant_dot(irreps, xr, yr) * gr).sum() \
        + orc.invariant(irreps, yr).sum()
    ref.backward()
    xg, yg, gg = (t.to(DEV).requires_grad_() for t in (x, y, g))
    etp = o3.ElementwiseTensorProduct(irreps, f"{C}x0e")
    out = etp(xg, gg).square().sum() + (EquivariantDot(irreps)(xg, yg) * gg).sum() + Invariant(irreps)(yg).sum()
    out.backward()
    assert abs(out.item() - ref.item()) < 1e-9 * abs(ref.item())
    for a, b in ((xg, xr), (yg, yr), (gg, gr)):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("irreps", ["128x0e + 64x1o + 32x2e", "8x0e+4x1o+2x2e", "16x1o"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.float64, 1e-11)])
def test_equivariant_layer_norm_fwd_bwd(irreps, dtype, tol):
    from xequinet_amd.nn.o3layer import EquivariantLayerNorm

    torch.manual_seed(2)
    n, D, C = 19, orc.irreps_dim(irreps), orc.irreps_num(irreps)
    m0 = orc.parse_irreps(irreps)[0][0] if orc.parse_irreps(irreps)[0][1] == 0 else 0
    x = torch.randn(n, D, dtype=torch.float64) * 3 + 0.5
    x[0] = 0.0  # first-layer case: zeros in, zeros (+bias) out, finite gradient
    w, b = torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(m0, dtype=torch.float64)
    gout = torch.randn(n, D, dtype=torch.float64)
    xr = x.clone().requires_grad_()
    ref = orc.equivariant_layer_norm(irreps, xr, w, b)
    (ref * gout).sum().backward()
    ln = EquivariantLayerNorm(irreps).to(dtype).to(DEV).requires_grad_(False)
    ln.affine_weight.copy_(w.to(dtype))
    ln.affine_bias.copy_(b.to(dtype))
    xg = x.to(dtype).to(DEV).requires_grad_()
    out = ln(xg)
    (out * gout.to(dtype).to(DEV)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().double().numpy(), ref.detach().numpy(), rtol=tol, atol=tol)
    scale = xr.grad.abs().max().item()
    np.testing.assert_allclose(xg.grad.cpu().double().numpy(), xr.grad.numpy(), rtol=tol, atol=tol * scale)


def test_exponential_radial_bases_against_the_reference():
    """nn/rbf.py:161-207 through xeq_radial_fwd: ExponentialBernstein as resolve_rbf builds it (the cutoff handed over as alpha) and with
    the class default, ExponentialNorm -- against the reference's own values (tests/golden/rbf_exp_*.npz); the modules carry the
    reference's parameter and buffer names."""
    from xequinet_amd.nn import rbf as prbf

    for tag, dtype, tol in (("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-11)):
        f = _load(f"rbf_exp_{tag}.npz")
        d = _t(f["dist"], dtype)
        torch.set_default_dtype(dtype)
        try:
            mods = {"expbern20_a5": prbf.resolve_rbf("expbern", 20, 5.0), "expbern20_a05": prbf.ExponentialBernstein(20),
                    "expnorm20_rc5": prbf.resolve_rbf("expnorm", 20, 5.0)}
            assert set(mods["expbern20_a5"].state_dict()) == {"_alpha", "logc", "n", "v"} and set(mods["expnorm20_rc5"].state_dict()) == {"beta", "mu"}
            for name, m in mods.items():
                for pname, p in m.named_parameters():
                    np.testing.assert_allclose(p.detach().numpy(), f[f"{name}_param_{pname}"], rtol=1e-6 if dtype == torch.float32 else 1e-14)
                np.testing.assert_allclose(m.to(DEV)(d).cpu().numpy(), f[name], rtol=tol, atol=tol * 1e-3, err_msg=name)
        finally:
            torch.set_default_dtype(torch.float32)


@pytest.mark.parametrize("kernel", ["expbern", "expnorm"])
@pytest.mark.parametrize("dtype,n_mol", [(torch.float64, 7), (torch.float32, 7), (torch.float32, 40)])
def test_model_with_exponential_radial_bases(kernel, dtype, n_mol):
    """The default XPaiNN with rbf_kernel = "expbern" / "expnorm" (nn/rbf.py:14-17): energies and forces against the fp64 oracle -- the
    sb kernels (7 molecules, both precisions), the matrix-core wq kernels (40 molecules, f32); the basis and its distance derivative are
    evaluated inside the record kernels (csrc/xeq_common.h::radial)."""
    model, oracle = _build(dtype, rbf_kernel=kernel)
    pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=5)
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, 5.0)
    _check_model(model, oracle, pos, z, ptr, ei, dtype)


def test_radial_and_scatter_ops():
    from xequinet_amd.nn import rbf as prbf
    from xequinet_amd.scatter import scatter, scatter_sum

    for tag, dtype, tol in (("f32", torch.float32, 3e-6), ("f64", torch.float64, 1e-12)):
        f = _load(f"rbf_{tag}.npz")
        d = _t(f["dist"], dtype)
        torch.set_default_dtype(dtype)
        try:
            np.testing.assert_allclose(prbf.SphericalBesselj0(20, 5.0).to(DEV)(d).cpu().numpy(), f["bessel20_rc5"], rtol=tol, atol=tol)
            np.testing.assert_allclose(prbf.SphericalBesselj0(8, 4.0).to(DEV)(d).cpu().numpy(), f["bessel8_rc4"], rtol=tol, atol=tol)
            np.testing.assert_allclose(prbf.CosineCutoff(5.0)(d).cpu().numpy(), f["cosine_rc5"], rtol=tol, atol=tol)
            np.testing.assert_allclose(prbf.PolynomialCutoff(5.0)(d).cpu().numpy(), f["poly3_rc5"], rtol=tol, atol=tol)
            np.testing.assert_allclose(prbf.GaussianSmearing(20, 5.0).to(DEV)(d).cpu().numpy(), f["gauss20_rc5"], rtol=tol, atol=tol)
        finally:
            torch.set_default_dtype(torch.float32)
    # segmented / atomic scatter-sum
    rng = np.random.default_rng(0)
    counts = rng.integers(0, 40, size=30)
    counts[3] = 0
    ptr = np.concatenate([[0], np.cumsum(counts)])
    batch = np.repeat(np.arange(30), counts)
    for width in (None, 1, 7, 130):
        shape = (ptr[-1],) if width is None else (ptr[-1], width)
        src = rng.normal(size=shape)
        want = np.zeros((30,) + shape[1:])
        np.add.at(want, batch, src)
        s = _t(src).requires_grad_()
        got = scatter_sum(s, _t(batch), ptr=_t(ptr))
        np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=1e-12, atol=1e-12)
        gw = rng.normal(size=want.shape)
        (got * _t(gw)).sum().backward()
        np.testing.assert_allclose(s.grad.cpu().numpy(), gw[batch], rtol=0, atol=0)
        shuffled = rng.permutation(batch)                                  # arbitrary (unsorted) index: float atomics
        want2 = np.zeros_like(want)
        np.add.at(want2, shuffled, src)
        got2 = scatter(_t(src), _t(shuffled), dim_size=30)
        assert got2.shape == want.shape
        np.testing.assert_allclose(got2.cpu().numpy(), want2, rtol=1e-12, atol=1e-12)   # f64 atomics: order-dependent in the last bits only


# ------------------------------------------------------------------- fused message
def _message_case(irreps, node_dim, B, rbf_kind, cutoff_kind, dtype, shuffle, seed=0, with_ptr=False, n_mol=6, lone_atoms=0):
    from xequinet_amd import ops

    rng = np.random.default_rng(seed)
    pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=seed + 10)
    if lone_atoms:   # atoms without any neighbour, as graphs of their own, in runs between the molecules and behind the last one
        where = np.sort(rng.integers(0, n_mol + 1, size=lone_atoms))
        chunks, cuts, far = [], [0], 0
        for g in range(n_mol + 1):
            for _ in range(int((where == g).sum())):
                far += 1
                chunks.append(np.array([[1000.0 * far, -500.0, 250.0]]))
                cuts.append(cuts[-1] + 1)
            if g < n_mol:
                chunks.append(pos[ptr[g]:ptr[g + 1]])
                cuts.append(cuts[-1] + int(ptr[g + 1] - ptr[g]))
        pos, ptr = np.concatenate(chunks), np.array(cuts, dtype=np.int64)
    rc = 4.0
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, rc)
    if shuffle:
        ei = ei[:, rng.permutation(ei.shape[1])]
    N, E = len(pos), ei.shape[1]
    C, D = orc.irreps_num(irreps), orc.irreps_dim(irreps)
    H = node_dim + 2 * C
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)
    h, xhat = t64(rng.normal(size=(N, H))), t64(rng.normal(size=(N, D)))
    s, x = t64(rng.normal(size=(N, node_dim))), t64(rng.normal(size=(N, D)))
    W, b = t64(rng.normal(size=(H, B)) / math.sqrt(B)), t64(rng.normal(size=(H,)))
    vec = t64(pos[ei[0]] - pos[ei[1]])
    gs, gx = t64(rng.normal(size=(N, node_dim))), t64(rng.normal(size=(N, D)))
    if rbf_kind == "bessel":
        p0, p1 = t64(math.pi * np.arange(1, B + 1) / rc * (1 + 0.01 * rng.normal(size=B))).view(1, -1), None
    else:
        p0, p1 = t64(np.linspace(0, rc, B)).view(1, -1), t64(0.5 + rng.uniform(size=B)).view(1, -1)
    # ---- oracle (autograd)
    hr, xr, vr, sr, xir = (t.clone().requires_grad_() for t in (h, xhat, vec, s, x))
    for t in (W, b, p0, p1):     # parameter gradients of a training pass (ops.message_param_grad)
        if t is not None:
            t.requires_grad_()
    dist = torch.linalg.norm(vr, dim=-1, keepdim=True)
    rbf = orc.bessel_rbf(dist, p0, rc) if rbf_kind == "bessel" else orc.gaussian_rbf(dist, p0, p1)
    fcut = orc.cosine_cutoff(dist, rc) if cutoff_kind == "cosine" else orc.polynomial_cutoff(dist, rc)
    rsh = orc.spherical_harmonics(irreps, vr[:, [1, 2, 0]])
    filt = torch.nn.functional.linear(rbf, W, b) * fcut
    fo = hr.index_select(0, torch.tensor(ei[1])) * filt
    g_state, g_edge, m_s = torch.split(fo, [C, C, node_dim], dim=-1)
    m_x = orc.elementwise_tp(irreps, xr.index_select(0, torch.tensor(ei[1])), g_state) + orc.elementwise_tp(irreps, rsh, g_edge)
    s_ref = sr.index_add(0, torch.tensor(ei[0]), m_s)
    x_ref = xir.index_add(0, torch.tensor(ei[0]), m_x)
    ((s_ref * gs).sum() + (x_ref * gx).sum()).backward()
    # ---- HIP
    dev = lambda t: None if t is None else t.detach().to(dtype).to(DEV)
    hg, xg, vg, sg, xig = (dev(t).requires_grad_() for t in (h, xhat, vec, s, x))
    Wg, bg, p0g = (dev(t).requires_grad_() for t in (W, b, p0))
    p1g = None if p1 is None else dev(p1).requires_grad_()
    graph = ops.EdgeGraph(_t(ei), N, ptr=_t(ptr) if with_ptr else None)  # graph boundaries (the matrix-core kernels walk segments)
    mul = [0, 0, 0]
    for m_, l_, _ in orc.parse_irreps(irreps):
        mul[l_] = m_
    cfg = (rbf_kind, cutoff_kind, B, rc, node_dim, tuple(mul))
    s_out, x_out = ops.FusedMessage.apply(hg, xg, vg, sg, xig, Wg, bg, p0g, p1g, graph, cfg)
    ((s_out * dev(gs)).sum() + (x_out * dev(gx)).sum()).backward()
    grad = lambda t: None if t is None else t.grad
    return ((s_out, x_out, hg.grad, xg.grad, vg.grad, sg.grad, xig.grad, Wg.grad, bg.grad, p0g.grad, grad(p1g)),
            (s_ref, x_ref, hr.grad, xr.grad, vr.grad, sr.grad, xir.grad, W.grad, b.grad, p0.grad, grad(p1)))


MSG_CASES = [
    ("128x0e + 64x1o + 32x2e", 128, 20, "bessel", "cosine", False),
    ("128x0e + 64x1o + 32x2e", 128, 20, "bessel", "cosine", True),
    ("16x1o", 16, 8, "bessel", "cosine", False),
    ("8x0e+4x1o+2x2e", 12, 12, "bessel", "polynomial", True),
    ("5x0e+3x2e", 7, 32, "gaussian", "cosine", False),
    ("40x0e + 30x1o + 20x2e", 200, 17, "bessel", "cosine", False),
]


@pytest.mark.parametrize("irreps,node_dim,B,rbf_kind,cutoff_kind,shuffle", MSG_CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
def test_fused_message_fwd_bwd(irreps, node_dim, B, rbf_kind, cutoff_kind, shuffle, dtype, tol):
    got, want = _message_case(irreps, node_dim, B, rbf_kind, cutoff_kind, dtype, shuffle)
    names = ["s_out", "x_out", "grad_h", "grad_xhat", "grad_vec", "grad_s", "grad_x"]
    for name, a, b in zip(names, got, want):
        a = a.detach().cpu().double().numpy()
        b = b.detach().numpy()
        scale = max(1.0, np.abs(b).max())
        np.testing.assert_allclose(a, b, rtol=tol, atol=tol * scale, err_msg=name)


@pytest.mark.parametrize("irreps,node_dim,B,rbf_kind,cutoff_kind,shuffle", MSG_CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 3e-5)])
def test_fused_message_parameter_gradients(irreps, node_dim, B, rbf_kind, cutoff_kind, shuffle, dtype, tol):
    """Training pass: dL/dW_rbf, dL/db_rbf and the gradients of the basis parameters (freq, or mean and std) of the fused message
    against the oracle's autograd -- the node-walk form (f64, any shape) and the matrix-core form (f32, multiples of 32)."""
    got, want = _message_case(irreps, node_dim, B, rbf_kind, cutoff_kind, dtype, shuffle, n_mol=6 if dtype == torch.float64 else 30)
    for name, a, b in zip(["grad_W", "grad_b", "grad_p0", "grad_p1"], got[7:], want[7:]):
        assert (a is None) == (b is None), name
        if a is None:
            continue
        a, b = a.detach().cpu().double().numpy(), b.detach().numpy()
        assert a.shape == b.shape, name
        np.testing.assert_allclose(a, b, rtol=tol, atol=tol * max(1.0, np.abs(b).max()), err_msg=name)


@pytest.mark.parametrize("impl", ["wq", "auto"])
def test_fused_message_with_many_atoms_that_have_no_neighbour(impl, monkeypatch):
    """300 lone atoms scattered through (and behind) 40 molecules: in the wq walk plan each owns one quad of padding slots and takes
    the ordinary path (residual rows kept, zero gradients); every output and gradient, parameter gradients included, against the
    oracle, and the lone atoms' rows bit for bit."""
    monkeypatch.setenv("XEQ_MESSAGE_IMPL", impl)
    got, want = _message_case("128x0e + 64x1o + 32x2e", 128, 20, "bessel", "cosine", torch.float32, False, n_mol=40, lone_atoms=300)
    names = ["s_out", "x_out", "grad_h", "grad_xhat", "grad_vec", "grad_s", "grad_x", "grad_W", "grad_b", "grad_p0"]
    for name, a, b in zip(names, got, want):
        a, b = a.detach().cpu().double().numpy(), b.detach().numpy()
        np.testing.assert_allclose(a, b, rtol=3e-5, atol=3e-5 * max(1.0, np.abs(b).max()), err_msg=name)
    lone = np.abs(want[2].numpy()).sum(1) == 0       # nobody's neighbour: the oracle's dL/dh row is exactly zero
    assert lone.sum() >= 300
    assert not got[2].detach().cpu().numpy()[lone].any() and not got[3].detach().cpu().numpy()[lone].any()


WM_CASES = [
    ("128x0e + 64x1o + 32x2e", 128, 20, "bessel", "cosine", False),
    ("128x0e + 64x1o + 32x2e", 128, 20, "bessel", "cosine", True),
    ("64x0e + 32x1o + 32x2e", 64, 8, "gaussian", "polynomial", True),
    ("32x0e + 64x2e", 32, 17, "bessel", "cosine", False),
    ("96x0e + 32x1o", 96, 31, "bessel", "polynomial", False),
]


@pytest.mark.parametrize("impl", ["wq"])
@pytest.mark.parametrize("eps", ["16", "128", "100000"])
@pytest.mark.parametrize("irreps,node_dim,B,rbf_kind,cutoff_kind,shuffle", WM_CASES)
def test_fused_message_matrix_core_fwd_bwd(irreps, node_dim, B, rbf_kind, cutoff_kind, shuffle, eps, impl, monkeypatch):
    """The matrix-core kernels (xeq_message_{fwd,bwd}_wq, f32) against the fp64 oracle: every output of the
    forward pass and every gradient of the reverse pass, ragged molecules (3..29 atoms), shuffled edges, K = B + 1
    from 9 to 32, and stream lengths from one tile per stream to one wave for the whole batch."""
    monkeypatch.setenv("XEQ_MESSAGE_IMPL", impl)
    monkeypatch.setenv("XEQ_WQ_EDGES_PER_STREAM", eps)
    got, want = _message_case(irreps, node_dim, B, rbf_kind, cutoff_kind, torch.float32, shuffle, n_mol=40)
    names = ["s_out", "x_out", "grad_h", "grad_xhat", "grad_vec", "grad_s", "grad_x"]
    for name, a, b in zip(names, got, want):
        a = a.detach().cpu().double().numpy()
        b = b.detach().numpy()
        scale = max(1.0, np.abs(b).max())
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-5 * scale, err_msg=name)


@pytest.mark.parametrize("impl", ["wq"])
def test_fused_message_matrix_core_isolated_nodes_and_empty_graph(impl, monkeypatch):
    """Nodes without edges keep their residual rows (forward) and get zero gradients (reverse); a batch without any
    edge at all runs through the same entry points."""
    from xequinet_amd import ops

    monkeypatch.setenv("XEQ_MESSAGE_IMPL", impl)
    torch.manual_seed(3)
    mul, F, B, rc = (32, 32, 32), 32, 8, 3.0
    C, D, H = 96, 32 + 96 + 160, 32 + 192
    for N, ei in ((7, [[1, 2, 4, 1], [2, 1, 1, 4]]), (5, [[], []])):
        ei = torch.tensor(ei, dtype=torch.int64, device=DEV).reshape(2, -1)
        E = ei.shape[1]
        pos = torch.randn(N, 3, device=DEV)
        vec = (pos[ei[0]] - pos[ei[1]]).contiguous().requires_grad_()
        h, xhat = torch.randn(N, H, device=DEV, requires_grad=True), torch.randn(N, D, device=DEV, requires_grad=True)
        s, x = torch.randn(N, F, device=DEV), torch.randn(N, D, device=DEV)
        W, b = torch.randn(H, B, device=DEV), torch.randn(H, device=DEV)
        p0 = (math.pi * torch.arange(1, B + 1, device=DEV) / rc).float()
        cfg = ("bessel", "cosine", B, rc, F, mul)
        graph = ops.EdgeGraph(ei, N)
        so, xo = ops.FusedMessage.apply(h, xhat, vec, s, x, W, b, p0, None, graph, cfg)
        (so.sum() + xo.sum()).backward()
        touched = torch.zeros(N, dtype=torch.bool, device=DEV)
        touched[ei[0]] = True
        assert torch.equal(so[~touched], s[~touched]) and torch.equal(xo[~touched], x[~touched])
        if E:
            assert not torch.equal(so[touched], s[touched])
        nb = torch.zeros(N, dtype=torch.bool, device=DEV)
        nb[ei[1]] = True
        assert torch.all(h.grad[~nb] == 0) and torch.all(xhat.grad[~nb] == 0)
        assert torch.isfinite(h.grad).all() and torch.isfinite(xhat.grad).all() and vec.grad.shape == (E, 3)


@pytest.mark.parametrize("impl", ["wq"])
def test_fused_message_matrix_core_matches_scalar_broadcast_and_is_reproducible(impl, monkeypatch):
    """wq against the sb kernels on the same f32 inputs (the same terms; the k-order of the filter sum differs and its first
    sixteen k are split-bf16 products), and bitwise reproducibility (register sums in walk order)."""
    args = ("128x0e + 64x1o + 32x2e", 128, 20, "bessel", "cosine", torch.float32, False)
    monkeypatch.setenv("XEQ_MESSAGE_IMPL", "sb")
    ref, _ = _message_case(*args, n_mol=24)
    monkeypatch.setenv("XEQ_MESSAGE_IMPL", impl)
    a, _ = _message_case(*args, n_mol=24)
    b, _ = _message_case(*args, n_mol=24)
    for u, v in zip(a, b):
        assert (u is None and v is None) or torch.equal(u, v)
    for u, r in zip(a[:7], ref[:7]):
        scale = max(1.0, r.abs().max().item())
        assert (u - r).abs().max().item() <= 2e-5 * scale


def test_fused_message_matrix_core_falls_back_or_refuses(monkeypatch):
    """Channel layouts outside the matrix-core form (not in multiples of 32) and f64 run on the sb kernels under the default
    selection, and raise when wq is demanded."""
    monkeypatch.delenv("XEQ_MESSAGE_IMPL", raising=False)
    got, want = _message_case("8x0e+4x1o+2x2e", 12, 12, "bessel", "polynomial", torch.float32, True)
    np.testing.assert_allclose(got[0].detach().cpu().double().numpy(), want[0].detach().numpy(), rtol=2e-5, atol=2e-5)
    monkeypatch.setenv("XEQ_MESSAGE_IMPL", "wq")
    with pytest.raises(RuntimeError, match="does not fit"):
        _message_case("8x0e+4x1o+2x2e", 12, 12, "bessel", "polynomial", torch.float32, True)
    with pytest.raises(RuntimeError, match="does not fit"):
        _message_case("128x0e + 64x1o + 32x2e", 128, 20, "bessel", "cosine", torch.float64, False)


def test_fused_message_is_bitwise_reproducible():
    a, _ = _message_case("128x0e + 64x1o + 32x2e", 128, 20, "bessel", "cosine", torch.float32, False)
    b, _ = _message_case("128x0e + 64x1o + 32x2e", 128, 20, "bessel", "cosine", torch.float32, False)
    for u, v in zip(a, b):
        assert (u is None and v is None) or torch.equal(u, v)


def test_edge_graph_reverse_edge_map_equals_stable_sort(monkeypatch):
    """EdgeGraph(symmetric=True) (what NeighborTransform builds for open boundaries): the neighbour-sorted view from the
    reverse-edge map is the permutation of the stable sort by neighbour, for the pair sweep and the cell-list builder,
    with atoms that have no neighbour; a list that is not symmetric is reported with -1 entries."""
    from xequinet_amd import ops
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.lib import call, ptr, stream

    pos, z, ptr_np = syn.synth_qm9_batch(40, seed=4)
    pos = pos.copy()
    pos[5] += 100.0                       # an isolated atom
    big = np.random.default_rng(3).uniform(0, 30, size=(1500, 3))
    for p, pp, env in ((pos, ptr_np, "0"), (big, np.array([0, 1500]), "1")):
        monkeypatch.setenv("XEQ_CELL_LIST", env)
        b = NeighborTransform(5.0)(XequiBatch(_t(p, torch.float32), _t(np.ones(len(p), dtype=np.int32)), _t(pp)))
        g = getattr(b, "_xeq_edge_graph")
        ref = ops.EdgeGraph(b.edge_index, len(p))
        assert g.c_perm is None and ref.c_perm is None
        assert torch.equal(g.c_rowptr, ref.c_rowptr) and torch.equal(g.n_rowptr, ref.n_rowptr)
        assert torch.equal(g.n_perm, ref.n_perm) and g.n_edges > 0
    ei = _t(np.array([[0, 0, 1, 2], [1, 2, 0, 1]]))          # (0,2) has no reverse, (2,1) neither
    rp = ops.csr_rowptr(ei[0].contiguous(), 3)
    rev = torch.empty(4, dtype=torch.int32, device=DEV)
    call("xeq_reverse_edge_map", ptr(ei), 4, 3, ptr(rp), ptr(rev), stream())
    assert rev.tolist() == [2, -1, 0, -1]


@pytest.mark.parametrize("name", ["radius_graph_pbc_slab30.npz", "radius_graph_pbc_triclinic40.npz", "radius_graph_pbc_two_graphs_unwrapped.npz",
                                  "radius_graph_pbc_water192.npz"])
def test_periodic_mirror_map_on_the_reference_lists(name):
    """xeq_reverse_edge_map_pbc on the REFERENCE's own periodic lists (tests/golden: slab, triclinic cell with several images per pair,
    two graphs with unwrapped positions, a water box): every edge (i, j, o) finds its mirror (j, i, -o), the map is an involution --
    and, as the open-boundary map, it is then the permutation of the stable sort by neighbor up to the order inside a (neighbor, center)
    run; with one edge taken out, its mirror reports -1 and nothing else changes."""
    from xequinet_amd import ops
    from xequinet_amd.lib import call, dtype_code, ptr, stream

    f = _load(name)
    ei, off = _t(f["edge_index"]), _t(f["cell_offsets"], torch.float32)
    N, E = len(f["pos"]), ei.shape[1]
    assert bool((ei[0][1:] >= ei[0][:-1]).all())
    g = ops.EdgeGraph(ei, N, center_sorted=True, symmetric=True, cell_offsets=off)
    rev = g.mirror_map.long()
    assert g.mirror_walk and int(rev.min()) >= 0
    assert torch.equal(rev[rev], torch.arange(E, device=DEV))
    assert torch.equal(ei[0][rev], ei[1]) and torch.equal(ei[1][rev], ei[0]) and torch.equal(off[rev], -off)
    # the sorted view proper is built on demand and holds the same edges per neighbor
    assert g._n_view is None
    n_rowptr, n_perm = g.n_rowptr, g.n_perm
    assert torch.equal(n_rowptr, g.c_rowptr)                                          # symmetric: as many edges towards a node as from it
    assert torch.equal(torch.sort(n_perm.long())[0], torch.arange(E, device=DEV))
    # one edge out: the list is center-sorted still, the mate of the missing edge has no mirror, every other entry follows the shift
    k = E // 3
    keep = torch.ones(E, dtype=torch.bool, device=DEV)
    keep[k] = False
    ei2, off2 = ei[:, keep].contiguous(), off[keep].contiguous()
    g2 = ops.EdgeGraph(ei2, N, center_sorted=True, symmetric=True, cell_offsets=off2)
    rev2 = g2.mirror_map.long()
    mate = int(rev[k]) - (1 if int(rev[k]) > k else 0)
    assert int(rev2[mate]) == -1 and int((rev2 < 0).sum()) == 1
    ok = rev2 >= 0
    assert torch.equal(ei2[0][rev2[ok]], ei2[1][ok]) and torch.equal(off2[rev2[ok]], -off2[ok])


@pytest.mark.parametrize("drop", [False, True])
def test_periodic_model_with_the_mirror_map_equals_the_sorted_reverse_walk(drop, monkeypatch):
    """A periodic force evaluation whose reverse pass walks the forward plan through the mirror map (the default for this package's own
    periodic lists) against the same evaluation with the stable sort by neighbor, a second walk plan and a second set of records
    (XEQ_PBC_MIRROR=0): energies bit for bit (the forward pass is the same), forces and virial to rounding (other summation orders).
    ``drop``: with one edge taken out of the list the mirror walk misses one edge's reverse contribution by construction (a periodic list
    is symmetric up to a rounding at the cutoff, where that contribution vanishes; here the dropped edge is an ordinary one, so only
    finiteness and the size of the change are checked) -- nothing is read or written out of bounds."""
    from xequinet_amd import keys, ops

    f = _load("radius_graph_pbc_water192.npz")
    _, z, ptr, _ = syn.synth_water_box(4, seed=5)
    ei, off = _t(f["edge_index"]), _t(f["cell_offsets"], torch.float32)
    if drop:
        keep = torch.ones(ei.shape[1], dtype=torch.bool, device=DEV)
        keep[1234] = False
        ei, off = ei[:, keep].contiguous(), off[keep].contiguous()
    model, _ = _build(torch.float32)
    base = {"pos": _t(f["pos"], torch.float32), "atomic_numbers": _t(z.astype(np.int32)), "edge_index": ei, "ptr": _t(ptr),
            "batch": _t(np.zeros(len(z), dtype=np.int64)), "cell": _t(f["cell"], torch.float32), "cell_offsets": off}
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("XEQ_PBC_MIRROR", flag)
        d = dict(base)
        d[keys.EDGE_GRAPH] = g = ops.EdgeGraph(ei, len(z), center_sorted=True, ptr=base["ptr"], symmetric=True, cell_offsets=off)
        assert g.mirror_walk == (flag == "1") and (g.mirror_map is not None) == (flag == "1")
        with torch.enable_grad():
            outs.append(model(d, compute_forces=True, compute_virial=True))
    a, b = outs
    assert torch.equal(a["energy"], b["energy"])
    assert bool(torch.isfinite(a["forces"]).all()) and bool(torch.isfinite(a["virial"]).all())
    scale = float(b["forces"].abs().max())
    if not drop:
        assert float((a["forces"] - b["forces"]).abs().max()) <= 2e-6 * scale
        assert float((a["virial"] - b["virial"]).abs().max()) <= 2e-6 * float(b["virial"].abs().max())
    else:
        # one ORDINARY edge's reverse contribution is missing (three blocks reach across the whole 192-atom box): a small, bounded
        # change -- one edge of ~54 per atom -- not a fault
        diff = (a["forces"] - b["forces"]).abs().amax(1)
        assert float(diff.median()) <= 1e-4 * scale and float(diff.max()) <= 2e-2 * scale


def test_two_periodic_graphs_through_the_list_builder_take_the_mirror_walk(monkeypatch):
    """A batch of TWO periodic boxes with unwrapped positions (the reference's own fixture) through NeighborTransform -- the list this
    package builds, with the mirror map over both graphs -- and the wq kernels: energies, forces and virial equal the evaluation of the
    same list without the promise (stable sort by neighbor, reverse plan) to rounding, and the reference's list is reproduced."""
    from xequinet_amd import keys
    from xequinet_amd.data import NeighborTransform, XequiBatch

    monkeypatch.setenv("XEQ_MESSAGE_IMPL", "wq")
    f = _load("radius_graph_pbc_two_graphs_unwrapped.npz")
    n = f["n_per_graph"].astype(np.int64)
    ptr = np.concatenate([[0], np.cumsum(n)])
    z = np.random.default_rng(2).choice([1, 6, 7, 8], size=int(n.sum())).astype(np.int32)
    model, _ = _build(torch.float32)
    b = NeighborTransform(float(f["cutoff"]))(XequiBatch(_t(f["pos"], torch.float32), _t(z), _t(ptr), pbc=_t(f["pbc"].reshape(1, 3).repeat(2, 0)),
                                                        cell=_t(f["cell"], torch.float32)))
    assert np.array_equal(b.edge_index.cpu().numpy(), f["edge_index"]) and np.array_equal(b.cell_offsets.cpu().numpy(), f["cell_offsets"].astype(np.float32))
    data = b.to_dict()
    g = data[keys.EDGE_GRAPH]
    assert g.mirror_walk and int(g.mirror_map.min()) >= 0
    plain = {k: v for k, v in data.items() if k != keys.EDGE_GRAPH}
    with torch.enable_grad():
        a = model(dict(data), compute_forces=True, compute_virial=True)
        c = model(plain, compute_forces=True, compute_virial=True)
    assert torch.equal(a["energy"], c["energy"])
    assert float((a["forces"] - c["forces"]).abs().max()) <= 2e-6 * float(c["forces"].abs().max())
    assert float((a["virial"] - c["virial"]).abs().max()) <= 2e-6 * float(c["virial"].abs().max())


def test_periodic_own_list_in_fp64_builds_the_sorted_view_on_demand():
    """fp64 has no wq kernels: on this package's own periodic list (mirror map present) the sb / generic reverse kernels ask for the
    neighbor-sorted view, which EdgeGraph then builds by the stable sort -- eagerly and inside the captured periodic step
    (runtime.GraphedStepPBC, capacity form).  Energies and forces against the fp64 oracle at 1e-9, the captured step bit for bit the
    eager one."""
    from xequinet_amd import keys, runtime
    from xequinet_amd.data import NeighborTransform, XequiBatch

    dt = torch.float64
    model, oracle = _build(dt)
    f = _load("radius_graph_pbc_water192.npz")
    _, z, ptr, _ = syn.synth_water_box(4, seed=5)
    pbc = _t(np.array([[True, True, True]]))
    b = NeighborTransform(5.0)(XequiBatch(_t(f["pos"], dt), _t(z.astype(np.int32)), _t(ptr), pbc=pbc, cell=_t(f["cell"], dt)))
    data = b.to_dict()
    g = data[keys.EDGE_GRAPH]
    assert g.mirror_walk and g._n_view is None
    with torch.enable_grad():
        got = model(dict(data), compute_forces=True, compute_virial=False)
    assert g._n_view is not None                      # somebody asked: the sorted view exists now
    batch = np.zeros(len(z), dtype=np.int64)
    want = oracle({"pos": torch.tensor(f["pos"].astype(np.float64)), "atomic_numbers": torch.tensor(z.astype(np.int64)),
                   "edge_index": torch.tensor(f["edge_index"]), "batch": torch.tensor(batch), "ptr": torch.tensor(ptr),
                   "cell": torch.tensor(f["cell"].astype(np.float64)), "cell_offsets": torch.tensor(f["cell_offsets"].astype(np.float64))},
                  compute_forces=True)
    assert float((got["energy"].detach().cpu() - want["energy"]).abs().max()) <= 1e-9 * max(1.0, float(want["energy"].abs().max()))
    assert float((got["forces"].detach().cpu() - want["forces"]).abs().max()) <= 1e-9 * max(1.0, float(want["forces"].abs().max()))
    step = runtime.GraphedStepPBC(model, len(z), int(1.25 * f["edge_index"].shape[1]) + 64)
    for _ in range(2):
        out = step(_t(f["pos"], dt), _t(z.astype(np.int32)), _t(f["cell"], dt)[0], [True, True, True])
    assert int(out["n_edges"]) == f["edge_index"].shape[1]
    assert torch.equal(out["energy"], got["energy"].detach()) and torch.equal(out["forces"], got["forces"].detach())


def test_graphed_model_keeps_a_periodic_list_periodic_once_its_sorted_view_exists(monkeypatch):
    """Round-5 advisor (high): GraphedModel used to tell a periodic list from `mirror_map is not None and _n_view is None`; once anything
    had built the sorted view (an fp64 evaluation, a second capture) the capture re-created the list as an OPEN-boundary exact-mirror
    list, whose offset-blind reverse-edge map pairs every image of (i, j) with the first (j, i) slot.  A box below twice the cutoff
    (24 atoms, L = 6.2 A: several images per pair) shows it: here the sorted view is built FIRST, then the list is
    captured; the replay must equal the eager evaluation bit for bit and the fp64 oracle within the fp32 bounds."""
    from xequinet_amd import keys
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.runtime import GraphedModel

    monkeypatch.setenv("XEQ_MESSAGE_IMPL", "wq")
    pos, z, ptr, cell = syn.synth_water_box(2, seed=5)
    assert cell[0, 0, 0] < 2 * 5.0
    model, oracle = _build(torch.float32)
    pbc = _t(np.array([[True, True, True]]))
    b = NeighborTransform(5.0)(XequiBatch(_t(pos, torch.float32), _t(z), _t(ptr), pbc=pbc, cell=_t(cell, torch.float32)))
    data = b.to_dict()
    g = data[keys.EDGE_GRAPH]
    assert g.periodic and g.mirror_walk and g._n_view is None
    ei, off = b.edge_index.cpu().numpy(), b.cell_offsets.cpu().numpy()
    pairs = ei[0] * len(z) + ei[1]
    assert len(np.unique(pairs)) < len(pairs)                               # several images per pair: where the offset-blind map goes wrong
    with torch.enable_grad():
        eager = model(dict(data), compute_forces=True, compute_virial=False)
    g.n_rowptr                                                            # somebody asks for the sorted view ...
    assert g._n_view is not None and g.periodic
    gm = GraphedModel(model, tune_gemms=False)                             # ... and THEN the list is captured
    for _ in range(2):
        out = gm(dict(data))
    c = gm._last if gm._last is not None else next(iter(gm._cache.values()))
    assert c.edge_graph.periodic and c.edge_graph.mirror_map is not None and torch.equal(c.edge_graph.mirror_map, g.mirror_map)
    assert torch.equal(out["energy"], eager["energy"].detach()) and torch.equal(out["forces"], eager["forces"].detach())
    want = oracle({"pos": torch.tensor(pos, dtype=torch.float64), "atomic_numbers": torch.tensor(z.astype(np.int64)),
                   "edge_index": torch.tensor(ei), "batch": torch.zeros(len(z), dtype=torch.long), "ptr": torch.tensor(ptr),
                   "cell": torch.tensor(cell.astype(np.float64)), "cell_offsets": torch.tensor(off.astype(np.float64))}, compute_forces=True)
    dF = (out["forces"].cpu().double() - want["forces"]).abs().max()
    # (a gross check -- a wrong map moves forces by their own size; the fp32 envelope proper is asserted by the model checks below)
    assert float(dF) <= 2e-3 * max(1.0, float(want["forces"].abs().max())), float(dF)


# -------------------------------------------------------------------- whole model
def _build(dtype, **kw):
    from xequinet_amd.nn import resolve_model

    torch.manual_seed(0)
    model = resolve_model("xpainn", **kw).eval().requires_grad_(False)
    # make every affine parameter non-trivial so that a wrong index shows up
    g = torch.Generator().manual_seed(1)
    for name, p in model.named_parameters():
        if name.endswith(("norm.weight", "affine_weight")):
            p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
        elif name.endswith(("bias", "affine_bias")):
            p.copy_(0.1 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().double().clone() for k, v in model.state_dict().items()}
    oracle = orc.XPaiNNOracle(sd, **kw)
    oracle._kw = kw
    return model.to(dtype).to(DEV), oracle


def _check_model(model, oracle, pos, z, ptr, ei, dtype, extra=None, label=None):
    batch = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    ref_in = {"pos": torch.tensor(pos, dtype=torch.float64), "atomic_numbers": torch.tensor(z.astype(np.int64)),
              "edge_index": torch.tensor(ei), "batch": torch.tensor(batch), "ptr": torch.tensor(ptr)}
    data = {"pos": _t(pos, dtype), "atomic_numbers": _t(z.astype(np.int32)), "edge_index": _t(ei),
            "batch": _t(batch), "ptr": _t(ptr)}
    if extra:
        for k, v in extra.items():
            ref_in[k] = torch.tensor(v, dtype=torch.float64)
            data[k] = _t(v, dtype)
    want = oracle(ref_in, compute_forces=True)
    with torch.enable_grad():
        got = model(data, compute_forces=True, compute_virial=False)
    E, Eref = got["energy"].detach().cpu().double().numpy(), want["energy"].numpy()
    Fg, Fref = got["forces"].detach().cpu().double().numpy(), want["forces"].numpy()
    if dtype == torch.float64:
        np.testing.assert_allclose(E, Eref, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(Fg, Fref, rtol=0, atol=1e-9 * max(1.0, np.abs(Fref).max()))
        np.testing.assert_allclose(got["atomic_energies"].detach().cpu().numpy(), want["atomic_energies"].numpy(), rtol=1e-10, atol=1e-10)
    else:
        # fp32 tolerances (achieved maxima of every call: profiles/parity_r05.json):
        #   |dE| <= 1e-5 |E| + 1e-4                                  (achieved <= 0.02 of it)
        #   |dF|: f32_force_bounds() -- max and 99th percentile within max(1e-4, 1.5 x the fp32 oracle's own error on these inputs)
        dE, dF = np.abs(E - Eref), np.abs(Fg - Fref)
        bounds = f32_force_bounds(oracle, ref_in, Fref)
        b_max, b_p99, e32_max, e32_p99 = bounds
        parity_record.add(dict(config=label or f"model check N={len(pos)} E={ei.shape[1]}", max_abs_dE=float(dE.max()),
                               max_dE_over_bound=float((dE / (1e-5 * np.abs(Eref) + 1e-4)).max()), max_abs_dF=float(dF.max()),
                               p99_abs_dF=float(np.quantile(dF, 0.99)), max_abs_F=float(np.abs(Fref).max()),
                               **bounds.record(), dtype="f32 HIP vs f64 oracle"))
        assert np.all(dE <= 1e-5 * np.abs(Eref) + 1e-4), (E - Eref)
        assert dF.max() <= b_max, (dF.max(), b_max, e32_max)
        assert np.quantile(dF, 0.99) <= b_p99, (np.quantile(dF, 0.99), b_p99, e32_p99)
    return got, want


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_model_aspirin_energy_forces(dtype):
    model, oracle = _build(dtype)
    pos, z, ptr = syn.synth_aspirin()
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, 5.0)
    got, _ = _check_model(model, oracle, pos, z, ptr, ei, dtype)
    assert got["forces"].shape == (21, 3) and got["energy"].shape == (1,)
    assert got["forces"].sum(0).abs().max().item() < (1e-9 if dtype == torch.float64 else 1e-3)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_model_qm9_batch_energy_forces(dtype):
    model, oracle = _build(dtype)
    pos, z, ptr = syn.synth_qm9_batch(48, seed=21)
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, 5.0)
    _check_model(model, oracle, pos, z, ptr, ei, dtype)


def test_model_batch_with_lone_atoms_through_neighbor_transform():
    """Edge cases of the whole pipeline (NeighborTransform -> symmetric EdgeGraph -> fused blocks): a one-atom molecule
    in the middle of a batch, a far-away atom inside a molecule, and a batch that is only a lone atom (no edge at all);
    energies / forces against the oracle, eager and through the HIP-graph replay."""
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.runtime import GraphedModel

    model, oracle = _build(torch.float64)
    gm = GraphedModel(model, tune_gemms=False)
    pos, z, ptr = syn.synth_qm9_batch(4, seed=2)
    cut = int(ptr[2])
    pos = np.concatenate([pos[:cut], [[50.0, 50.0, 50.0]], pos[cut:]])        # a one-atom "molecule" as graph 2
    z = np.concatenate([z[:cut], [8], z[cut:]])
    ptr = np.concatenate([ptr[:3], ptr[2:] + 1])
    pos[0] += 200.0                                                            # and an atom with no neighbour
    cases = [(pos, z, ptr), (np.array([[0.0, 0.0, 0.0]]), np.array([6]), np.array([0, 1]))]
    for p, zz, pp in cases:
        b = NeighborTransform(5.0)(XequiBatch(_t(p, torch.float64), _t(zz.astype(np.int32)), _t(pp)))
        ei = b.edge_index.cpu().numpy()
        np.testing.assert_array_equal(ei, orc.radius_graph_canonical(p, pp, 5.0))
        d = b.to_dict()
        got, want = _check_model(lambda data, **kw: model({**d, "pos": data["pos"]}, **kw), oracle, p, zz, pp, ei, torch.float64)
        rep = gm(d)
        assert torch.equal(rep["energy"], got["energy"].detach()) and torch.equal(rep["forces"], got["forces"].detach())
        assert got["energy"].shape == (len(pp) - 1,)
    assert ei.shape[1] == 0 and float(got["forces"].abs().max()) == 0.0


@pytest.mark.parametrize("kw", [
    dict(node_dim=32, node_irreps="32x0e+16x1o+8x2e", num_basis=8, cutoff=4.0, action_blocks=2, hidden_dim=16),
    dict(node_dim=16, node_irreps="16x1o", num_basis=8, cutoff=4.0, action_blocks=2, layer_norm=False, embed_basis="one-hot"),
    dict(node_dim=64, node_irreps="64x0e+32x1o", rbf_kernel="gaussian", cutoff_fn="polynomial", aux_basis="aux28"),
])
def test_model_other_configs_fp64(kw):
    model, oracle = _build(torch.float64, **kw)
    pos, z, ptr = syn.synth_qm9_batch(7, seed=5)
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, kw.get("cutoff", 5.0))
    _check_model(model, oracle, pos, z, ptr, ei, torch.float64)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_model_pbc_water_energy_forces(dtype):
    model, oracle = _build(dtype)
    f = _load("radius_graph_pbc_water192.npz")
    pos, z, ptr, cell = syn.synth_water_box(4, seed=5)
    # 192 atoms at ~54 neighbours each: the densest graph of the suite and only 576 force components, so the 99th
    # percentile is its 6th largest error (the fp32 oracle itself: ~2e-4 here, which is what the bound follows)
    _check_model(model, oracle, f["pos"].astype(np.float64), z, ptr, f["edge_index"], dtype,
                 extra={"cell": f["cell"].astype(np.float64), "cell_offsets": f["cell_offsets"].astype(np.float64)},
                 label="water_192 (periodic, golden edge list)")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_model_virial_pbc_and_molecules(dtype):
    """Virial / strain branch (nn/basic.py:93-107,162-199): forces + virial, and virial alone, on a periodic water box
    (edge vectors with cell offsets) and on a molecule batch with shuffled (not center-sorted) edges, against the
    oracle, whose virial is pinned by finite differences on the CPU."""
    model, oracle = _build(dtype)
    f = _load("radius_graph_pbc_water192.npz")
    _, z, ptr, _ = syn.synth_water_box(4, seed=5)
    pos_m, z_m, ptr_m = syn.synth_qm9_batch(12, seed=21)
    ei_m = orc.radius_graph_canonical(pos_m.astype(np.float32), ptr_m, 5.0)
    ei_m = ei_m[:, np.random.default_rng(1).permutation(ei_m.shape[1])]
    cases = [(f["pos"].astype(np.float64), z, ptr, f["edge_index"],
              {"cell": f["cell"].astype(np.float64), "cell_offsets": f["cell_offsets"].astype(np.float64)}),
             (pos_m, z_m, ptr_m, ei_m, {})]
    for pos, zz, pp, ei, extra in cases:
        batch = np.repeat(np.arange(len(pp) - 1), np.diff(pp))
        ref_in = {"pos": torch.tensor(pos), "atomic_numbers": torch.tensor(zz.astype(np.int64)), "edge_index": torch.tensor(ei),
                  "batch": torch.tensor(batch), "ptr": torch.tensor(pp)}
        mk = lambda: {"pos": _t(pos, dtype), "atomic_numbers": _t(zz.astype(np.int32)), "edge_index": _t(ei), "batch": _t(batch),
                      "ptr": _t(pp), **{k: _t(v, dtype) for k, v in extra.items()}}
        for k, v in extra.items():
            ref_in[k] = torch.tensor(v)
        want = oracle(ref_in, compute_forces=True, compute_virial=True)
        with torch.enable_grad():
            got = model(mk(), compute_forces=True, compute_virial=True)
            only = model(mk(), compute_forces=False, compute_virial=True)
        V, Vref = got["virial"].detach().cpu().double().numpy(), want["virial"].numpy()
        Fg, Fref = got["forces"].detach().cpu().double().numpy(), want["forces"].numpy()
        assert V.shape == Vref.shape == (len(pp) - 1, 3, 3)
        # fp32: the reference's own fp32 evaluation is ~2e-4 from fp64 on 50-neighbour graphs (see _check_model)
        tol = 1e-9 if dtype == torch.float64 else 2e-3
        np.testing.assert_allclose(V, Vref, rtol=0, atol=tol * max(1.0, np.abs(Vref).max()))
        np.testing.assert_allclose(Fg, Fref, rtol=0, atol=tol * max(1.0, np.abs(Fref).max()))
        np.testing.assert_allclose(only["virial"].detach().cpu().double().numpy(), V, rtol=0, atol=1e-12 if dtype == torch.float64 else 1e-5)
        assert "forces" not in only


@pytest.mark.parametrize("impl", ["generic", "sb", "wq"])
def test_model_message_kernel_families_agree(impl, monkeypatch):
    """The fused-message kernel families (generic, scalar-broadcast, wave / matrix-core) and the
    operator-level module path all reproduce the oracle on a molecule batch (fp32)."""
    monkeypatch.setenv("XEQ_MESSAGE_IMPL", impl)
    model, oracle = _build(torch.float32)
    pos, z, ptr = syn.synth_qm9_batch(40, seed=8)
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, 5.0)
    _check_model(model, oracle, pos, z, ptr, ei, torch.float32)
    if impl == "generic":  # operator-level drop-in path: the reference's op sequence on the HIP ops
        for m in model.mods.values():
            if hasattr(m, "fused"):
                m.fused = False
        _check_model(model, oracle, pos, z, ptr, ei, torch.float32)


def test_model_pipeline_with_neighbor_transform_and_unsorted_edges():
    """NeighborTransform -> model gives the same result as an externally built,
    randomly permuted edge list (index_add is order independent up to rounding)."""
    from xequinet_amd.data import NeighborTransform, XequiBatch

    model, oracle = _build(torch.float64)
    pos, z, ptr = syn.synth_qm9_batch(9, seed=77)
    batch = XequiBatch(_t(pos, torch.float64), _t(z), _t(ptr))
    batch = NeighborTransform(5.0)(batch)
    with torch.enable_grad():
        a = model(batch.to_dict(), compute_forces=True)
    ei = batch.edge_index.cpu().numpy()
    ei = ei[:, np.random.default_rng(0).permutation(ei.shape[1])]
    _, want = _check_model(model, oracle, pos, z, ptr, ei, torch.float64)
    np.testing.assert_allclose(a["energy"].detach().cpu().numpy(), want["energy"].numpy(), rtol=1e-10)
    np.testing.assert_allclose(a["forces"].cpu().numpy(), want["forces"].numpy(), atol=1e-9)


def test_full_size_properties_qm9_1024():
    """BASELINE config 2 at full size through size-independent properties:
    per-molecule sum of forces = 0, rotation + translation invariance of energies,
    forces co-rotate, molecule order permutation."""
    from xequinet_amd.data import NeighborTransform, XequiBatch

    model, _ = _build(torch.float32)
    pos, z, ptr = syn.synth_qm9_batch(1024, seed=1234)
    assert len(pos) == 18609

    def run(p, zz, pp):
        b = NeighborTransform(5.0)(XequiBatch(_t(p, torch.float32), _t(zz), _t(pp)))
        with torch.enable_grad():
            out = model(b.to_dict(), compute_forces=True)
        return out["energy"].detach().cpu().double().numpy(), out["forces"].cpu().double().numpy(), b.edge_index.shape[1]

    E, Fo, n_edges = run(pos, z, ptr)
    assert n_edges == 311994  # this generator; SURVEY 8d-2 quotes 300 406 for its own draw of the same recipe
    assert np.isfinite(E).all() and np.isfinite(Fo).all()
    seg = np.repeat(np.arange(1024), np.diff(ptr))
    net = np.zeros((1024, 3))
    np.add.at(net, seg, Fo)
    assert np.abs(net).max() < 2e-5 * max(1.0, np.abs(Fo).max()) * 30
    # rigid motion
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] *= -1
    E2, F2, _ = run(pos @ Q.T + np.array([1.0, -2.0, 0.5]), z, ptr)
    assert np.all(np.abs(E2 - E) <= 2e-5 * np.abs(E) + 2e-4)
    # fp32: a few molecules are ill-conditioned (the fp32 CPU oracle itself is rotation-consistent
    # only to ~1e-3 on them), so bound the bulk tightly and the worst case loosely
    fscale = max(1.0, np.abs(Fo).max())
    dev = np.abs(F2 - Fo @ Q.T)
    assert np.quantile(dev, 0.999) < 1e-4 * fscale and dev.max() < 1e-2 * fscale, (np.quantile(dev, 0.999), dev.max(), fscale)
    # reverse molecule order
    order = np.arange(1024)[::-1]
    idx = np.concatenate([np.arange(ptr[g], ptr[g + 1]) for g in order])
    ptr2 = np.concatenate([[0], np.cumsum(np.diff(ptr)[order])])
    E3, F3, _ = run(pos[idx], z[idx], ptr2)
    assert np.all(np.abs(E3 - E[order]) <= 1e-5 * np.abs(E[order]) + 1e-4)
    assert np.abs(F3 - Fo[idx]).max() < 1e-5 * fscale


def test_full_size_properties_md17_4096_and_water_512():
    """BASELINE configs 3 and 4 at full size through size-independent properties.
    MD17 x 4096 (N = 86 016): per-frame net force = 0, every frame's energy/forces independent of its place in the
    batch (frames 0..15 re-evaluated alone), bitwise reproducible.  Water-512 (PBC, ~51 neighbours/atom): net force = 0
    per box, energy invariant and forces unchanged under a rigid translation by a non-lattice vector, edge count
    symmetric (every edge has its reverse)."""
    from xequinet_amd.data import NeighborTransform, XequiBatch

    model, _ = _build(torch.float32)
    p0, z0, _ = syn.synth_aspirin()
    rng = np.random.default_rng(11)
    n_fr = 4096
    pos = (p0[None] + rng.normal(0, 0.05, size=(n_fr, 21, 3))).reshape(-1, 3)
    z = np.tile(z0, n_fr)
    ptr = np.arange(0, (n_fr + 1) * 21, 21, dtype=np.int64)

    def run(p, zz, pp, **kw):
        b = NeighborTransform(5.0)(XequiBatch(_t(p, torch.float32), _t(zz), _t(pp), **kw))
        with torch.enable_grad():
            out = model(b.to_dict(), compute_forces=True)
        return out["energy"].detach().cpu().double().numpy(), out["forces"].cpu().double().numpy(), b

    E, F, b = run(pos, z, ptr)
    assert F.shape == (86016, 3) and np.isfinite(E).all() and np.isfinite(F).all()
    net = F.reshape(n_fr, 21, 3).sum(1)
    assert np.abs(net).max() < 1e-3 * max(1.0, np.abs(F).max())
    E2, F2, _ = run(pos, z, ptr)
    assert np.array_equal(E, E2) and np.array_equal(F, F2)
    Es, Fs, _ = run(pos[: 16 * 21], z[: 16 * 21], ptr[:17])
    np.testing.assert_allclose(Es, E[:16], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(Fs, F[: 16 * 21], rtol=0, atol=1e-5 * max(1.0, np.abs(F).max()))

    posw, zw, ptrw, cell = syn.synth_water_box(8, seed=5)
    kw = lambda: dict(pbc=torch.tensor([[True, True, True]], device=DEV), cell=_t(cell, torch.float32))
    Ew, Fw, bw = run(posw, zw, ptrw, **kw())
    n_edges = bw.edge_index.shape[1]
    assert 45 * 1536 < n_edges < 57 * 1536
    ei = bw.edge_index.cpu().numpy()
    fwd = np.sort(ei[0].astype(np.int64) * 1536 + ei[1]), np.sort(ei[1].astype(np.int64) * 1536 + ei[0])
    assert np.array_equal(*fwd)
    fscale = max(1.0, np.abs(Fw).max())
    assert np.abs(Fw.sum(0)).max() < 2e-3 * fscale
    Et, Ft, bt = run(posw + np.array([0.37, -1.21, 2.05]), zw, ptrw, **kw())
    assert bt.edge_index.shape[1] == n_edges
    assert abs(Et[0] - Ew[0]) <= 2e-5 * abs(Ew[0]) + 1e-3
    dev = np.abs(Ft - Fw)
    # fp32 on a 51-neighbour graph at |pos| ~ 25 A: re-wrapping changes every rounding; bulk and worst case bounded loosely
    assert np.quantile(dev, 0.99) < 2e-3 * fscale and dev.max() < 1e-2 * fscale, (np.quantile(dev, 0.99), dev.max())


def test_graphed_model_replays_bitwise_and_recaptures_on_new_shapes():
    """runtime.GraphedModel: HIP-graph replay of one evaluation equals the eager path bit for bit, also after the atoms
    move (same topology: replay; different edge count: a new capture), for a molecule batch and a periodic box."""
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.runtime import GraphedModel

    model, _ = _build(torch.float32)
    gm = GraphedModel(model, tune_gemms=False)    # bitwise comparison with the eager path: same library GEMM picks
    tr = NeighborTransform(5.0)
    pos, z, ptr = syn.synth_qm9_batch(5, seed=9)
    rng = np.random.default_rng(0)

    def both(p, zz, pp, **kw):
        d = tr(XequiBatch(_t(p, torch.float32), _t(zz), _t(pp), **kw)).to_dict()
        with torch.enable_grad():
            want = model(dict(d), compute_forces=True)
        got = gm(d)
        for k in ("energy", "forces", "atomic_energies"):
            assert torch.equal(got[k], want[k].detach()), k
        return d["edge_index"].shape[1]

    e0 = both(pos, z, ptr)
    assert gm.captures == 1
    e1 = both(pos + rng.normal(0, 1e-3, size=pos.shape), z, ptr)      # same topology
    assert e1 == e0 and gm.captures == 1
    p2 = pos.copy()
    p2[: ptr[1]] *= 2.5                                                  # first molecule blown up: fewer edges
    e2 = both(p2, z, ptr)
    assert e2 < e0 and gm.captures == 2
    both(pos, z, ptr)
    assert gm.captures == 2                                              # first signature still cached
    posw, zw, ptrw, cell = syn.synth_water_box(3, seed=2)
    kw = dict(pbc=torch.tensor([[True, True, True]], device=DEV), cell=_t(cell, torch.float32))
    both(posw, zw, ptrw, **kw)
    both(posw + 0.01, zw, ptrw, **dict(kw))
    assert gm.captures in (3, 4)


def test_reused_edge_graph_follows_new_positions():
    """A transformed batch evaluated again after its atoms moved (NeighborTransform returns early when edge_index is
    set, XequiBatch.to_dict() re-attaches the same EdgeGraph): the per-edge records cached on the graph must not
    survive the geometry they were computed from.  Compared with a fresh transform of the moved batch."""
    from xequinet_amd.data import NeighborTransform, XequiBatch

    model, _ = _build(torch.float32)
    tr = NeighborTransform(5.0)
    pos, z, ptr = syn.synth_qm9_batch(6, seed=31)
    batch = tr(XequiBatch(_t(pos, torch.float32), _t(z), _t(ptr)))
    graph = batch.to_dict()["_xeq_edge_graph"]
    rng = np.random.default_rng(3)
    outs = []
    for step in range(3):
        batch.pos = _t(pos + 2e-4 * step * rng.normal(size=pos.shape), torch.float32)   # same topology, new geometry
        d = tr(batch).to_dict()
        assert d["_xeq_edge_graph"] is graph
        with torch.enable_grad():
            got = model(d, compute_forces=True)
        fresh = tr(XequiBatch(batch.pos.detach().clone(), _t(z), _t(ptr)))
        assert torch.equal(fresh.edge_index, batch.edge_index)
        with torch.enable_grad():
            want = model(fresh.to_dict(), compute_forces=True)
        assert torch.equal(got["energy"], want["energy"]) and torch.equal(got["forces"], want["forces"]), step
        outs.append(got["energy"].detach().clone())
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])


def test_auto_picks_a_kernel_family_that_fits(monkeypatch):
    """XEQ_MESSAGE_IMPL=auto looks at the sizes too: sb where it is the faster family (ops.prefers_sb), beyond the 32-bit byte offsets of the matrix-core kernels it takes
    the scalar-broadcast form, beyond that one's 32-bit element offsets the generic form; f64 never takes the matrix-core kernels."""
    from xequinet_amd import ops

    monkeypatch.delenv("XEQ_MESSAGE_IMPL", raising=False)
    mul = (128, 64, 32)
    pick = lambda dt, n, e: ops.select_message_impl(dt, n, e, 20, 128, mul)
    assert pick(torch.float32, 18_609, 311_994) == "wq"
    assert pick(torch.float32, 1_200_000, 14_000_000) == "wq"
    assert pick(torch.float32, 1_200_000, 31_000_000) == "sb"          # records: padded slots * 160 B >= 2^32
    assert pick(torch.float32, 1_900_000, 1_000_000) == "sb"           # rows of h: N * 576 * 4 B >= 2^32
    assert pick(torch.float32, 4_000_000, 1_000_000) == "generic"      # N * 576 elements >= 2^31
    assert pick(torch.float64, 18_609, 311_994) == "sb"
    assert ops.select_message_impl(torch.float32, 100, 1000, 20, 96, (96, 48, 24)) == "sb"   # multiplicities not in 32s
    assert ops.select_message_impl(torch.float32, 10_000, 100_000, 30, 128, mul) == "wq"      # num_basis up to 31 since round 4
    assert ops.select_message_impl(torch.float32, 10_000, 100_000, 32, 128, mul) == "sb"      # 32: the scalar-broadcast form
    assert pick(torch.float32, 1_536, 82_996) == "wq"                  # dense neighbourhoods (water box): wq too since the split-bf16 filter
    assert pick(torch.float32, 21, 360) == "sb"                        # one small molecule: launch-bound, sb needs no walk plan
    assert pick(torch.float32, 1_175, 19_984) == "wq"                  # 64 QM9-shaped molecules: wq from there on
    monkeypatch.setenv("XEQ_MESSAGE_IMPL", "wq")
    with pytest.raises(RuntimeError):
        ops.select_message_impl(torch.float64, 100, 1000, 20, 128, mul)


@pytest.mark.parametrize("n", [0, 1, 8191, 8193, 100003])
def test_exclusive_scan_grid_wide_matches_numpy(n):
    """xeq_exclusive_scan_i32_ws (grid-wide) and xeq_exclusive_scan_i32 (one workgroup): rowptr of the neighbour counts, bit-exact."""
    from xequinet_amd import lib
    from xequinet_amd.lib import call, ptr, stream
    from xequinet_amd.ops import _exclusive_scan

    rng = np.random.default_rng(n)
    deg = rng.integers(0, 40, size=n).astype(np.int32)
    want = np.concatenate([[0], np.cumsum(deg, dtype=np.int64)]).astype(np.int32)
    d = _t(deg) if n else torch.empty(0, dtype=torch.int32, device=DEV)
    out = torch.empty(n + 1, dtype=torch.int32, device=DEV)
    _exclusive_scan(d, n, out)
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    out1 = torch.empty(n + 1, dtype=torch.int32, device=DEV)
    call("xeq_exclusive_scan_i32", ptr(d), n, ptr(out1), stream())
    np.testing.assert_array_equal(out1.cpu().numpy(), want)
    assert lib.load().xeq_exclusive_scan_i32_workspace(-1) == -1


@pytest.mark.parametrize("mult,eps", [(3, 24), (2, 32), (4, 16)])
def test_wq_stream_classes_change_no_bit(mult, eps, monkeypatch):
    """Two stream classes (csrc/xeq_message_wq.hip::wq_long_mult; measured slower and switched off, XEQ_WQ_LONG_MULT brings them back):
    the l = 0 units walk `mult` table streams at a time, the l > 0 units the table streams themselves.  A node's sums are running sums
    over ITS edges in walk order whatever the cut of the walk into streams, so forward outputs, node gradients and dL/dvec are bit for
    bit those of one class -- the same property the sharding tests rest on."""
    from xequinet_amd import ops

    monkeypatch.setenv("XEQ_MESSAGE_IMPL", "wq")
    rng = np.random.default_rng(8)
    pos, z, ptr = syn.synth_qm9_batch(96, seed=23)
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, 5.0)
    N, E = len(pos), ei.shape[1]
    node_dim, mul, B = 128, (128, 64, 32), 20
    C, D = sum(mul), mul[0] + 3 * mul[1] + 5 * mul[2]
    H = node_dim + 2 * C
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    h, xhat, s, x = (f32(rng.normal(size=sh)) for sh in ((N, H), (N, D), (N, node_dim), (N, D)))
    W, b = f32(rng.normal(size=(H, B)) / math.sqrt(B)), f32(rng.normal(size=(H,)))
    p0 = f32(math.pi * np.arange(1, B + 1) / 5.0).view(1, -1)
    vec = f32(pos[ei[0]] - pos[ei[1]])
    gs, gx = f32(rng.normal(size=(N, node_dim))), f32(rng.normal(size=(N, D)))
    cfg = ("bessel", "cosine", B, 5.0, node_dim, mul)
    outs = []
    for m_ in (1, mult):
        monkeypatch.setenv("XEQ_WQ_LONG_MULT", str(m_))
        monkeypatch.setenv("XEQ_WQ_EDGES_PER_STREAM", str(eps))
        graph = ops.EdgeGraph(_t(ei), N, ptr=_t(ptr), symmetric=True)
        s_out, x_out, saved, impl = ops.message_forward(h, xhat, vec, s, x, W, b, p0, None, graph, cfg, want_backward=True)
        assert impl == "wq"
        g = ops.message_backward(saved, graph, cfg, impl, gs, gx, node_grads=True)
        outs.append((s_out, x_out, g[0], g[1], g[2]))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("irreps,node_dim,B", [("128x0e + 64x1o + 32x2e", 128, 20), ("32x0e + 64x2e", 32, 17), ("64x0e + 32x1o + 32x2e", 64, 8)])
@pytest.mark.parametrize("layout", [0, 1])
def test_wq_first_block_hint_changes_nothing(irreps, node_dim, B, layout, monkeypatch):
    """XEQ_XHAT_HIGHER_L_ZERO (include/xeq.h): with xhat zero on the l > 0 columns -- the model's first message block --
    the hinted kernels skip the gate_state terms there and, in a force evaluation, the value filters of the reverse pass.
    Forward: the same bits as the general kernel; reverse (no node gradients): dL/dvec to rounding (the order of the
    per-edge sums is unchanged, a zero term is left out)."""
    from xequinet_amd import lib, ops

    monkeypatch.setenv("XEQ_MESSAGE_IMPL", "wq")
    rng = np.random.default_rng(5)
    pos, z, ptr = syn.synth_qm9_batch(40, seed=21)
    rc = 4.0
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, rc)
    N, E = len(pos), ei.shape[1]
    mul = [0, 0, 0]
    for m_, l_, _ in orc.parse_irreps(irreps):
        mul[l_] = m_
    C, D = sum(mul), mul[0] + 3 * mul[1] + 5 * mul[2]
    H = node_dim + 2 * C
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    h, s = f32(rng.normal(size=(N, H))), f32(rng.normal(size=(N, node_dim)))
    x = torch.zeros(N, D, device=DEV)
    xhat = torch.zeros(N, D, device=DEV)
    bias0 = f32(rng.normal(size=mul[0]))
    if layout == 0:
        xhat[:, : mul[0]] = bias0                                # e3nn layout: the 0e block leads every row
    else:
        xhat = xhat.reshape(-1)
        xhat[: N * mul[0]] = bias0.repeat(N)                     # BT layout: block l = 0 is [N, mul0] row-major
    W, b = f32(rng.normal(size=(H, B)) / math.sqrt(B)), f32(rng.normal(size=(H,)))
    p0 = f32(math.pi * np.arange(1, B + 1) / rc).view(1, -1)
    vec = f32(pos[ei[0]] - pos[ei[1]])
    gs, gx = f32(rng.normal(size=(N, node_dim))), f32(rng.normal(size=(N, D)))
    outs = []
    for hint in (0, lib.XHAT_HIGHER_L_ZERO):
        graph = ops.EdgeGraph(_t(ei), N, ptr=_t(ptr))
        cfg = ("bessel", "cosine", B, rc, node_dim, tuple(mul), layout | hint)
        s_out, x_out, saved, impl = ops.message_forward(h, xhat, vec, s, x, W, b, p0, None, graph, cfg)
        assert impl == "wq"
        g_h, g_xhat, g_vec, _, _ = ops.message_backward(saved, graph, cfg, impl, gs, gx, node_grads=False)
        assert g_h is None and g_xhat is None
        g_full = ops.message_backward(saved, graph, cfg, impl, gs, gx, node_grads=True)
        outs.append((s_out, x_out, g_vec, g_full))
    (s0, x0, v0, f0), (s1, x1, v1, f1) = outs
    assert torch.equal(s0, s1) and torch.equal(x0, x1)
    scale = v0.abs().max().item()
    assert (v0 - v1).abs().max().item() <= 2e-6 * scale
    for a_, b_ in zip(f0, f1):      # with node gradients wanted the hint does not apply to the reverse kernel
        assert torch.equal(a_, b_)


def test_copy_many_one_launch_for_a_list_of_copies():
    """ops.copy_many / xeq_copy_many: what HIP-graph replay refreshes its captured inputs with.  Sizes from one word to a few
    MB, 4-byte aligned views (the narrow path), more buffers than one launch takes, and the Tensor.copy_ fall-backs."""
    from xequinet_amd import ops

    g = torch.Generator(device=DEV).manual_seed(0)
    pairs = []
    for i, n in enumerate([1, 3, 4, 17, 1000, 4096, 65537, 1 << 20, 5, 12, 999, 2, 8, 16, 33, 64, 127, 300001, 7]):
        dt = (torch.float32, torch.int32, torch.int64, torch.float64)[i % 4]
        src = torch.randint(-1000, 1000, (n + 1,), device=DEV, generator=g).to(dt)
        dst = torch.zeros(n + 1, dtype=dt, device=DEV)
        if i % 3 == 0:
            src, dst = src[1:], dst[1:]                    # 4-byte (or 8-byte) aligned, not 16
        else:
            src, dst = src[:n], dst[:n]
        pairs.append((dst, src))
    mism = (torch.zeros(10, dtype=torch.int64, device=DEV), torch.arange(10, dtype=torch.int32, device=DEV))   # dtype conversion
    strided = (torch.zeros(6, 2, device=DEV), torch.arange(24., device=DEV).reshape(6, 4)[:, ::2])             # non-contiguous source
    ops.copy_many(pairs + [mism, strided])
    for dst, src in pairs:
        assert torch.equal(dst, src)
    assert torch.equal(mism[0], mism[1].long()) and torch.equal(strided[0], strided[1])
    ops.copy_many([])


@pytest.mark.parametrize("first_block", [False, True])
def test_wq_reverse_mirror_walk_is_the_reverse_plan_bit_for_bit(first_block, monkeypatch):
    """Symmetric center-sorted lists (what NeighborTransform builds for open boundaries): the reverse wq kernel walks the FORWARD
    plan and records, every slot standing for its mirror edge (XEQ_WQ_MIRROR_WALK) -- no reverse plan, no reverse records.  Same
    edges in the same order with the same bits (the mirror's vector is the negated vector): every gradient equals the reverse-plan
    walk's bit for bit, general and first-block forms."""
    from xequinet_amd import lib, ops
    from xequinet_amd.data import NeighborTransform, XequiBatch

    monkeypatch.setenv("XEQ_MESSAGE_IMPL", "wq")
    pos, z, ptr = syn.synth_qm9_batch(96, seed=5)
    b = NeighborTransform(5.0)(XequiBatch(_t(pos.astype(np.float32)), _t(z), _t(ptr)))
    ei, N = b.edge_index, len(pos)
    E = ei.shape[1]
    mul, F, B, rc = (128, 64, 32), 128, 20, 5.0
    C, D = 224, 480
    H = F + 2 * C
    g = torch.Generator().manual_seed(11)
    r = lambda *shape: torch.randn(*shape, generator=g).to(DEV)
    h, s, x = r(N, H), r(N, F), r(N, D)
    xhat = r(N * D)
    if first_block:                      # zero on the l > 0 columns (BT layout: the l = 0 block comes first)
        xhat[N * mul[0]:] = 0
    W, bias = r(H, B) / math.sqrt(B), r(H)
    p0 = (math.pi * torch.arange(1, B + 1, device=DEV) / rc).float().view(1, -1)
    gs, gx = r(N, F), r(N, D)
    vec = (b.pos[ei[0]] - b.pos[ei[1]]).contiguous()
    cfg = ("bessel", "cosine", B, rc, F, mul, 1 | (lib.XHAT_HIGHER_L_ZERO if first_block else 0))
    results = []
    for mirror in (True, False):
        graph = ops.EdgeGraph(ei, N, center_sorted=True, ptr=b.ptr, symmetric=True)
        assert graph.mirror_walk
        graph.mirror_walk = mirror
        s_out, x_out, saved, impl = ops.message_forward(h, xhat, vec, s, x, W, bias, p0, None, graph, cfg, want_backward=True)
        assert impl == "wq"
        for node_grads in ((True, False) if first_block else (True,)):
            out = ops.message_backward(saved, graph, cfg, impl, gs, gx, node_grads=node_grads)
            results.append((mirror, node_grads, [s_out, x_out] + [t for t in out[:3] if t is not None]))
        assert (graph._wq is not None) and ((True, ops._wq_edges_per_stream(E, N)) in graph._wq) == (not mirror)   # no reverse plan under the mirror walk
    half = len(results) // 2
    for (m1, n1, a), (m2, n2, c) in zip(results[:half], results[half:]):
        assert m1 and not m2 and n1 == n2 and len(a) == len(c)
        for u, v in zip(a, c):
            assert torch.equal(u, v)

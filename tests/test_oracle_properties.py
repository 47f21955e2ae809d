"""Known-answer / property tests that pin the e3nn-dependent part of the oracle
(e3nn is not installable here: SURVEY 8c)."""
import math

import numpy as np
import pytest
import torch

from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn

IRREPS = "128x0e + 64x1o + 32x2e"


def test_sh_known_answers_and_layout():
    # direct call in ORIGINAL axis order -> reference passes vec[:, [1,2,0]]
    def Y(v):
        v = torch.tensor([v], dtype=torch.float64)
        return orc.spherical_harmonics("1x0e+1x1o+1x2e", v[:, [1, 2, 0]])[0].numpy()

    s3, s5, s15 = math.sqrt(3), math.sqrt(5), math.sqrt(15)
    np.testing.assert_allclose(Y([0, 0, 2.0]), [1, 0, s3, 0, 0, 0, s5, 0, 0], atol=1e-14)       # r = z
    np.testing.assert_allclose(Y([3.0, 0, 0]), [1, 0, 0, s3, 0, 0, -s5 / 2, 0, s15 / 2], atol=1e-14)  # r = x
    np.testing.assert_allclose(Y([0, 1.0, 0]), [1, s3, 0, 0, 0, 0, -s5 / 2, 0, -s15 / 2], atol=1e-14)  # r = y
    v = torch.randn(50, 3, dtype=torch.float64)
    y = orc.spherical_harmonics("1x0e+1x1o+1x2e", v[:, [1, 2, 0]])
    np.testing.assert_allclose((y[:, 1:4] ** 2).sum(-1).numpy(), 3.0, rtol=1e-13)   # sum_m Y_lm^2 = 2l+1
    np.testing.assert_allclose((y[:, 4:9] ** 2).sum(-1).numpy(), 5.0, rtol=1e-13)
    np.testing.assert_allclose(y[:, 1:4].numpy(), (math.sqrt(3) * v[:, [1, 2, 0]] / v.norm(dim=-1, keepdim=True)).numpy(), rtol=1e-13)
    y2 = orc.spherical_harmonics("1x0e+1x1o+1x2e", 7.3 * v[:, [1, 2, 0]])                # scale invariance
    np.testing.assert_allclose(y.numpy(), y2.numpy(), atol=1e-13)
    full = orc.spherical_harmonics(IRREPS, v[:, [1, 2, 0]])                              # repetition pattern
    assert full.shape == (50, 480)
    np.testing.assert_array_equal(full[:, :128].numpy(), 1.0)
    np.testing.assert_array_equal(full[:, 128:320].reshape(50, 64, 3).numpy(), y[:, None, 1:4].expand(50, 64, 3).numpy())
    np.testing.assert_array_equal(full[:, 320:].reshape(50, 32, 5).numpy(), y[:, None, 4:9].expand(50, 32, 5).numpy())


def test_o3_linear_and_layernorm_properties():
    torch.manual_seed(0)
    x = torch.randn(4000, 480, dtype=torch.float64)
    W = torch.randn(21504, dtype=torch.float64)
    out = orc.o3_linear(IRREPS, x, W, torch.zeros(128, dtype=torch.float64))
    assert abs(out.var().item() - 1.0) < 0.05                                   # variance preserving
    x2 = x.clone()
    x2[:, 128:] = 0
    out2 = orc.o3_linear(IRREPS, x2, W, None)
    assert out2[:, 128:].abs().max() == 0                                       # block diagonal in l
    b = torch.randn(128, dtype=torch.float64)
    d = orc.o3_linear(IRREPS, x, W, b) - orc.o3_linear(IRREPS, x, W, None)
    np.testing.assert_allclose(d[:, :128].numpy(), b.expand(4000, 128).numpy(), atol=1e-12)
    assert d[:, 128:].abs().max() == 0                                          # bias on 0e only
    z = orc.equivariant_layer_norm(IRREPS, torch.zeros(3, 480, dtype=torch.float64), torch.ones(224, dtype=torch.float64), b)
    np.testing.assert_allclose(z[:, :128].numpy(), b.expand(3, 128).numpy(), atol=0)
    assert z[:, 128:].abs().max() == 0                                          # zeros -> zeros (+bias)
    y = orc.equivariant_layer_norm(IRREPS, x, torch.ones(224, dtype=torch.float64), torch.zeros(128, dtype=torch.float64))
    np.testing.assert_allclose(orc.invariant(IRREPS, y, squared=True).mean(1).numpy(), 1.0, rtol=1e-4)  # unit mean channel norm
    np.testing.assert_allclose(y[:, :128].mean(1).numpy(), 0.0, atol=1e-12)


def _small_model(dtype=torch.float64, **kw):
    from xequinet_amd.nn import resolve_model

    torch.manual_seed(0)
    kw = dict(dict(node_dim=32, node_irreps="32x0e+16x1o+8x2e", num_basis=8, action_blocks=2, hidden_dim=16), **kw)
    m = resolve_model("xpainn", **kw)
    g = torch.Generator().manual_seed(1)
    sd = {}
    for k, v in m.state_dict().items():
        v = v.detach().double().clone()
        if k.endswith(("bias", "affine_bias")) and v.numel():
            v = 0.1 * torch.randn(v.shape, generator=g, dtype=torch.float64)
        sd[k] = v
    return orc.XPaiNNOracle(sd, **kw), kw


def _inputs(seed=3, n_mol=3):
    pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=seed)
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, 5.0)
    batch = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    return pos, z, ptr, ei, batch


def _run(model, pos, z, ptr, ei, batch):
    return model({"pos": torch.tensor(pos), "atomic_numbers": torch.tensor(z.astype(np.int64)), "edge_index": torch.tensor(ei),
                  "batch": torch.tensor(batch), "ptr": torch.tensor(ptr)})


def test_energy_invariance_force_equivariance_and_finite_differences():
    model, _ = _small_model()
    pos, z, ptr, ei, batch = _inputs()
    out = _run(model, pos, z, ptr, ei, batch)
    E, F_ = out["energy"].numpy(), out["forces"].numpy()
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))  # proper or improper: both are symmetries of an energy model
    out2 = _run(model, pos @ Q.T + np.array([0.3, -1.0, 2.0]), z, ptr, ei, batch)
    np.testing.assert_allclose(out2["energy"].numpy(), E, rtol=1e-12)
    np.testing.assert_allclose(out2["forces"].numpy(), F_ @ Q.T, atol=1e-11)
    # sum of forces per molecule vanishes
    net = np.zeros((len(ptr) - 1, 3))
    np.add.at(net, batch, F_)
    assert np.abs(net).max() < 1e-11
    # permutation of atoms inside the batch (with relabelled edges)
    perm = rng.permutation(ptr[1])  # shuffle the first molecule
    full = np.concatenate([perm, np.arange(ptr[1], len(pos))])
    inv = np.argsort(full)
    out3 = _run(model, pos[full], z[full], ptr, inv[ei], batch)
    np.testing.assert_allclose(out3["energy"].numpy(), E, rtol=1e-12)
    np.testing.assert_allclose(out3["forces"].numpy(), F_[full], atol=1e-11)
    # forces = -dE/dpos by central differences
    h = 1e-5
    for (i, a) in [(0, 0), (5, 2), (len(pos) - 1, 1)]:
        p1, p2 = pos.copy(), pos.copy()
        p1[i, a] += h
        p2[i, a] -= h
        e1 = _run(model, p1, z, ptr, ei, batch)["energy"].sum().item()
        e2 = _run(model, p2, z, ptr, ei, batch)["energy"].sum().item()
        assert abs(-(e1 - e2) / (2 * h) - F_[i, a]) < 1e-6 * max(1.0, abs(F_[i, a]))


def test_virial_is_minus_dE_dstrain_by_finite_differences():
    """Virial branch of the oracle (nn/basic.py:93-107,162-199): virial = -dE/dstrain, symmetric, checked by central
    differences of the energy under a homogeneous deformation of positions AND cell of a small periodic box; for an
    isolated molecule it equals sum_i r_i (x) F_i (symmetrised)."""
    model, _ = _small_model()
    rng = np.random.default_rng(4)
    # two-graph periodic batch, 14 atoms each, cell ~ 7 A (images needed at rc = 5)
    pos, cell, z = [], [], []
    for _ in range(2):
        c = np.eye(3) * 7.0 + rng.normal(0, 0.4, size=(3, 3))
        pos.append(rng.uniform(0, 1, size=(14, 3)) @ c)
        cell.append(c)
        z.append(rng.choice([1, 6, 8], size=14))
    pos, cell, z = np.concatenate(pos), np.stack(cell), np.concatenate(z)
    ptr = np.array([0, 14, 28])
    batch = np.repeat(np.arange(2), 14)
    ei, co = orc.radius_graph_pbc_oracle(pos, np.array([14, 14]), [True, True, True], cell, 5.0)

    def run(p, c, **kw):
        return model({"pos": torch.tensor(p), "atomic_numbers": torch.tensor(z.astype(np.int64)), "edge_index": torch.tensor(ei),
                      "batch": torch.tensor(batch), "ptr": torch.tensor(ptr), "cell": torch.tensor(c),
                      "cell_offsets": torch.tensor(co.astype(np.float64))}, **kw)

    out = run(pos, cell, compute_forces=True, compute_virial=True)
    V = out["virial"].numpy()
    assert V.shape == (2, 3, 3)
    np.testing.assert_allclose(V, V.transpose(0, 2, 1), atol=1e-12)
    np.testing.assert_allclose(out["forces"].numpy(), run(pos, cell)["forces"].numpy(), atol=1e-13)
    np.testing.assert_allclose(run(pos, cell, compute_forces=False, compute_virial=True)["virial"].numpy(), V, atol=1e-13)
    h = 1e-5
    for g, (a, b) in [(0, (0, 0)), (0, (0, 2)), (1, (1, 2)), (1, (2, 2))]:
        es = []
        for sgn in (+1, -1):
            eps = np.zeros((3, 3))
            eps[a, b] += 0.5 * sgn * h
            eps[b, a] += 0.5 * sgn * h          # symmetric strain of graph g only
            p2, c2 = pos.copy(), cell.copy()
            sl = slice(ptr[g], ptr[g + 1])
            p2[sl] = pos[sl] + pos[sl] @ eps
            c2[g] = cell[g] + cell[g] @ eps
            es.append(run(p2, c2, compute_forces=False)["energy"][g].item())
        fd = -(es[0] - es[1]) / (2 * h)
        want = V[g, a, b]   # d/dh of E under eps = h (e_ab + e_ba) / 2 is sym(dE/dstrain)_ab = -V_ab
        assert abs(fd - want) < 2e-6 * max(1.0, abs(want)), (g, a, b, fd, want)
    # isolated molecules: virial = sym(sum_i r_i (x) F_i)
    pos_m, z_m, ptr_m, ei_m, batch_m = _inputs()
    om = model({"pos": torch.tensor(pos_m), "atomic_numbers": torch.tensor(z_m.astype(np.int64)), "edge_index": torch.tensor(ei_m),
                "batch": torch.tensor(batch_m), "ptr": torch.tensor(ptr_m)}, compute_forces=True, compute_virial=True)
    Fm = om["forces"].numpy()
    W = np.zeros((len(ptr_m) - 1, 3, 3))
    np.add.at(W, batch_m, pos_m[:, :, None] * Fm[:, None, :])
    np.testing.assert_allclose(om["virial"].numpy(), 0.5 * (W + W.transpose(0, 2, 1)), atol=1e-10)


def test_first_block_reduces_to_edge_term():
    """x^0 = 0 => after the first message, x = sum_e rsh (x) gate_edge only."""
    model, kw = _small_model()
    pos, z, ptr, ei, batch = _inputs(n_mol=1)
    d = orc.compute_edge_data({"pos": torch.tensor(pos), "atomic_numbers": torch.tensor(z.astype(np.int64)),
                               "edge_index": torch.tensor(ei), "batch": torch.tensor(batch), "ptr": torch.tensor(ptr)}, False)
    d = model.embedding(d)
    assert d["node_equivariant"].abs().max() == 0
    s0 = d["node_invariant"].clone()
    d = model.message(0, d)
    p = "mods.message_0."
    sd = model.sd
    s_hat = torch.nn.functional.layer_norm(s0, (32,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    h = torch.nn.functional.linear(torch.nn.functional.silu(torch.nn.functional.linear(s_hat, sd[p + "scalar_mlp.0.weight"], sd[p + "scalar_mlp.0.bias"])),
                                   sd[p + "scalar_mlp.2.weight"], sd[p + "scalar_mlp.2.bias"])
    filt = torch.nn.functional.linear(d["rbf"], sd[p + "rbf_lin.weight"], sd[p + "rbf_lin.bias"]) * d["fcut"]
    C = 56
    # the norm of zeros is zeros except the 0e bias, which the state gate then carries
    xhat0 = orc.equivariant_layer_norm(kw["node_irreps"], torch.zeros(len(pos), 120, dtype=torch.float64),
                                       sd[p + "o3norm.affine_weight"], sd[p + "o3norm.affine_bias"])
    g = h[ei[1]] * filt
    want = torch.zeros(len(pos), 120, dtype=torch.float64).index_add(
        0, torch.tensor(ei[0]), orc.elementwise_tp(kw["node_irreps"], d["rsh"], g[:, C:2 * C])
        + orc.elementwise_tp(kw["node_irreps"], xhat0[ei[1]], g[:, :C]))
    np.testing.assert_allclose(d["node_equivariant"].numpy(), want.numpy(), atol=1e-12)

"""General Clebsch-Gordan tensor product: Wigner-3j tables, get_feasible_tp, and (-m gpu) the HIP kernel
against the oracle's einsums.  e3nn is absent from /root/reference: parity unpinned, conventions pinned by properties."""
import math

import numpy as np
import pytest
import torch

from oracle import tp_oracle as tpo
from oracle import xpainn_oracle as orc
from xequinet_amd import tp

RNG = np.random.default_rng(5)


def test_wigner_3j_tables_are_invariant_orthonormal_and_known():
    for l1 in range(5):
        for l2 in range(5):
            for l3 in range(abs(l1 - l2), min(4, l1 + l2) + 1):
                C = tp.wigner_3j(l1, l2, l3).numpy()
                assert C.shape == (2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1) and abs(np.linalg.norm(C) - 1) < 1e-12
                w = RNG.normal(size=3)
                D1, D2, D3 = (tpo.wigner_D_real(l, w) for l in (l1, l2, l3))
                assert np.abs(np.einsum("ijk,ai,bj,ck->abc", C, D1, D2, D3) - C).max() < 1e-11
                # sum_ij C_ijk C_ijk' = delta_kk' / (2 l3 + 1): what 'component' normalisation rests on
                np.testing.assert_allclose(np.einsum("ijk,ijm->km", C, C), np.eye(2 * l3 + 1) / (2 * l3 + 1), atol=1e-12)
                # exchange symmetry: C(l2, l1, l3)_jik = (-1)^(l1 + l2 + l3) C(l1, l2, l3)_ijk up to the table's sign convention
                Ct = tp.wigner_3j(l2, l1, l3).numpy().transpose(1, 0, 2)
                assert min(np.abs(Ct - C).max(), np.abs(Ct + C).max()) < 1e-12
    assert np.allclose(np.abs(tp.wigner_3j(1, 1, 0).numpy()[:, :, 0]), np.eye(3) / math.sqrt(3))
    with pytest.raises(NotImplementedError):
        tp.wigner_3j(5, 5, 0)


def test_real_wigner_D_rotates_the_paths_spherical_harmonics():
    """The D matrices the equivariance tests use act on the basis of the hot path's Y_1, Y_2 (o3.SphericalHarmonics as the
    reference calls it, on vec[:, [1, 2, 0]]): Y_l(R r) = D^l(R) Y_l(r) in the ORIGINAL axis order."""
    r = torch.tensor(RNG.normal(size=(7, 3)))
    w = RNG.normal(size=3)
    R = torch.tensor(tpo.rotation_matrix(w))
    Y = orc.spherical_harmonics("1x1o+1x2e", r[:, [1, 2, 0]])
    Yr = orc.spherical_harmonics("1x1o+1x2e", (r @ R.T)[:, [1, 2, 0]])
    np.testing.assert_allclose(Yr[:, :3].numpy(), Y[:, :3].numpy() @ tpo.wigner_D_real(1, w).T, atol=1e-12)
    np.testing.assert_allclose(Yr[:, 3:].numpy(), Y[:, 3:].numpy() @ tpo.wigner_D_real(2, w).T, atol=1e-12)


def test_irreps_and_get_feasible_tp_follow_the_reference_rules():
    a = tp.Irreps("32x0e + 32x1o + 32x2e")
    assert a.dim == 32 * 9 and a.lmax == 2 and repr(a) == "32x0e+32x1o+32x2e"
    assert [repr(i) for i in tp.Irrep("1o") * tp.Irrep("2e")] == ["1o", "2o", "3o"]
    srt, p, inv = tp.Irreps("4x2e+4x0e+4x1e+4x1o").sort()
    assert repr(srt) == "4x0e+4x1o+4x1e+4x2e" and p == [3, 0, 2, 1] and inv == [1, 3, 2, 0]
    # SelfMixTP's product (nn/xe3net.py:118-136): hidden x hidden -> every even/odd l up to 2 lmax, 'uuu'
    hid = tp.Irreps("8x0e+8x1o+8x2e")
    mix = [(8, (0, 1))] + [(8, (l, s)) for l in range(2, 4) for s in (-1, 1)] + [(8, (4, 1))]
    out, ins = tp.get_feasible_tp(hid, hid, tp.Irreps(mix), "uuu")
    assert repr(out) == "8x0e+8x2e+8x2o+8x3o+8x3e+8x4e"
    assert all(i[3] == "uuu" and i[4] for i in ins) and len(ins) == 13
    # every path weight is sqrt(dim_out / number of paths) for 'uuu' (nn/tp.py:88-99)
    for i1, i2, io, mode, train, alpha in ins:
        assert abs(alpha - math.sqrt(out[io][1].dim / len(ins))) < 1e-12
        assert out[io][1] in hid[i1][1] * hid[i2][1]
    out2, ins2 = tp.get_feasible_tp("4x0e+4x1o", "4x0e+4x1o", "6x0e+6x1o+6x2e", "uuw")
    assert repr(out2) == "6x0e+6x1o+6x2e" and {i[2] for i in ins2} == {0, 1, 2}


def _case(mode, has_w, shared, n=37, dtype=torch.float64):
    in1, in2 = tp.Irreps("6x0e+6x1o+6x2e"), tp.Irreps("6x0e+6x1o+6x2e") if mode in ("uuu", "uuw") else tp.Irreps("3x0e+5x1o+2x2e")
    flt = tp.Irreps("6x0e+6x1o+6x1e+6x2e+6x3o") if mode in ("uuw", "uvw") else tp.Irreps("1x0e+1x1o+1x1e+1x2e+1x2o+1x3o+1x4e")
    out, ins = tp.get_feasible_tp(in1, in2, flt, mode, trainable=has_w)
    mod = tp.TensorProduct(in1, in2, out, ins, internal_weights=has_w and shared, shared_weights=shared)
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(n, in1.dim, generator=g, dtype=dtype), torch.randn(n, in2.dim, generator=g, dtype=dtype)
    w = None
    if has_w:
        w = mod.weight.detach().to(dtype) if shared else torch.randn(n, mod.weight_numel, generator=g, dtype=dtype)
    l_of = lambda irr: [(m, ir.l) for m, ir in irr]
    want = tpo.tensor_product(l_of(in1), l_of(in2), l_of(out), ins, tp.wigner_3j, x, y, w, shared_weights=shared)
    return mod, in1, in2, out, ins, x, y, w, want


@pytest.mark.gpu
@pytest.mark.parametrize("mode,has_w,shared", [("uuu", True, True), ("uuu", False, True), ("uuw", True, False), ("uuw", True, True),
                                               ("uvw", True, True), ("uvu", True, True), ("uvv", True, False), ("uvuv", False, True)])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
def test_tensor_product_kernel_matches_oracle(mode, has_w, shared, dtype, tol):
    mod, in1, in2, out, ins, x, y, w, want = _case(mode, has_w, shared)
    mod = mod.to("cuda")
    got = mod(x.to(dtype).cuda(), y.to(dtype).cuda(), None if (w is None or (shared and has_w)) else w.to(dtype).cuda())
    assert got.shape == (x.shape[0], out.dim)
    scale = max(1.0, float(want.abs().max()))
    np.testing.assert_allclose(got.detach().cpu().double().numpy(), want.numpy(), rtol=0, atol=tol * scale)


@pytest.mark.gpu
def test_tensor_product_is_equivariant_and_normalised():
    """D_out(R) TP(x, y) = TP(D_1(R) x, D_2(R) y) for the self-mix product of the reference (parity of an improper map
    included through the irreps' parities), and unit-variance inputs give O(1) outputs per path normalisation."""
    mod, in1, in2, out, ins, x, y, w, _ = _case("uuu", True, True, n=64)
    mod = mod.to("cuda")

    def D(irr, wv, improper):
        blocks = []
        for m, ir in irr:
            d = tpo.wigner_D_real(ir.l, wv) * (ir.p if improper else 1)
            blocks += [d] * m
        return torch.tensor(np.block([[b if i == j else np.zeros((b2.shape[0], b.shape[1])) for j, b in enumerate(blocks)]
                                      for i, b2 in enumerate(blocks)]))

    for improper in (False, True):
        wv = RNG.normal(size=3)
        D1, D2, Do = D(in1, wv, improper), D(in2, wv, improper), D(out, wv, improper)
        a = mod((x @ D1.T).cuda(), (y @ D2.T).cuda()).detach().cpu()
        b = mod(x.cuda(), y.cuda()).detach().cpu() @ Do.T
        np.testing.assert_allclose(a.numpy(), b.numpy(), atol=1e-10)
    z = mod(x.cuda(), y.cuda()).detach().cpu()
    assert 0.2 < float(z.pow(2).mean()) < 5.0
    # l x l -> 0e is the scaled dot product, 1o x 1o -> 1e the cross product (in the (y, z, x) component order)
    one = tp.TensorProduct("1x1o", "1x1o", "1x0e+1x1e", [(0, 0, 0, "uuu", False, 1.0), (0, 0, 1, "uuu", False, 1.0)]).to("cuda")
    u, v = torch.tensor(RNG.normal(size=(5, 3))), torch.tensor(RNG.normal(size=(5, 3)))
    r = one(u[:, [1, 2, 0]].cuda(), v[:, [1, 2, 0]].cuda()).cpu().numpy()
    np.testing.assert_allclose(np.abs(r[:, 0]), np.abs((u * v).sum(1).numpy()) / math.sqrt(3), atol=1e-12)
    cross = np.cross(u.numpy(), v.numpy())[:, [1, 2, 0]]
    assert min(np.abs(r[:, 1:] - cross / math.sqrt(2)).max(), np.abs(r[:, 1:] + cross / math.sqrt(2)).max()) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("mode,has_w,shared", [("uuu", True, True), ("uuu", False, True), ("uuw", True, False), ("uuw", True, True),
                                               ("uvw", True, True), ("uvw", True, False), ("uvu", True, True), ("uvu", True, False),
                                               ("uvv", True, False), ("uvv", True, True)])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 3e-5)])
def test_tensor_product_gradients_match_oracle(mode, has_w, shared, dtype, tol):
    """dL/dx1, dL/dx2 (the same kernel on a permuted 3j table, tp._REVERSE) and dL/dW against the oracle's einsums
    differentiated by autograd, L = <out, G> with a random G."""
    mod, in1, in2, out, ins, x, y, w, _ = _case(mode, has_w, shared)
    g = torch.Generator().manual_seed(11)
    G = torch.randn(x.shape[0], out.dim, generator=g, dtype=torch.float64)
    # oracle (fp64, CPU)
    xr, yr = x.clone().requires_grad_(), y.clone().requires_grad_()
    wr = None if w is None else w.clone().requires_grad_()
    l_of = lambda irr: [(m, ir.l) for m, ir in irr]
    ref = tpo.tensor_product(l_of(in1), l_of(in2), l_of(out), ins, tp.wigner_3j, xr, yr, wr, shared_weights=shared)
    want = torch.autograd.grad((ref * G).sum(), [t for t in (xr, yr, wr) if t is not None])
    # HIP
    mod = mod.to("cuda")
    xg, yg = x.to(dtype).cuda().requires_grad_(), y.to(dtype).cuda().requires_grad_()
    if w is None:
        wg, leaves = None, [xg, yg]
    elif shared:                      # internal weights: the module's parameter
        with torch.no_grad():
            mod.weight.copy_(w)
        mod = mod.to(dtype)
        wg, leaves = None, [xg, yg, mod.weight]
    else:
        wg = w.to(dtype).cuda().requires_grad_()
        leaves = [xg, yg, wg]
    res = mod(xg, yg, wg)
    got = torch.autograd.grad((res * G.to(dtype).cuda()).sum(), leaves)
    assert len(got) == len(want)
    for name, a, b in zip(("grad_x1", "grad_x2", "grad_w"), got, want):
        scale = max(1.0, float(b.abs().max()))
        np.testing.assert_allclose(a.detach().cpu().double().numpy(), b.numpy(), rtol=0, atol=tol * scale, err_msg=name)


@pytest.mark.gpu
def test_tensor_product_outer_mode_is_forward_only():
    mod, in1, in2, out, ins, x, y, w, want = _case("uvuv", False, True)
    mod = mod.to("cuda")
    xg = x.cuda().requires_grad_()
    res = mod(xg, y.cuda())
    np.testing.assert_allclose(res.detach().cpu().numpy(), want.numpy(), atol=1e-12)
    with pytest.raises(NotImplementedError):
        res.sum().backward()


def _reference_products(ch):
    """The two general products the reference builds: SelfMixTP's 'uuu' product with shared internal weights (nn/xe3net.py:118-146)
    and the Cartesian-tensor head's 'uuw' product with one weight set per sample (nn/output.py:411-421)."""
    hid = tp.Irreps([(ch, (l, (-1) ** l)) for l in range(3)])
    mix = [(ch, (0, 1))] + [(ch, (l, s)) for l in range(2, 4) for s in (-1, 1)] + [(ch, (4, 1))]
    out, ins = tp.get_feasible_tp(hid, hid, tp.Irreps(mix), "uuu")
    selfmix = tp.TensorProduct(hid, hid, out, ins, internal_weights=True, shared_weights=True)
    out2, ins2 = tp.get_feasible_tp(out, out, tp.Irreps("1x0e+1x2e"), "uuw")
    head = tp.TensorProduct(out, out, out2, ins2, internal_weights=False, shared_weights=False)
    return selfmix, head


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["selfmix", "head", "uvv", "uvu", "uvw", "uuw"])
def test_all_paths_in_one_launch_equal_one_launch_per_path(which):
    """xeq_tensor_product (round 4: every path of a pass in one launch -- a thread per (node, u) for the tied modes, an element per
    thread otherwise -- and a native weight gradient) against the round-2 form (xeq_tensor_product_path per instruction, weight
    gradients through einsum) on the same module: output and all three gradients, including output blocks that several paths with
    DIFFERENT multiplicities end in (the reverse pass of 'uvv')."""
    torch.manual_seed(3)
    n = 257
    if which in ("selfmix", "head"):
        mod = _reference_products(70)[0 if which == "selfmix" else 1]     # 70 channels: more than one lane round per node
        mod = mod.double().cuda()
        x = torch.randn(n, mod.irreps_in1.dim, dtype=torch.float64, device="cuda", requires_grad=True)
        y = torch.randn(n, mod.irreps_in2.dim, dtype=torch.float64, device="cuda", requires_grad=True)
        w = None if mod.internal_weights else torch.randn(n, mod.weight_numel, dtype=torch.float64, device="cuda", requires_grad=True)
    else:
        mod, in1, in2, out, ins, x, y, w, _ = _case(which, True, which != "uuw", n=n)
        mod = mod.cuda()
        x, y = x.cuda().requires_grad_(), y.cuda().requires_grad_()
        w = None if mod.internal_weights else w.cuda().requires_grad_()
    got = {}
    for fused in (False, True):
        mod.fused = fused
        o = mod(x, y) if w is None else mod(x, y, w)
        g = torch.cos(torch.arange(o.numel(), device="cuda", dtype=torch.float64)).reshape(o.shape)
        got[fused] = [o.detach()] + list(torch.autograd.grad(o, [x, y] + ([w] if w is not None else [mod.weight]), g))
    assert mod._fused_table("fwd", x) is not None
    for name, a, b in zip(("out", "grad_x1", "grad_x2", "grad_w"), got[True], got[False]):
        assert float((a - b).abs().max()) <= 1e-12 * max(1.0, float(b.abs().max())), name


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["selfmix", "head"])
def test_zero_rows_give_zero_weight_gradients(which):
    """n == 0 (an empty shard): the fused weight-gradient entry returns without writing; shared weights get exact zeros,
    per-sample weights an empty gradient of their own shape (uninitialised memory / a shape error before)."""
    mod = _reference_products(70)[0 if which == "selfmix" else 1].double().cuda()
    x = torch.zeros(0, mod.irreps_in1.dim, dtype=torch.float64, device="cuda", requires_grad=True)
    y = torch.zeros(0, mod.irreps_in2.dim, dtype=torch.float64, device="cuda", requires_grad=True)
    w = None if mod.internal_weights else torch.zeros(0, mod.weight_numel, dtype=torch.float64, device="cuda", requires_grad=True)
    o = mod(x, y) if w is None else mod(x, y, w)
    gw, = torch.autograd.grad(o.sum(), [w] if w is not None else [mod.weight], allow_unused=True)
    want = w if w is not None else mod.weight
    assert gw is None or (gw.shape == want.shape and float(gw.abs().sum()) == 0.0)


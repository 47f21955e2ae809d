"""GPU parity of the matrix-core two-layer MLPs (xeq_mlp2_fwd / _bwd, csrc/xeq_mlp.hip) through the C ABI against an fp64
restatement of nn.Sequential(Linear, SiLU, Linear) (nn/xpainn.py:103-107, :177-181) and of its input gradient.

Tolerance: exact-f32 products with f32 accumulation over k <= 576 terms of O(1) operands: |err| <= 2e-5 absolute (the library
GEMM chain it replaces sits at 3e-6..6e-6 on the same inputs)."""
import numpy as np
import pytest
import torch

from xequinet_amd import lib
from xequinet_amd.lib import call, ptr, stream
from xequinet_amd.nn import fused

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-5


def _seq(k1, n2, seed):
    torch.manual_seed(seed)
    seq = torch.nn.Sequential(torch.nn.Linear(k1, 128), torch.nn.SiLU(), torch.nn.Linear(128, n2)).to(DEV)
    with torch.no_grad():
        seq[0].bias.normal_()
        seq[2].bias.normal_()
    return seq.requires_grad_(False)


def _ref(seq, x, g):
    xd = x.double().requires_grad_(True)
    w1, b1, w2, b2 = (t.double() for t in (seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias))
    pre = xd @ w1.t() + b1
    y = torch.nn.functional.silu(pre) @ w2.t() + b2
    (gx,) = torch.autograd.grad(y, xd, g.double())
    return pre.detach(), y.detach(), gx


# the four stacks of the default model (message / update, forward k1 -> n2), ragged row counts around the 32-row tile, one row
@pytest.mark.parametrize("n", [1, 31, 32, 33, 777, 4100])
@pytest.mark.parametrize("k1,n2", [(128, 576), (352, 480), (128, 32), (32, 128)])
def test_mlp2_matches_fp64(n, k1, n2):
    seq = _seq(k1, n2, seed=n + k1)
    x = torch.randn(n, k1, device=DEV)
    g = torch.randn(n, n2, device=DEV)
    pre, y = fused._mlp_fwd(seq, x)
    gx = fused._mlp_bwd(seq, g, pre)
    assert getattr(seq, "_xeq_mlp_pack", None) is not None, "the matrix-core path did not run"
    pre_r, y_r, gx_r = _ref(seq, x, g)
    for name, got, ref in (("pre", pre, pre_r), ("y", y, y_r), ("grad_x", gx, gx_r)):
        err = (got.double() - ref).abs().max().item()
        assert err <= TOL, f"{name}: {err:.2e}"


def test_mlp2_strided_rows_and_row_count_invariance():
    """The update MLP reads [shat | v] rows out of a wider buffer; a row's result does not depend on how many rows ride along."""
    seq = _seq(352, 480, seed=7)
    wide = torch.randn(300, 400, device=DEV)          # row stride 400, first 352 columns are the input
    x = wide[:, :352]
    pre, y = fused._mlp_fwd(seq, x)
    pre_c, y_c = fused._mlp_fwd(seq, x.contiguous())
    assert torch.equal(pre, pre_c) and torch.equal(y, y_c)
    pre_h, y_h = fused._mlp_fwd(seq, x[:45].contiguous())
    assert torch.equal(pre[:45], pre_h) and torch.equal(y[:45], y_h)


@pytest.mark.parametrize("k1,n2", [(128, 576), (352, 480)])
def test_mlp2_64_row_form_gives_the_same_rows(k1, n2):
    """From 16 384 rows on a workgroup owns 64 rows (two row tiles share every weight fragment); a row's sums do not depend on the
    form that computed them: bit-equal to the 32-row form, forward and reverse, ragged tail included."""
    seq = _seq(k1, n2, seed=11)
    n = 16384 + 37
    x = torch.randn(n, k1, device=DEV)
    g = torch.randn(n, n2, device=DEV)
    pre, y = fused._mlp_fwd(seq, x)
    gx = fused._mlp_bwd(seq, g, pre)
    for lo, hi in ((0, 300), (16100, n)):       # the same rows through the 32-row form
        pre_s, y_s = fused._mlp_fwd(seq, x[lo:hi].contiguous())
        gx_s = fused._mlp_bwd(seq, g[lo:hi].contiguous(), pre_s)
        assert torch.equal(pre[lo:hi], pre_s) and torch.equal(y[lo:hi], y_s) and torch.equal(gx[lo:hi], gx_s)
    pre_r, y_r, gx_r = _ref(seq, x[:2000], g[:2000])
    assert (y[:2000].double() - y_r).abs().max().item() <= TOL and (gx[:2000].double() - gx_r).abs().max().item() <= TOL


def test_mlp2_repacks_when_a_weight_changes():
    seq = _seq(128, 576, seed=3)
    x = torch.randn(64, 128, device=DEV)
    _, y0 = fused._mlp_fwd(seq, x)
    with torch.no_grad():
        seq[2].bias.add_(1.0)
    _, y1 = fused._mlp_fwd(seq, x)
    np.testing.assert_allclose((y1 - y0).cpu().numpy(), 1.0, atol=1e-5)


def test_mlp2_falls_back_for_other_stacks():
    """Another activation or f64 goes to the library GEMMs (same numbers as eager torch)."""
    seq = torch.nn.Sequential(torch.nn.Linear(128, 128), torch.nn.Tanh(), torch.nn.Linear(128, 576)).to(DEV).requires_grad_(False)
    x = torch.randn(50, 128, device=DEV)
    pre, y = fused._mlp_fwd(seq, x)
    assert getattr(seq, "_xeq_mlp_pack", None) is None
    torch.testing.assert_close(y, seq(x))
    seq64 = _seq(128, 576, seed=1).double()
    x64 = x.double()
    torch.testing.assert_close(fused._mlp_fwd(seq64, x64)[1], seq64(x64))


def test_mlp2_rejects_bad_sizes():
    w = torch.zeros(1024, device=DEV)
    with pytest.raises(RuntimeError, match="xeq_mlp_pack"):
        call("xeq_mlp_pack", ptr(w), None, 30, 8, 0, ptr(w), stream())
    with pytest.raises(RuntimeError, match="xeq_mlp2_fwd"):
        call("xeq_mlp2_fwd", ptr(w), 6, 4, 8, ptr(w), ptr(w), 32, ptr(w), ptr(w), 32, stream())   # row stride not 16-byte
    assert lib.load().xeq_mlp2_supported(lib.XEQ_F32, 128, 64, 576) == 0
    assert lib.load().xeq_mlp2_supported(lib.XEQ_F32, 8, 128, 576) == 0
    assert lib.load().xeq_mlp2_supported(lib.XEQ_F64, 128, 128, 576) == 0


# ---- first half of XPainnUpdate.forward in one launch (xeq_update_uv_fwd, csrc/xeq_update.hip) -------------------------
@pytest.mark.parametrize("n", [1, 31, 33, 1000])
@pytest.mark.parametrize("fuse_norm_bwd", [True, False])
@pytest.mark.parametrize("irreps,layer_norm", [("128x0e + 64x1o + 32x2e", True), ("128x0e + 64x1o + 32x2e", False), ("64x0e + 32x1o", True)])
def test_update_block_fused_front_matches_the_kernel_chain(monkeypatch, n, irreps, layer_norm, fuse_norm_bwd):
    """norm -> U, V -> v, p on the matrix cores against the separate kernels + library GEMMs it replaces (the reference's
    op sequence nn/xpainn.py:208-222), through the whole block: forward outputs and the input gradients (the reverse pass
    runs xeq_update_uv_bwd in its fused and its split form)."""
    from xequinet_amd.nn.xpainn import XPainnUpdate

    torch.manual_seed(n)
    if not fuse_norm_bwd:   # the reverse kernel's split form (what batches beyond one tile per CU take)
        monkeypatch.setattr(fused, "UV_BWD_FUSE_NORM_MAX_NODES", 0)
    node_dim = 128 if irreps.startswith("128") else 64
    blk = XPainnUpdate(node_dim=node_dim, node_irreps=irreps, layer_norm=layer_norm).to(DEV).eval().requires_grad_(False)
    with torch.no_grad():
        for prm in blk.parameters():
            if prm.dim() == 1:
                prm.add_(0.3 * torch.randn_like(prm))
    D = blk.node_irreps.dim
    s0, x0 = torch.randn(n, node_dim, device=DEV), torch.randn(n, D, device=DEV)
    gs, gx = torch.randn(n, node_dim, device=DEV), torch.randn(n, D, device=DEV)

    def run(block, dtype):
        s, x = s0.to(dtype).requires_grad_(True), x0.to(dtype).requires_grad_(True)
        with torch.enable_grad():
            so, xo = fused.UpdateBlock.apply(s, x, block)
            g = torch.autograd.grad([so, xo], [s, x], [gs.to(dtype), gx.to(dtype)])
        return so.detach().double(), xo.detach().double(), g[0].double(), g[1].double()

    got = run(blk, torch.float32)
    if node_dim == 128:
        assert getattr(blk, "_uv_frag", None) is not None, "the fused front did not run"
    monkeypatch.setattr(fused, "_packed_uv_frag", lambda module: None)
    chain = run(blk, torch.float32)
    import copy
    exact = run(copy.deepcopy(blk).double(), torch.float64)      # the kernel chain in f64
    # both f32 paths sit at rounding distance from the f64 result; the fused one may not be further than the chain it replaces
    for name, a_, c_, e_ in zip(("s_out", "x_out", "grad_s", "grad_x"), got, chain, exact):
        err, err_chain = (a_ - e_).abs().max().item(), (c_ - e_).abs().max().item()
        scale = max(1.0, e_.abs().max().item())
        assert err <= max(3.0 * err_chain, 2e-5 * scale), f"{name}: fused {err:.2e}, chain {err_chain:.2e} (scale {scale:.1f})"


@pytest.mark.parametrize("n", [33, 1000, 9000])
def test_update_block_without_a_consumer_of_x_out(monkeypatch, n):
    """Last block of a force evaluation: the head reads s_out only, dL/dx_out arrives as None and the reverse kernels skip its
    terms -- same gradients as with an explicit zero (fused and split reverse forms, and the kernel chain)."""
    from xequinet_amd.nn.xpainn import XPainnUpdate

    torch.manual_seed(n)
    blk = XPainnUpdate().to(DEV).eval().requires_grad_(False)
    D = blk.node_irreps.dim
    s0, x0 = torch.randn(n, 128, device=DEV), torch.randn(n, D, device=DEV)
    gs = torch.randn(n, 128, device=DEV)

    def run(explicit_zero):
        s, x = s0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
        with torch.enable_grad():
            so, xo = fused.UpdateBlock.apply(s, x, blk)
            if explicit_zero:
                return torch.autograd.grad([so, xo], [s, x], [gs, torch.zeros_like(xo)])
            return torch.autograd.grad([so], [s, x], [gs])

    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    monkeypatch.setattr(fused, "_packed_uv_frag", lambda module: None)   # the elementwise kernels + library GEMMs
    c = run(False)   # sanity against the chain (rounding-level agreement is pinned against f64 in the test above)
    for got, ref in zip(a, c):
        assert (got - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())


# ---- single linear layers (csrc/xeq_linear.hip): dot_lin both ways, the embedding with its table gather, the energy head ----
@pytest.mark.parametrize("n", [1, 31, 33, 1000])
@pytest.mark.parametrize("k_in,n_out,bias", [(224, 128, False), (128, 224, False), (56, 128, True), (128, 64, True), (64, 128, False)])
def test_linear_matches_fp64_and_does_not_depend_on_the_batch(n, k_in, n_out, bias):
    torch.manual_seed(n + k_in)
    lin = torch.nn.Linear(k_in, n_out, bias=bias).to(DEV).requires_grad_(False)
    if bias:
        lin.bias.normal_()
    x = torch.randn(n, k_in, device=DEV)
    y = fused.linear_module_fwd(lin, x)
    ref = x.double() @ lin.weight.double().t() + (lin.bias.double() if bias else 0.0)
    assert y.shape == (n, n_out) and float((y.double() - ref).abs().max()) <= TOL
    # the same rows inside a larger batch, at another offset: the same bits
    big = torch.cat([torch.randn(45, k_in, device=DEV), x, torch.randn(19, k_in, device=DEV)])
    assert torch.equal(fused.linear_module_fwd(lin, big)[45:45 + n], y)
    # input gradient through the transposed pack of the same weight
    g = torch.randn(n, n_out, device=DEV)
    gx = fused.linear_module_bwd(lin, g)
    assert float((gx.double() - g.double() @ lin.weight.double()).abs().max()) <= TOL


def test_embedding_lookup_and_energy_head_kernels():
    from xequinet_amd.nn import resolve_model

    torch.manual_seed(3)
    model = resolve_model("xpainn").eval().requires_grad_(False).to(DEV)
    emb = model.mods["embedding"]
    z = torch.randint(0, 87, (999,), device=DEV, dtype=torch.int32)
    got = emb._embed(z)
    lin = emb.embedding[1]
    with torch.no_grad():
        lin.bias.normal_()
    got = emb._embed(z)
    ref = emb.embedding[0].embed_ten[z.long()].double() @ lin.weight.double().t() + lin.bias.double()
    assert float((got.double() - ref).abs().max()) <= TOL * max(1.0, float(ref.abs().max()))
    assert torch.equal(emb._embed(z[100:200]), got[100:200])
    # energy head: Linear(128, 64) - SiLU - Linear(64, 1), values and input gradient
    head = model.mods["output_energy"].out_mlp
    s = torch.randn(777, 128, device=DEV, requires_grad=True)
    assert fused.EnergyHead.supported(head, s)
    e = fused.EnergyHead.apply(s, head)
    g = torch.randn(777, device=DEV)
    (gs,) = torch.autograd.grad(e, s, g)
    sd = s.detach().double().requires_grad_(True)
    w1, b1, w2, b2 = (t.double() for t in (head[0].weight, head[0].bias, head[2].weight, head[2].bias))
    er = (torch.nn.functional.silu(sd @ w1.t() + b1) @ w2.t() + b2).reshape(-1)
    (gr,) = torch.autograd.grad(er, sd, g.double())
    assert float((e.double() - er).abs().max()) <= TOL and float((gs.double() - gr).abs().max()) <= TOL


@pytest.mark.parametrize("n,m,k", [(18609, 576, 128), (18609, 128, 352), (1001, 128, 56), (777, 64, 128), (5, 32, 32), (0, 64, 64),
                                   (55827, 64, 128)])
def test_weight_gradient_kernel_matches_fp64(n, m, k):
    """xeq_wgrad (training pass): a^T b over the node rows against the fp64 product, strided row views, repeatable bit for bit."""
    g = torch.Generator().manual_seed(n + m + k)
    a_full = torch.randn((n, m + 8), generator=g).to("cuda")
    b_full = torch.randn((n, k + 4), generator=g).to("cuda")
    a, b = a_full[:, 4 : 4 + m], b_full[:, :k]       # row strides m + 8 / k + 4
    got = fused._wgrad(a, b)
    want = torch.mm(a.double().t(), b.double())
    scale = max(1.0, want.abs().max().item())
    assert got.shape == (m, k) and (got.double() - want).abs().max().item() <= 2e-6 * scale * max(1.0, n ** 0.5 / 30)
    assert torch.equal(got, fused._wgrad(a, b))
    assert torch.equal(fused._wgrad(a.double(), b.double()), want)     # f64: the library product
    got_w, got_b = fused._wgrad(a, b, with_bias=True)                  # bias gradient (column sums of a) out of the same launch
    assert torch.equal(got_w, got)
    want_b = a.double().sum(0)
    assert got_b.shape == (m,) and (got_b.double() - want_b).abs().max().item() <= 2e-6 * max(1.0, want_b.abs().max().item()) * max(1.0, n ** 0.5 / 30)

"""The twice-differentiable kernels of the training pass, one function at a time: ops.DiffMessage (message aggregation on
the sb-family kernels, second order by multilinearity) and nn/training_ops (norms, Invariant / channel dot, update products: second
order from the same kernel bodies on dual numbers), against finite differences of themselves (``gradcheck`` / ``gradgradcheck``, fp64)
and against the tensor form of the same arithmetic (nn/training.py with the kernel forms switched off) on a whole model.

Tolerances: fp64; gradcheck's defaults (atol 1e-5 on difference quotients of step 1e-6); kernel form vs tensor form of the model:
1e-9 of the largest entry of each gradient (two op orders of the same arithmetic)."""
import numpy as np
import pytest
import torch
from torch.autograd import gradcheck, gradgradcheck

from xequinet_amd import keys, ops, train
from xequinet_amd.nn import training as tr
from xequinet_amd.nn import training_ops as tops

from .test_gpu_training import DEV, SMALL, _batch, _model, _targets

pytestmark = pytest.mark.gpu
F, MUL = 6, (5, 3, 2)
C, D = sum(MUL), MUL[0] + 3 * MUL[1] + 5 * MUL[2]


def _rand(*shape, seed=0, grad=True):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64).to(DEV).requires_grad_(grad)


@pytest.mark.parametrize("layout", [0, 1])
def test_norm_kernels_first_and_second_order(layout):
    n = 5
    s, x = _rand(n, F, seed=1), _rand(n, D, seed=2)
    ln_w, ln_b, eq_w, eq_b = _rand(F, seed=3), _rand(F, seed=4), _rand(C, seed=5), _rand(MUL[0], seed=6)
    meta = (F, MUL, 1e-5, 1e-5, layout)
    fn = lambda *a: tops.NormFn.apply(*a, meta)
    assert gradcheck(fn, (s, x, ln_w, ln_b, eq_w, eq_b))
    # second order w.r.t. the activations and the weights (cotangents of first-order parameter gradients are refused by design)
    frozen = lambda s_, x_, lw_, ew_: tops.NormFn.apply(s_, x_, lw_, ln_b.detach(), ew_, eq_b.detach(), meta)
    sub = lambda s_, x_, lw_, ew_: torch.autograd.grad([o.sum() * 0 + (o * o).sum() for o in frozen(s_, x_, lw_, ew_)], (s_, x_), create_graph=True)
    assert gradcheck(sub, (s, x, ln_w, eq_w))


def test_norm_kernel_equals_the_tensor_form():
    from xequinet_amd import o3
    from xequinet_amd.nn.o3layer import EquivariantLayerNorm

    n = 7
    irreps = o3.Irreps(f"{MUL[0]}x0e + {MUL[1]}x1o + {MUL[2]}x2e")
    eq = EquivariantLayerNorm(irreps).to(torch.float64).to(DEV)
    ln = torch.nn.LayerNorm(F).to(torch.float64).to(DEV)
    with torch.no_grad():
        for p in (*eq.parameters(), *ln.parameters()):
            p.add_(0.3 * torch.randn(p.shape, dtype=p.dtype, device=p.device))
    s, x = _rand(n, F, seed=1, grad=False), _rand(n, D, seed=2, grad=False)
    got_s, got_x = tops.NormFn.apply(s, x, ln.weight, ln.bias, eq.affine_weight, eq.affine_bias, (F, MUL, ln.eps, eq.eps, 0))
    assert torch.allclose(got_s, ln(s), rtol=0, atol=1e-12)
    assert torch.allclose(got_x, tr.equivariant_layer_norm(eq, x), rtol=0, atol=1e-12)
    bt = tops.NormFn.apply(s, x, ln.weight, ln.bias, eq.affine_weight, eq.affine_bias, (F, MUL, ln.eps, eq.eps, 1))[1]
    base = 0
    for l in range(3):       # BT: per l a row-major [N (2l+1), mul_l] matrix
        m, k = MUL[l], 2 * l + 1
        blk = bt[n * base : n * base + n * k * m].view(n, k, m)
        off = sum(MUL[j] * (2 * j + 1) for j in range(l))
        assert torch.equal(blk, got_x[:, off : off + m * k].view(n, m, k).transpose(1, 2))
        base += k * m


def _uv(n, seed):
    return [_rand(n * (2 * l + 1), 2 * MUL[l], seed=seed + l) for l in range(3)]


def test_uv_kernels_first_and_second_order():
    n = 4
    uv = _uv(n, 10)
    fn = lambda a, b, c: tops.UvFn.apply(a, b, c, (MUL, 1e-3))
    assert gradcheck(fn, tuple(uv))
    assert gradgradcheck(fn, tuple(uv))
    out = fn(*uv)
    for l in range(3):       # values against the definition
        k, m = 2 * l + 1, MUL[l]
        blk = uv[l].view(n, k, 2 * m)
        U, V = blk[:, :, :m], blk[:, :, m:]
        ch = sum(MUL[:l])
        assert torch.allclose(out[:, ch : ch + m], torch.sqrt((V * V).sum(1) + 1e-6) - 1e-3, rtol=0, atol=1e-13)
        assert torch.allclose(out[:, C + ch : C + ch + m], (U * V).sum(1), rtol=0, atol=1e-13)


def test_update_product_kernels_first_and_second_order():
    n = 4
    uv = _uv(n, 20)
    a, inner = _rand(n, C + 2 * F, seed=30), _rand(n, F, seed=31)
    fn = lambda p, q, r, a_, i_: tops.UpdateOutFn.apply(p, q, r, a_, i_, (F, MUL))
    assert gradcheck(fn, (*uv, a, inner))
    assert gradgradcheck(fn, (*uv, a, inner))
    d_s, d_x = fn(*uv, a, inner)
    assert torch.allclose(d_s, a[:, C : C + F] * inner + a[:, C + F :], rtol=0, atol=1e-13)
    for l in range(3):
        k, m = 2 * l + 1, MUL[l]
        U = uv[l].view(n, k, 2 * m)[:, :, :m]
        ch, off = sum(MUL[:l]), sum(MUL[j] * (2 * j + 1) for j in range(l))
        want = (U * a[:, None, ch : ch + m]).transpose(1, 2).reshape(n, m * k)
        assert torch.allclose(d_x[:, off : off + m * k], want, rtol=0, atol=1e-13)


@pytest.mark.parametrize("sorted_edges", [True, False])
def test_message_kernels_first_and_second_order(sorted_edges):
    n, B = 5, 6
    H = F + 2 * C
    pairs = [(i, j) for i in range(n) for j in range(n) if i != j and (i + 2 * j) % 3 != 0]     # not symmetric, a node without edges allowed
    ei = torch.tensor(pairs, dtype=torch.int64).t().contiguous()
    if not sorted_edges:
        ei = ei[:, torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(0))].contiguous()
    graph = ops.EdgeGraph(ei.to(DEV), n)
    E = ei.shape[1]
    bp = (B + 3) & ~3
    h, xhat = _rand(n, H, seed=1), _rand(n, D, seed=2)
    rec = _rand(E, bp + 12, seed=3)
    w, b = _rand(H, B, seed=4), _rand(H, seed=5)
    cfg = (B, F, MUL)
    assert ops.diff_message_supported(h, graph, cfg)
    fn = lambda *a: ops.DiffMessage.apply(*a, graph, cfg)
    assert gradcheck(fn, (h, xhat, rec, w, b))
    # the force evaluation differentiates w.r.t. the geometry (records) and the activations; its derivative w.r.t. everything
    first = lambda h_, x_, r_, w_, b_: torch.autograd.grad([(o * o).sum() for o in fn(h_, x_, r_, w_, b_)], (h_, x_, r_), create_graph=True)
    assert gradcheck(first, (h, xhat, rec, w, b))
    # values against the definition (nn/xpainn.py:140-159 without the residual)
    ds, dx = fn(h, xhat, rec, w, b)
    filt = rec[:, :B] @ w.t() + rec[:, bp : bp + 1] * b
    g = h[graph.edge_index[1]] * filt
    want_s = torch.zeros(n, F, dtype=torch.float64, device=DEV).index_add(0, graph.edge_index[0], g[:, 2 * C :])
    assert torch.allclose(ds, want_s, rtol=0, atol=1e-12)
    y = [torch.ones(E, 1, dtype=torch.float64, device=DEV), rec[:, bp + 1 : bp + 4], rec[:, bp + 4 : bp + 9]]
    parts, ch, off = [], 0, 0
    for l in range(3):
        k, m = 2 * l + 1, MUL[l]
        xj = xhat[graph.edge_index[1]][:, off : off + m * k].view(E, m, k)
        parts.append((xj * g[:, ch : ch + m, None] + y[l][:, None, :] * g[:, C + ch : C + ch + m, None]).reshape(E, m * k))
        ch, off = ch + m, off + m * k
    want_x = torch.zeros(n, D, dtype=torch.float64, device=DEV).index_add(0, graph.edge_index[0], torch.cat(parts, 1))
    assert torch.allclose(dx, want_x, rtol=0, atol=1e-12)


@pytest.mark.parametrize("periodic,linear", [(False, False), (True, False), (False, True)])
def test_force_loss_gradients_kernel_form_equals_tensor_form(periodic, linear, monkeypatch):
    """The whole model, energy + forces (+ virial) in the loss: every parameter gradient from the kernel forms of the training pass
    equals the one from the tensor form of the same arithmetic; the kernel forms did run."""
    weights = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 10.0}
    if periodic:
        weights[keys.VIRIAL] = 0.5
    host, dev = _batch(6, 5, torch.float64, periodic)
    tgt = {k: v.to(DEV) for k, v in _targets(host, 7, periodic).items()}
    grads = {}
    calls = []
    real = tops.call
    monkeypatch.setattr(tops, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    for native in (True, False):
        monkeypatch.setattr(tr, "NATIVE_MESSAGE", native)
        monkeypatch.setattr(tr, "NATIVE_NODE", native)
        monkeypatch.setattr(tr, "NATIVE_LINEAR", native and linear)     # linear layers as xeq::linear (LinearFn / WGradFn nodes) or torch.nn
        model = _model(torch.float64, **SMALL).train()
        ops.KERNEL_TIMER.reset(True)
        loss, _ = train.weighted_loss(model(dict(dev), True, periodic), tgt, weights)
        loss.backward()
        launched = set(ops.KERNEL_TIMER.summary())
        ops.KERNEL_TIMER.reset(False)
        assert ("xeq_message_bwd_sbq" in launched) == native and ("xeq_message_q_wgrad" in launched) == native
        grads[native] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert {"xeq_train_norm", "xeq_train_uv", "xeq_train_out"} <= set(calls)
    assert grads[True].keys() == grads[False].keys() and len(grads[True]) >= 50
    for name, g in grads[False].items():
        err = (grads[True][name] - g).abs().max().item()
        assert err <= 1e-9 * max(1e-6, g.abs().max().item()), f"{name}: {err:.2e} of {g.abs().max().item():.2e}"


def test_linear_function_pair_is_closed_under_differentiation():
    """LinearFn / WGradFn: values and two orders of derivatives equal the library's, in fp64 (library products inside) and in fp32 (the
    row reductions on xeq_wgrad; tolerance 2e-5 of the largest entry: different summation orders of 300 fp32 rows)."""
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
        x = _rand(300, 64, seed=1).detach().to(dtype).requires_grad_()
        W = _rand(96, 64, seed=2).detach().to(dtype).requires_grad_()
        b = _rand(96, seed=3).detach().to(dtype).requires_grad_()
        t = _rand(300, 96, seed=4).detach().to(dtype)
        tops.linear(x, W, b)                                   # loads the operator library
        forms = [lambda: tops.LinearFn.apply(x, W, b), lambda: torch.ops.xeq.linear(x, W, b), lambda: torch.nn.functional.linear(x, W, b)]
        results = []
        for fn in forms:
            y = torch.tanh(fn())
            (gx,) = torch.autograd.grad((y * t).sum(), x, create_graph=True)
            second = torch.autograd.grad((gx * gx).sum(), (x, W, b), allow_unused=True)
            results.append((y, gx, *second))
        for outs in (results[0], results[1]):                  # the Python pair and the C++ pair against the library's layer
            for got, want in zip(outs, results[2]):
                if want is None:
                    assert got is None or float(got.abs().max()) == 0.0
                else:
                    assert float((got - want).detach().abs().max()) <= tol * max(1.0, float(want.detach().abs().max()))


def test_force_loss_kernel_form_without_l2_channels_and_with_an_isolated_atom(monkeypatch):
    """Irreps that stop at l = 1 (no Y_2 rows, an empty uv_2) and an atom without neighbours (a row of the walk without edges): the kernel
    form of the force-loss pass against the tensor form."""
    from oracle import xpainn_oracle as orc
    from xequinet_amd.data import synthetic as syn

    cfg = dict(node_dim=64, node_irreps="64x0e + 32x1o", action_blocks=2, hidden_dim=32)
    pos, z, ptr = syn.synth_qm9_batch(4, seed=3)
    pos = pos.copy()
    pos[0] += 100.0                                              # out of every cutoff sphere
    ei = orc.radius_graph_canonical(pos, ptr, 5.0)
    assert not (ei == 0).any()
    batch = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    dev = {"pos": torch.tensor(pos, dtype=torch.float64, device=DEV), "atomic_numbers": torch.tensor(z.astype(np.int64), device=DEV),
           "edge_index": torch.tensor(ei, device=DEV), "batch": torch.tensor(batch, device=DEV), "ptr": torch.tensor(ptr, device=DEV)}
    g = torch.Generator().manual_seed(2)
    tgt = {keys.TOTAL_ENERGY: torch.randn(4, generator=g, dtype=torch.float64).to(DEV),
           keys.FORCES: torch.randn(len(pos), 3, generator=g, dtype=torch.float64).to(DEV), keys.BATCH_PTR: dev["ptr"]}
    weights = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 5.0}
    grads = {}
    for native in (True, False):
        monkeypatch.setattr(tr, "NATIVE_MESSAGE", native)
        monkeypatch.setattr(tr, "NATIVE_NODE", native)
        model = _model(torch.float64, **cfg).train()
        result = model(dict(dev), True, False)
        loss, _ = train.weighted_loss(result, tgt, weights)
        loss.backward()
        assert float(result[keys.FORCES][0].detach().abs().max()) == 0.0          # the isolated atom feels nothing
        grads[native] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert grads[True].keys() == grads[False].keys() and len(grads[True]) >= 40
    for name, want in grads[False].items():
        err = (grads[True][name] - want).abs().max().item()
        assert err <= 1e-9 * max(1e-6, want.abs().max().item()), f"{name}: {err:.2e} of {want.abs().max().item():.2e}"


@pytest.mark.parametrize("B", [5, 12, 20, 31])
def test_message_kernels_every_basis_width_in_both_precisions(B):
    """The four record widths the sb-family kernels are instantiated for (num_basis <= 8 / 16 / 20 / 32), fp32 against fp64: values,
    first-order gradients and the second-order gradients of a quadratic in the first-order ones (2e-5 of the largest entry)."""
    n = 40
    H = F + 2 * C
    g = torch.Generator().manual_seed(B)
    pairs = [(i, j) for i in range(n) for j in range(n) if i != j and (3 * i + j) % 5 < 2]
    ei = torch.tensor(pairs, dtype=torch.int64).t().contiguous().to(DEV)
    graph = ops.EdgeGraph(ei, n)
    E, bp = ei.shape[1], (B + 3) & ~3
    base = [torch.randn(n, H, generator=g, dtype=torch.float64), torch.randn(n, D, generator=g, dtype=torch.float64),
            torch.randn(E, bp + 12, generator=g, dtype=torch.float64), torch.randn(H, B, generator=g, dtype=torch.float64) / B ** 0.5,
            torch.randn(H, generator=g, dtype=torch.float64)]
    res = {}
    for dt in (torch.float64, torch.float32):
        h, xhat, rec, w, b = (t.to(dt).to(DEV).requires_grad_() for t in base)
        ds, dx = ops.DiffMessage.apply(h, xhat, rec, w, b, graph, (B, F, MUL))
        first = torch.autograd.grad((ds * ds).sum() + (dx * dx).sum(), (h, xhat, rec), create_graph=True)
        second = torch.autograd.grad(sum((t * t).sum() for t in first), (h, xhat, rec, w, b))
        res[dt] = [t.detach().double() for t in (ds, dx, *first, *second)]
    for got, want in zip(res[torch.float32], res[torch.float64]):
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def test_geometry_only_pass_skips_the_filter_gradients_and_nothing_else():
    """``ops.geometry_only_backward``: inside it (the force evaluation's own reverse pass, nn/basic.py:143-159) DiffMessage leaves out
    dL/d[W | b]; every other gradient is the same, and a reverse pass outside it -- also one running while another is inside -- gets all."""
    n, B = 6, 5
    H = F + 2 * C
    pairs = [(i, j) for i in range(n) for j in range(n) if i != j]
    graph = ops.EdgeGraph(torch.tensor(pairs, dtype=torch.int64).t().contiguous().to(DEV), n)
    E, bp = len(pairs), (B + 3) & ~3
    h, xhat, rec, w, b = _rand(n, H, seed=1), _rand(n, D, seed=2), _rand(E, bp + 12, seed=3), _rand(H, B, seed=4), _rand(H, seed=5)
    ds, dx = ops.DiffMessage.apply(h, xhat, rec, w, b, graph, (B, F, MUL))
    e = (ds * ds).sum() + (dx * dx).sum()
    full = torch.autograd.grad(e, (h, xhat, rec, w, b), retain_graph=True)
    with ops.geometry_only_backward(e):
        part = torch.autograd.grad(e, (h, xhat, rec, w, b), retain_graph=True, allow_unused=True)
    assert part[3] is None and part[4] is None
    for got, want in zip(part[:3], full[:3]):
        assert torch.equal(got, want)
    assert not ops._GEOMETRY_ONLY_TASKS                         # the task ids are forgotten on exit
    again = torch.autograd.grad(e, (w, b))
    assert torch.equal(again[0], full[3]) and torch.equal(again[1], full[4])

"""Node-side functions of the twice-differentiable training pass (nn/training.py) on the kernels of csrc/xeq_train_node.hip.

Each function is an ``autograd.Function`` whose reverse pass is a second ``Function`` (``*Grad``), so autograd can differentiate the
force evaluation itself (nn/basic.py:143-159 with ``create_graph=training``).  The ``*Grad`` nodes get their own reverse pass from the
same two kernels run on dual numbers: with u the cotangent of the first-order input gradient, the forward kernel at tangent u is the
derivative w.r.t. the incoming cotangent, the reverse kernel at tangent u the derivative w.r.t. the inputs and the parameters
(include/xeq.h, xeq_train_norm).  Cotangents of first-order PARAMETER gradients (a third order) are refused.
"""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ..lib import call, dtype_code, mul3, ptr, ptr3, require_hip, stream


def _z(like: torch.Tensor, t):
    return torch.zeros_like(like) if t is None else t.contiguous()


def _no_param_cotangent(*ts) -> None:
    if any(t is not None for t in ts):
        raise NotImplementedError("training_ops: derivatives of first-order parameter gradients are not implemented")


# ---- LayerNorm + EquivariantLayerNorm -------------------------------------------------------------------------------------------------
def _norm_call(reverse, s, s_t, x, x_t, ln_w, ln_b, eq_w, eq_b, g_s, g_x, meta):
    F, mul, eps_ln, eps_eq, layout = meta
    N, D, C = s.shape[0], x.shape[1], sum(mul)
    o_s = torch.empty_like(s)
    if reverse:
        o_x = torch.empty_like(x)
        rows = torch.empty((N, 2 * F + C + mul[0]), dtype=s.dtype, device=s.device)
    else:
        o_x = torch.empty((N, D) if layout == 0 else (N * D,), dtype=s.dtype, device=s.device)
        rows = None
    call("xeq_train_norm", dtype_code(s), int(reverse), N, ptr(s), ptr(s_t), ptr(x), ptr(x_t), ptr(ln_w), ptr(ln_b), ptr(eq_w), ptr(eq_b),
         ptr(g_s), ptr(g_x), F, mul3(mul), float(eps_ln), float(eps_eq), int(layout), ptr(o_s), ptr(o_x), ptr(rows), stream())
    return o_s, o_x, rows


def _norm_rows(rows, F, C, m0):
    t = rows.sum(0)
    return t[:F], t[F : 2 * F], t[2 * F : 2 * F + C], t[2 * F + C :]


class NormFn(Function):
    """(s, x, ln.weight, ln.bias, eqln.weight, eqln.bias) -> (LayerNorm(s), EquivariantLayerNorm(x)); meta = (F, mul, eps_ln, eps_eq,
    layout of xhat: 0 the e3nn row [N, D], 1 BT as one flat buffer [N D])."""

    @staticmethod
    def forward(ctx, s, x, ln_w, ln_b, eq_w, eq_b, meta):
        require_hip(s, x, ln_w, ln_b, eq_w, eq_b)
        s, x, ln_w, ln_b, eq_w, eq_b = (t.contiguous() for t in (s, x, ln_w, ln_b, eq_w, eq_b))
        o_s, o_x, _ = _norm_call(0, s, None, x, None, ln_w, ln_b, eq_w, eq_b, None, None, meta)
        ctx.save_for_backward(s, x, ln_w, ln_b, eq_w, eq_b)
        ctx.meta = meta
        return o_s, o_x

    @staticmethod
    def backward(ctx, g_s, g_x):
        s, x, ln_w, ln_b, eq_w, eq_b = ctx.saved_tensors
        g_s = _z(s, g_s)
        g_x = torch.zeros(x.numel() if ctx.meta[4] else x.shape, dtype=x.dtype, device=x.device) if g_x is None else g_x.contiguous()
        from ..ops import _in_geometry_only_task

        # the force evaluation's own reverse pass (ops.geometry_only_backward) wants no parameter gradient: the sum over the nodes is left out
        return (*NormGrad.apply(s, x, ln_w, ln_b, eq_w, eq_b, g_s, g_x, ctx.meta, not _in_geometry_only_task()), None)


class NormGrad(Function):
    @staticmethod
    def forward(ctx, s, x, ln_w, ln_b, eq_w, eq_b, g_s, g_x, meta, want_params=True):
        F, mul = meta[0], meta[1]
        d_s, d_x, rows = _norm_call(1, s, None, x, None, ln_w, ln_b, eq_w, eq_b, g_s, g_x, meta)
        ctx.save_for_backward(s, x, ln_w, ln_b, eq_w, eq_b, g_s, g_x)
        ctx.meta = meta
        ctx.set_materialize_grads(False)
        if not want_params:
            return d_s, d_x, None, None, None, None
        return (d_s, d_x, *_norm_rows(rows, F, sum(mul), mul[0]))

    @staticmethod
    @once_differentiable
    def backward(ctx, u_s, u_x, u_lw, u_lb, u_ew, u_eb):
        _no_param_cotangent(u_lw, u_lb, u_ew, u_eb)
        s, x, ln_w, ln_b, eq_w, eq_b, g_s, g_x = ctx.saved_tensors
        if u_s is None and u_x is None:
            return (None,) * 10
        F, mul = ctx.meta[0], ctx.meta[1]
        u_s, u_x = _z(s, u_s), _z(x, u_x)
        d_gs, d_gx, _ = _norm_call(0, s, u_s, x, u_x, ln_w, ln_b, eq_w, eq_b, None, None, ctx.meta)
        d_s, d_x, rows = _norm_call(1, s, u_s, x, u_x, ln_w, ln_b, eq_w, eq_b, g_s, g_x, ctx.meta)
        d_lw, _, d_ew, _ = _norm_rows(rows, F, sum(mul), mul[0])
        return d_s, d_x, d_lw, None, d_ew, None, d_gs, d_gx, None, None


# ---- Invariant(V) and the channel-wise U . V ------------------------------------------------------------------------------------------
def _uv_call(reverse, uv, uv_t, g, meta):
    mul, eps = meta
    C = sum(mul)
    N = max((uv[l].shape[0] // (2 * l + 1) for l in range(3) if mul[l] > 0), default=0)
    ref = next(uv[l] for l in range(3) if mul[l] > 0)
    if reverse:
        d_uv = [torch.empty_like(t) for t in uv]
        call("xeq_train_uv", dtype_code(ref), 1, N, ptr3(uv), ptr3(uv_t), ptr(g), mul3(mul), float(eps), None, ptr3(d_uv), stream())
        return d_uv
    out = torch.empty((N, 2 * C), dtype=ref.dtype, device=ref.device)
    call("xeq_train_uv", dtype_code(ref), 0, N, ptr3(uv), ptr3(uv_t), None, mul3(mul), float(eps), ptr(out), None, stream())
    return out


class UvFn(Function):
    """uv_l [N (2l+1), 2 mul_l] (U | V, BT rows), l = 0, 1, 2 (an empty tensor for an absent l) -> [N, 2C] = [Invariant(V) | sum_m U V];
    meta = (mul, eps)."""

    @staticmethod
    def forward(ctx, uv0, uv1, uv2, meta):
        uv = [t.contiguous() for t in (uv0, uv1, uv2)]
        require_hip(*uv)
        ctx.save_for_backward(*uv)
        ctx.meta = meta
        return _uv_call(0, uv, None, None, meta)

    @staticmethod
    def backward(ctx, g):
        return (*UvGrad.apply(*ctx.saved_tensors, g.contiguous(), ctx.meta), None)


class UvGrad(Function):
    @staticmethod
    def forward(ctx, uv0, uv1, uv2, g, meta):
        ctx.save_for_backward(uv0, uv1, uv2, g)
        ctx.meta = meta
        ctx.set_materialize_grads(False)
        return tuple(_uv_call(1, [uv0, uv1, uv2], None, g, meta))

    @staticmethod
    @once_differentiable
    def backward(ctx, u0, u1, u2):
        uv0, uv1, uv2, g = ctx.saved_tensors
        if u0 is None and u1 is None and u2 is None:
            return (None,) * 5
        uv = [uv0, uv1, uv2]
        u = [_z(a, b) for a, b in zip(uv, (u0, u1, u2))]
        d_g = _uv_call(0, uv, u, None, ctx.meta)
        d_uv = _uv_call(1, uv, u, g, ctx.meta)
        return (*d_uv, d_g, None)


# ---- the update block's products -------------------------------------------------------------------------------------------------------
def _out_call(reverse, uv, uv_t, a, a_t, inner, inner_t, g_s, g_x, meta):
    F, mul = meta
    N, C, D = a.shape[0], sum(mul), mul[0] + 3 * mul[1] + 5 * mul[2]
    if reverse:
        d_uv = [torch.empty_like(t) for t in uv]
        d_a, d_in = torch.empty_like(a), torch.empty_like(inner)
        call("xeq_train_out", dtype_code(a), 1, N, ptr3(uv), ptr3(uv_t), ptr(a), ptr(a_t), ptr(inner), ptr(inner_t), ptr(g_s), ptr(g_x), F,
             mul3(mul), ptr(d_a), ptr(d_in), ptr3(d_uv), stream())
        return d_uv, d_a, d_in
    o_s = torch.empty((N, F), dtype=a.dtype, device=a.device)
    o_x = torch.empty((N, D), dtype=a.dtype, device=a.device)
    call("xeq_train_out", dtype_code(a), 0, N, ptr3(uv), ptr3(uv_t), ptr(a), ptr(a_t), ptr(inner), ptr(inner_t), None, None, F, mul3(mul),
         ptr(o_s), ptr(o_x), None, stream())
    return o_s, o_x


class UpdateOutFn(Function):
    """(uv_0, uv_1, uv_2, a [N, C + 2F] = [a_vv | a_sv | a_ss], inner [N, F]) -> (a_sv inner + a_ss [N, F], U a_vv [N, D] e3nn rows);
    meta = (F, mul)."""

    @staticmethod
    def forward(ctx, uv0, uv1, uv2, a, inner, meta):
        ts = [t.contiguous() for t in (uv0, uv1, uv2, a, inner)]
        require_hip(*ts)
        ctx.save_for_backward(*ts)
        ctx.meta = meta
        return _out_call(0, ts[:3], None, ts[3], None, ts[4], None, None, None, meta)

    @staticmethod
    def backward(ctx, g_s, g_x):
        uv0, uv1, uv2, a, inner = ctx.saved_tensors
        F, mul = ctx.meta
        D = mul[0] + 3 * mul[1] + 5 * mul[2]
        g_s = a.new_zeros((a.shape[0], F)) if g_s is None else g_s.contiguous()
        g_x = a.new_zeros((a.shape[0], D)) if g_x is None else g_x.contiguous()
        return (*UpdateOutGrad.apply(uv0, uv1, uv2, a, inner, g_s, g_x, ctx.meta), None)


class UpdateOutGrad(Function):
    @staticmethod
    def forward(ctx, uv0, uv1, uv2, a, inner, g_s, g_x, meta):
        ctx.save_for_backward(uv0, uv1, uv2, a, inner, g_s, g_x)
        ctx.meta = meta
        ctx.set_materialize_grads(False)
        d_uv, d_a, d_in = _out_call(1, [uv0, uv1, uv2], None, a, None, inner, None, g_s, g_x, meta)
        return (*d_uv, d_a, d_in)

    @staticmethod
    @once_differentiable
    def backward(ctx, u0, u1, u2, u_a, u_in):
        uv0, uv1, uv2, a, inner, g_s, g_x = ctx.saved_tensors
        if all(t is None for t in (u0, u1, u2, u_a, u_in)):
            return (None,) * 8
        uv = [uv0, uv1, uv2]
        u = [_z(p, q) for p, q in zip(uv, (u0, u1, u2))]
        u_a, u_in = _z(a, u_a), _z(inner, u_in)
        d_gs, d_gx = _out_call(0, uv, u, a, u_a, inner, u_in, None, None, ctx.meta)
        d_uv, d_a, d_in = _out_call(1, uv, u, a, u_a, inner, u_in, g_s, g_x, ctx.meta)
        return (*d_uv, d_a, d_in, d_gs, d_gx, None)


# ---- linear layers: the weight-gradient products on the row-chunk kernel ----------------------------------------------------------------
class LinearFn(Function):
    """y = x W^T (+ b) on rows (nn.Linear, and the o3.Linear blocks on BT rows).  Its reverse pass is written with ``LinearFn`` and
    ``WGradFn`` themselves (dL/dx = g W, dL/dW = g^T x), so every order of derivative stays on these two products -- and every
    reduction over the N rows, which the library's GEMMs run at a tenth of their rate for these shapes ([576 x N] x [N x 128] ...), goes
    to ``xeq_wgrad`` (fp32; nn/fused.py ``_wgrad``).  The row-parallel products stay library GEMMs."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return torch.addmm(b, x, W.t()) if b is not None else x @ W.t()

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        d_x = LinearFn.apply(g, W.t(), None) if ctx.needs_input_grad[0] else None
        d_W = WGradFn.apply(g, x) if ctx.needs_input_grad[1] else None
        d_b = g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return d_x, d_W, d_b


class WGradFn(Function):
    """a^T b over the rows: a [N, M], b [N, K] -> [M, K]."""

    @staticmethod
    def forward(ctx, a, b):
        from .fused import _wgrad

        ctx.save_for_backward(a, b)
        return _wgrad(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, U):
        a, b = ctx.saved_tensors
        d_a = LinearFn.apply(b, U, None) if ctx.needs_input_grad[0] else None          # b U^T
        d_b = LinearFn.apply(a, U.t(), None) if ctx.needs_input_grad[1] else None      # a U
        return d_a, d_b


_cpp_linear = None


def linear(x: torch.Tensor, W: torch.Tensor, b) -> torch.Tensor:
    """x W^T (+ b) through the LinearFn / WGradFn pair: the C++ nodes of csrc/xeq_torch.cpp (``xeq::linear``: the same two functions as
    ``torch::autograd::Function``s -- a C++ node costs the host a fifth of a Python one, and a host-launched training step is decided
    there) when the operator library is built, the Python pair above otherwise."""
    global _cpp_linear
    if _cpp_linear is None:
        try:
            from ..interface.scripted import load_torch_library

            load_torch_library()
            _cpp_linear = torch.ops.xeq.linear
        except (ImportError, OSError, AttributeError, RuntimeError):
            _cpp_linear = False
    if _cpp_linear:
        return _cpp_linear(x, W, b)
    return LinearFn.apply(x, W, b)


def mlp(seq: torch.nn.Module, x: torch.Tensor) -> torch.Tensor:
    """An nn.Sequential of nn.Linear and activations (or one nn.Linear) with the linear layers as ``linear``."""
    mods = [seq] if isinstance(seq, torch.nn.Linear) else list(seq)
    for m in mods:
        x = linear(x, m.weight, m.bias) if isinstance(m, torch.nn.Linear) else m(x)
    return x

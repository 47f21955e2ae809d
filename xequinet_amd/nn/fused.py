"""Block-level fused paths of XPainnMessage / XPainnUpdate.

One ``torch.autograd.Function`` per block: every elementwise stage is a hand-written HIP
kernel (``xeq_node.hip``), the two-layer scalar MLPs and their input gradients are one matrix-core
launch each (``xeq_mlp.hip``), the o3.Linear contractions are plain library GEMMs on contiguous
views of the internal BT layout, and the reverse pass is explicit (no autograd graph of
small ops).  Semantics are those of nn/xpainn.py:128-161 and :206-231 of the reference;
gradients are provided w.r.t. the node features and the edge vectors (force evaluation,
nn/basic.py:143-159) and, when a block is given its parameters as trailing inputs (the native
training pass of an energy loss, nn/model.py), w.r.t. those parameters as well: the radial
filter's gradients come from ``xeq_message_param_grad`` (the only one that is not a contraction
over saved node tensors), every other weight gradient is a library GEMM / column sum over node
tensors the forward pass keeps anyway.  A loss on forces or virials needs the derivative of
the reverse pass itself: that training pass takes the differentiable form of the blocks in
nn/training.py instead.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
import torch.nn.functional as F_
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import lib, ops
from ..lib import call, dtype_code, mul3, ptr, stream

def _bt_blocks(buf: torch.Tensor, n: int, mul, width: int):
    """Views of the BT buffer as plain row-major matrices [n (2l+1), width * mul_l] per non-empty block."""
    out = []
    base = 0
    for l, m in enumerate(mul):
        d = 2 * l + 1
        if m > 0:
            out.append((l, m, buf[n * base * width : n * (base + d * m) * width].view(n * d, width * m)))
        base += d * m
    return out


def _norm_fwd(s, x, norm, o3norm, node_dim, mul, shat_out=None, ld=None):
    n, D = x.shape
    do_norm = int(not isinstance(norm, torch.nn.Identity))
    if shat_out is None:
        shat_out = torch.empty((n, node_dim), dtype=s.dtype, device=s.device)
        ld = node_dim
    xhat = torch.empty(n * D, dtype=x.dtype, device=x.device)
    stats = torch.empty((n, 4), dtype=s.dtype, device=s.device)
    lw, lb, ew, eb = (norm.weight, norm.bias, o3norm.affine_weight, o3norm.affine_bias) if do_norm else (None,) * 4
    call("xeq_norm_fwd", dtype_code(s), ptr(s), ptr(x), ptr(lw), ptr(lb), ptr(ew), ptr(eb), n, node_dim, mul3(mul), do_norm,
         ptr(shat_out), ld, ptr(xhat), ptr(stats), stream())
    return shat_out, xhat, stats, do_norm


def _norm_bwd(s, x, norm, o3norm, stats, do_norm, node_dim, mul, g_shat, ld, g_xhat, res_s, res_x):
    n, D = x.shape
    g_s, g_x = torch.empty_like(s), torch.empty_like(x)
    lw, ew = (norm.weight, o3norm.affine_weight) if do_norm else (None, None)
    call("xeq_norm_bwd", dtype_code(s), ptr(s), ptr(x), ptr(lw), ptr(ew), ptr(stats), n, node_dim, mul3(mul), do_norm,
         ptr(g_shat), ld, ptr(g_xhat), ptr(res_s), ptr(res_x), ptr(g_s), ptr(g_x), stream())
    return g_s, g_x


def _silu_bwd(g, pre, act):
    if isinstance(act, torch.nn.SiLU):
        return torch.ops.aten.silu_backward(g, pre)
    with torch.enable_grad():  # other activations of resolve_activation: let autograd differentiate the scalar map
        p = pre.detach().requires_grad_()
        (out,) = torch.autograd.grad(act(p), p, g)
    return out


def _mlp_packs(seq: torch.nn.Sequential):
    """Matrix-core fragment-order copies of a Linear-SiLU-Linear stack for ``xeq_mlp2_fwd / _bwd`` (forward pair with the
    biases folded in, transposed pair for the input gradients), cached on the module and refreshed when a weight changes.
    None when the kernels do not take the stack (other activation, f64, widths): the caller then runs the library GEMMs."""
    lin1, act, lin2 = seq[0], seq[1], seq[2]
    w1, w2 = lin1.weight, lin2.weight
    if not (isinstance(act, torch.nn.SiLU) and w1.dtype == torch.float32 and lin1.bias is not None and lin2.bias is not None
            and lib.load().xeq_mlp2_supported(lib.XEQ_F32, w1.shape[1], w1.shape[0], w2.shape[0])):
        return None
    key = (w1._version, w1.data_ptr(), lin1.bias._version, lin1.bias.data_ptr(), w2._version, w2.data_ptr(), lin2.bias._version,
           lin2.bias.data_ptr(), lib.pack_epoch())
    cache = getattr(seq, "_xeq_mlp_pack", None)
    if cache is not None and cache[0] == key:
        return cache[1]

    def pack(w, bias, n_out, k_in, transposed):
        out = torch.empty(lib.load().xeq_mlp_packed_floats(n_out, k_in), dtype=torch.float32, device=w.device)
        call("xeq_mlp_pack", ptr(w), ptr(bias), n_out, k_in, int(transposed), ptr(out), stream())
        return out

    with torch.no_grad():
        h, k1 = w1.shape
        n2 = w2.shape[0]
        w1c, w2c = w1.detach().contiguous(), w2.detach().contiguous()
        packs = (pack(w1c, lin1.bias.detach(), h, k1, False), pack(w2c, lin2.bias.detach(), n2, h, False),
                 pack(w2c, None, h, n2, True), pack(w1c, None, k1, h, True))     # reverse: W2 as [k = n2][n = h], W1 as [k = h][n = k1]
    seq._xeq_mlp_pack = (key, packs)
    return packs


def _mlp_fwd(seq, x):
    """(pre, y) of Linear-SiLU-Linear on rows of x ([n, k1], row stride = x.stride(0))."""
    lin1, act, lin2 = seq[0], seq[1], seq[2]
    packs = _mlp_packs(seq) if x.is_cuda and x.stride(1) == 1 and x.stride(0) % 4 == 0 else None
    if packs is None:
        pre = torch.addmm(lin1.bias, x, lin1.weight.t())
        return pre, torch.addmm(lin2.bias, act(pre), lin2.weight.t())
    n, k1 = x.shape
    n2 = lin2.weight.shape[0]
    pre = torch.empty((n, lin1.weight.shape[0]), dtype=x.dtype, device=x.device)
    y = torch.empty((n, n2), dtype=x.dtype, device=x.device)
    call("xeq_mlp2_fwd", ptr(x), x.stride(0), n, k1, ptr(packs[0]), ptr(packs[1]), n2, ptr(pre), ptr(y), n2, stream())
    return pre, y


def _mlp_bwd(seq, g_y, pre):
    """dL/dx of Linear-SiLU-Linear from dL/dy and the saved pre-activation."""
    lin1, act, lin2 = seq[0], seq[1], seq[2]
    packs = _mlp_packs(seq) if g_y.is_cuda else None
    if packs is None:
        return torch.mm(_silu_bwd(torch.mm(g_y, lin2.weight), pre, act), lin1.weight)
    g_y = g_y.contiguous()
    n, n2 = g_y.shape
    k1 = lin1.weight.shape[1]
    g_x = torch.empty((n, k1), dtype=g_y.dtype, device=g_y.device)
    call("xeq_mlp2_bwd", ptr(g_y), n2, n, n2, ptr(packs[2]), ptr(pre), ptr(packs[3]), k1, ptr(g_x), k1, stream())
    return g_x


# ---- parameter gradients (native training pass of an energy loss) ------------------------------------------------------------
def message_params(module, rbf) -> list:
    """The parameters of an XPainnMessage in the order MessageBlock takes them as trailing inputs (and returns their gradients)."""
    ps = []
    if not isinstance(module.norm, torch.nn.Identity):
        ps += [module.norm.weight, module.norm.bias, module.o3norm.affine_weight, module.o3norm.affine_bias]
    m = module.scalar_mlp
    ps += [m[0].weight, m[0].bias, m[2].weight, m[2].bias, module.rbf_lin.weight, module.rbf_lin.bias]
    ps += [p for p in rbf.params() if p is not None]
    return ps


def update_params(module) -> list:
    ps = []
    if not isinstance(module.norm, torch.nn.Identity):
        ps += [module.norm.weight, module.norm.bias, module.o3norm.affine_weight, module.o3norm.affine_bias]
    m = module.update_mlp
    ps += [module.update_U.weight, module.update_U.bias, module.update_V.weight, module.update_V.bias,
           m[0].weight, m[0].bias, m[2].weight, m[2].bias, module.dot_lin.weight]
    return ps


def _wgrad(a: torch.Tensor, b: torch.Tensor, with_bias: bool = False):
    """a^T b over the rows (a [n, M], b [n, K] -> [M, K]): the weight gradient of a linear layer from dL/dy rows and input rows;
    ``with_bias``: -> (a^T b, column sums of a [M]) = (dL/dW, dL/dbias) out of the same launch and the same sum over the parts.
    f32 on the GPU: ``xeq_wgrad`` (row chunks on the matrix cores, parts summed in a fixed order); else the library product."""
    if not (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.stride(1) == 1 and b.stride(1) == 1):
        d_w = torch.mm(a.t(), b)
        return (d_w, a.sum(0)) if with_bias else d_w
    n, M = a.shape
    K = b.shape[1]
    chunks = int(lib.load().xeq_wgrad_chunks(n, M, K))
    parts = torch.empty((chunks, M * K + (M if with_bias else 0)), dtype=torch.float32, device=a.device)
    call("xeq_wgrad", ptr(a), a.stride(0), ptr(b), b.stride(0), n, M, K, int(with_bias), chunks, ptr(parts), stream())
    tot = parts.sum(0) if chunks > 1 else parts[0]
    d_w = tot[: M * K].view(M, K)
    return (d_w, tot[M * K :]) if with_bias else d_w


def _mlp_param_grads(seq, x_in, pre, g_y):
    """Linear - act - Linear on rows: (dW1, db1, dW2, db2, dL/dx) from the input, the saved pre-activation and dL/dy."""
    lin1, act, lin2 = seq[0], seq[1], seq[2]
    hidden = act(pre)
    d_w2, d_b2 = _wgrad(g_y, hidden, with_bias=True)
    g_pre = _silu_bwd(torch.mm(g_y, lin2.weight), pre, act)
    d_w1, d_b1 = _wgrad(g_pre, x_in, with_bias=True)
    return d_w1, d_b1, d_w2, d_b2, torch.mm(g_pre, lin1.weight)


def _norm_param_grads(s, x, stats, g_shat, g_xhat, node_dim, mul):
    """dL/d(LayerNorm weight, bias) and dL/d(EquivariantLayerNorm affine_weight, affine_bias) from the block inputs, the saved
    statistics (mean, rstd, mean of the 0e block, rsqrt of the mean square norm: xeq_node.hip) and the gradients of the normalised
    features (g_shat [n, F] rows, g_xhat in the BT layout)."""
    n = x.shape[0]
    if s.is_cuda and s.dtype == torch.float32 and g_shat.stride(1) == 1:
        # one launch + one sum over the row chunks (csrc/xeq_train.hip) instead of six reductions and a dozen elementwise launches
        F, C, m0 = node_dim, sum(mul), mul[0]
        chunks = int(lib.load().xeq_norm_param_grad_chunks(n))
        parts = torch.empty((chunks, 2 * F + C + m0), dtype=torch.float32, device=s.device)
        call("xeq_norm_param_grad", ptr(s.contiguous()), ptr(x.contiguous()), ptr(stats), ptr(g_shat), g_shat.stride(0), ptr(g_xhat), n, F,
             mul3(mul), chunks, ptr(parts), stream())
        tot = parts.sum(0)
        return tot[:F], tot[F : 2 * F], tot[2 * F : 2 * F + C], tot[2 * F + C :]
    mean, rstd, mean0, r = stats.unbind(1)
    d_lnw = (g_shat * ((s - mean[:, None]) * rstd[:, None])).sum(0)
    d_lnb = g_shat.sum(0)
    d_eqw, d_eqb, off = [], None, 0
    for l, m in enumerate(mul):
        d = 2 * l + 1
        if m == 0:
            continue
        xb = x[:, off : off + m * d].view(n, m, d)
        gb = g_xhat[n * off : n * (off + m * d)].view(n, d, m)
        if l == 0:
            xb = xb - mean0[:, None, None]
            d_eqb = gb.sum((0, 1))
        d_eqw.append((gb.transpose(1, 2) * xb * r[:, None, None]).sum((0, 2)))   # (an einsum here becomes 128 batched dot products: 6 ms)
        off += m * d
    return d_lnw, d_lnb, torch.cat(d_eqw), d_eqb


def _linear_pack(mod_or_key, weight: torch.Tensor, bias, transposed: bool):
    """Fragment-order copy of one Linear's weight for ``xeq_linear_fwd`` (forward: W as [n_out, k_in] with its bias; ``transposed``:
    the same weight as the input-gradient product, [k_in = n_out of the layer][n_out = k_in of the layer], no bias), cached on the
    module and refreshed when the weight changes.  None when the kernel does not take the layer (f64, widths)."""
    n_out, k_in = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
    if weight.dtype != torch.float32 or not weight.is_cuda or not lib.load().xeq_linear_supported(lib.XEQ_F32, k_in, n_out):
        return None
    key = (weight._version, weight.data_ptr(), None if bias is None else (bias._version, bias.data_ptr()), transposed, lib.pack_epoch())
    name = "_xeq_lin_pack_t" if transposed else "_xeq_lin_pack"
    cache = getattr(mod_or_key, name, None)
    if cache is not None and cache[0] == key:
        return cache[1]
    with torch.no_grad():
        out = torch.empty(lib.load().xeq_mlp_packed_floats(n_out, k_in), dtype=torch.float32, device=weight.device)
        call("xeq_mlp_pack", ptr(weight.detach().contiguous()), ptr(None if (bias is None or transposed) else bias.detach().contiguous()),
             n_out, k_in, int(transposed), ptr(out), stream())
    setattr(mod_or_key, name, (key, out))
    return out


def _linear(x: torch.Tensor, pack: torch.Tensor, k_in: int, n_out: int, has_bias: bool, act: int = 0, row_index=None, want_pre: bool = False):
    """y = act(x W^T + b) on rows of x (row stride x.stride(0)) through ``xeq_linear_fwd``; -> (y, pre or None)."""
    n = x.shape[0] if row_index is None else row_index.shape[0]
    y = torch.empty((n, n_out), dtype=torch.float32, device=x.device)
    pre = torch.empty((n, n_out), dtype=torch.float32, device=x.device) if want_pre else None
    call("xeq_linear_fwd", ptr(x), x.stride(0), n, k_in, ptr(row_index), ptr(pack), n_out, int(has_bias), act, ptr(pre), ptr(y), n_out, stream())
    return y, pre


def linear_module_fwd(lin: torch.nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """A bias-free or biased nn.Linear on the matrix cores when the kernel takes it, else the library GEMM (f64, other widths)."""
    pack = _linear_pack(lin, lin.weight, lin.bias, False) if (x.is_cuda and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0) else None
    if pack is None:
        return torch.nn.functional.linear(x, lin.weight, lin.bias)
    return _linear(x, pack, lin.weight.shape[1], lin.weight.shape[0], lin.bias is not None)[0]


def linear_module_bwd(lin: torch.nn.Linear, g: torch.Tensor) -> torch.Tensor:
    """dL/dx of y = x W^T (+ b): g W, through the transposed pack of the same weight."""
    g = g.contiguous()
    pack = _linear_pack(lin, lin.weight, None, True) if g.is_cuda else None
    if pack is None:
        return torch.mm(g, lin.weight)
    return _linear(g, pack, lin.weight.shape[0], lin.weight.shape[1], False)[0]


def mlp_and_linear_fwd(seq, x, lin: torch.nn.Linear, p: torch.Tensor):
    """(pre, y) of Linear-SiLU-Linear on rows of x and lin(p), two INDEPENDENT products of XPainnUpdate.forward (nn/xpainn.py:219-223),
    through ``xeq_mlp2_and_linear``: one launch for MD-sized systems, the two launches otherwise -- the same bits either way."""
    packs = _mlp_packs(seq) if x.is_cuda and x.stride(1) == 1 and x.stride(0) % 4 == 0 else None
    lpack = _linear_pack(lin, lin.weight, lin.bias, False) if (lin.bias is None and p.is_cuda and p.dim() == 2 and p.stride(1) == 1 and p.stride(0) % 4 == 0) else None
    if packs is None or lpack is None:
        pre, y = _mlp_fwd(seq, x)
        return pre, y, linear_module_fwd(lin, p)
    n, k1 = x.shape
    n2, n_out = seq[2].weight.shape[0], lin.weight.shape[0]
    pre = torch.empty((n, seq[0].weight.shape[0]), dtype=x.dtype, device=x.device)
    y = torch.empty((n, n2), dtype=x.dtype, device=x.device)
    ip = torch.empty((n, n_out), dtype=x.dtype, device=x.device)
    call("xeq_mlp2_and_linear", 0, ptr(x), x.stride(0), n, k1, ptr(packs[0]), ptr(packs[1]), n2, ptr(pre), ptr(y), n2,
         ptr(p), p.stride(0), lin.weight.shape[1], ptr(lpack), n_out, ptr(ip), n_out, stream())
    return pre, y, ip


def mlp_and_linear_bwd(seq, g_y, pre, lin: torch.nn.Linear, g_lin: torch.Tensor):
    """dL/dx of the stack and dL/dp of lin (g_lin W), the reverse twins of ``mlp_and_linear_fwd``."""
    packs = _mlp_packs(seq) if g_y.is_cuda else None
    g_lin = g_lin.contiguous()
    lpack = _linear_pack(lin, lin.weight, None, True) if g_lin.is_cuda else None
    if packs is None or lpack is None:
        return _mlp_bwd(seq, g_y, pre), linear_module_bwd(lin, g_lin)
    g_y = g_y.contiguous()
    n, n2 = g_y.shape
    k1, k_lin, n_lin = seq[0].weight.shape[1], lin.weight.shape[0], lin.weight.shape[1]
    g_x = torch.empty((n, k1), dtype=g_y.dtype, device=g_y.device)
    g_p = torch.empty((n, n_lin), dtype=g_y.dtype, device=g_y.device)
    call("xeq_mlp2_and_linear", 1, ptr(g_y), n2, n, n2, ptr(packs[2]), ptr(packs[3]), k1, ptr(pre), ptr(g_x), k1,
         ptr(g_lin), g_lin.stride(0), k_lin, ptr(lpack), n_lin, ptr(g_p), n_lin, stream())
    return g_x, g_p


class EnergyHead(Function):
    """EnergyOut's MLP on node scalars (nn/output.py:104-118): Linear - SiLU - Linear(., 1) -> atomic energies [n], with the
    explicit reverse pass of a force evaluation (input gradient only).  HIP kernels throughout: a row's sums do not depend on
    the batch it sits in."""

    @staticmethod
    def supported(seq: torch.nn.Sequential, s: torch.Tensor) -> bool:
        lin1, act, lin2 = seq[0], seq[1], seq[2]
        return (isinstance(act, torch.nn.SiLU) and s.is_cuda and s.dtype == torch.float32 and lin1.bias is not None and lin2.bias is not None
                and lin2.weight.shape[0] == 1
                and lin1.weight.shape[0] % 4 == 0 and bool(lib.load().xeq_linear_supported(lib.XEQ_F32, lin1.weight.shape[1], lin1.weight.shape[0]))
                and bool(lib.load().xeq_linear_supported(lib.XEQ_F32, lin1.weight.shape[0], lin1.weight.shape[1])))

    @staticmethod
    def params(seq) -> list:
        return [seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias]

    @staticmethod
    def forward(ctx, s, seq, *params):
        """``params`` = EnergyHead.params(seq) when their gradients are wanted (training pass), else nothing."""
        lin1, lin2 = seq[0], seq[2]
        s = s.contiguous()
        H = lin1.weight.shape[0]
        hidden, pre = _linear(s, _linear_pack(lin1, lin1.weight, lin1.bias, False), lin1.weight.shape[1], H, True, act=1, want_pre=True)
        out = torch.empty(s.shape[0], dtype=s.dtype, device=s.device)
        w2 = lin2.weight.detach().reshape(-1).contiguous()
        call("xeq_head_dot", ptr(hidden), s.shape[0], H, ptr(w2), ptr(lin2.bias), ptr(out), stream())
        ctx.train = len(params) > 0
        ctx.save_for_backward(pre, w2, *((s, hidden) if ctx.train else ()))
        ctx.seq = seq
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        pre, w2 = ctx.saved_tensors[:2]
        lin1 = ctx.seq[0]
        g_out = g_out.contiguous()
        g_pre = torch.empty_like(pre)    # dL/d(pre-activation of the hidden layer)
        call("xeq_head_bwd_hidden", ptr(pre), pre.shape[0], pre.shape[1], ptr(w2), ptr(g_out), ptr(g_pre), stream())
        g_s = linear_module_bwd(lin1, g_pre)
        if not ctx.train:
            return g_s, None
        s, hidden = ctx.saved_tensors[2:]
        return (g_s, None, *_wgrad(g_pre, s, with_bias=True), torch.mv(hidden.t(), g_out).view(1, -1), g_out.sum().view(1))


_SEEDS = {}


def constant_vector(n: int, value: float, dtype, device) -> torch.Tensor:
    """A read-only [n] tensor of one value, cached per (n, value, dtype, device): the seed of a force evaluation's reverse pass
    (nn/basic.py:150: ``grad_outputs = [ones_like(energy)]``) without a fill launch per evaluation.  Not cached while a HIP graph is being
    captured (the tensor would live in that graph's pool); the warm-up steps in front of a capture fill the cache."""
    key = (int(n), float(value), dtype, str(device))
    t = _SEEDS.get(key)
    if t is None:
        t = torch.full((int(n),), float(value), dtype=dtype, device=device)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            if t.is_cuda:
                torch.cuda.current_stream().synchronize()   # once per key: the constant is complete before ANY stream may read it
            if len(_SEEDS) > 256:
                torch.cuda.synchronize()   # (rare: another stream may still read an evicted constant -- steps in flight; round-5 advisor)
                _SEEDS.clear()
            _SEEDS[key] = t
    return t


class EnergyReadout(Function):
    """EnergyOut.forward of an evaluation that wants no parameter gradients (nn/output.py:104-128): (atomic energies [n], total energies
    [G]) in two launches -- the head's MLP with its whole reverse pass saved as a row per node (xeq_head_fwd) and the per-graph sum
    (xeq_segment_sum) -- and its reverse pass in one (xeq_head_bwd: g_s[n] = (g_atomic[n] + g_total[graph(n)]) J[n]).  Round 3's head took
    seven launches per force evaluation."""

    @staticmethod
    def supported(seq: torch.nn.Sequential, s: torch.Tensor) -> bool:
        lin1, act, lin2 = seq[0], seq[1], seq[2]
        return (isinstance(act, torch.nn.SiLU) and s.is_cuda and s.dtype == torch.float32 and lin1.bias is not None and lin2.weight.shape[0] == 1
                and bool(lib.load().xeq_head_supported(lib.XEQ_F32, lin1.weight.shape[1], lin1.weight.shape[0])))

    @staticmethod
    def forward(ctx, s, seq, batch, ptr_):
        lin1, lin2 = seq[0], seq[2]
        s = s.contiguous()
        n, F = s.shape
        H = lin1.weight.shape[0]
        want_bwd = ctx.needs_input_grad[0]
        atomic = torch.empty(n, dtype=s.dtype, device=s.device)
        jac = torch.empty((n, F), dtype=s.dtype, device=s.device) if want_bwd else None
        w2 = lin2.weight.detach().reshape(-1).contiguous()
        call("xeq_head_fwd", ptr(s), s.stride(0), n, F, H, ptr(_linear_pack(lin1, lin1.weight, lin1.bias, False)),
             ptr(_linear_pack(lin1, lin1.weight, None, True) if want_bwd else None), ptr(w2), ptr(lin2.bias), ptr(atomic), ptr(jac), stream())
        ptr64 = ptr_.to(torch.int64).contiguous()
        total = torch.empty(ptr64.numel() - 1, dtype=s.dtype, device=s.device)
        call("xeq_segment_sum", dtype_code(s), ptr(atomic), ptr(ptr64), ptr64.numel() - 1, 1, ptr(total), stream())
        ctx.save_for_backward(*((jac, batch) if want_bwd else ()))
        ctx.set_materialize_grads(False)
        return atomic, total

    @staticmethod
    @once_differentiable
    def backward(ctx, g_atomic, g_total):
        if not ctx.saved_tensors or (g_atomic is None and g_total is None):
            return None, None, None, None
        jac, batch = ctx.saved_tensors
        n, F = jac.shape
        if g_atomic is not None and g_atomic.stride(0) not in (0, 1):
            g_atomic = g_atomic.contiguous()
        if g_total is not None and g_total.stride(0) not in (0, 1):
            g_total = g_total.contiguous()
        g_s = torch.empty_like(jac)
        call("xeq_head_bwd", ptr(jac), n, F, ptr(g_atomic), 0 if g_atomic is None else g_atomic.stride(0), ptr(g_total),
             0 if g_total is None else g_total.stride(0), ptr(batch), ptr(g_s), stream())
        return g_s, None, None, None


class EmbeddingLinear(Function):
    """Table rows of the atomic numbers through Linear(embed_dim, node_dim) (nn/xpainn.py:62) as XEmbedding._embed launches it, with
    the Linear's parameter gradients for a training pass (the table is a buffer, the atomic numbers are integers)."""

    @staticmethod
    def forward(ctx, z32, table, lin, weight, bias):
        out = _linear(table, _linear_pack(lin, weight, bias, False), weight.shape[1], weight.shape[0], bias is not None, row_index=z32)[0]
        ctx.save_for_backward(z32, table)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        z32, table = ctx.saved_tensors
        rows = table.index_select(0, z32.long())
        d_w, d_b = _wgrad(g.contiguous(), rows, with_bias=True)
        return None, None, None, d_w, (d_b if ctx.has_bias else None)


def first_block_front(module, z: torch.Tensor, rows: torch.Tensor, n: int, higher_l_unread: bool = False):
    """(s, h, xhat) of the model's FIRST message block from the element table: behind XEmbedding s = rows[Z] and x = 0, so LayerNorm,
    EquivariantLayerNorm and scalar_mlp (nn/xpainn.py:128-139) are functions of the element alone.  They run on the table rows (once per
    weight version: the same xeq_norm_fwd / xeq_mlp2_fwd launches, which give a row the same bits in any batch) and all three are
    gathered by atomic number in ONE launch (xeq_first_block_front; round 4: three ATen gathers and a fill).  xhat is in BT layout; its
    l > 0 blocks are the equivariant norm of zero: zero -- and are not even written when the caller knows that nobody reads them
    (``higher_l_unread``: the wq message kernels under XEQ_XHAT_HIGHER_L_ZERO drop every term with a factor xhat_{l>0}; 27 MB of zeros
    per QM9-1024 evaluation).  None when the block is not the layout the table form covers."""
    F, mul = module.node_dim, module._mul
    D = sum(m * (2 * l + 1) for l, m in enumerate(mul))
    if rows.dtype != torch.float32 or mul[0] != F or isinstance(module.norm, torch.nn.Identity) or z.dtype not in (torch.int32, torch.int64):
        return None
    mlp = module.scalar_mlp
    key = (rows.data_ptr(), rows._version, module.norm.weight._version, module.norm.bias._version, module.o3norm.affine_weight._version,
           module.o3norm.affine_bias._version, mlp[0].weight._version, mlp[0].bias._version, mlp[2].weight._version, mlp[2].bias._version,
           lib.pack_epoch())
    cache = getattr(module, "_element_front", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            zt = rows.shape[0]
            shat, xhat_t, _, _ = _norm_fwd(rows, torch.zeros((zt, D), dtype=rows.dtype, device=rows.device), module.norm, module.o3norm, F, mul)
            _, h_t = _mlp_fwd(mlp, shat)
            cache = module._element_front = (key, h_t.contiguous(), xhat_t[: zt * F].view(zt, F).contiguous())
    _, h_t, xhat0_t = cache
    H = h_t.shape[1]
    z = z.contiguous()
    s = torch.empty((n, F), dtype=rows.dtype, device=rows.device)
    h = torch.empty((n, H), dtype=rows.dtype, device=rows.device)
    xhat = torch.empty(n * D, dtype=rows.dtype, device=rows.device)
    call("xeq_first_block_front", ptr(z), int(z.dtype == torch.int64), n, rows.shape[0], ptr(rows), ptr(h_t), ptr(xhat0_t), F, H,
         F if higher_l_unread else D, ptr(s), ptr(h), ptr(xhat), stream())
    return s, h, xhat


class MessageBlock(Function):
    """XPainnMessage.forward (nn/xpainn.py:128-161): norms -> scalar_mlp -> fused message kernel."""

    @staticmethod
    def forward(ctx, s, x, vec, module, graph, rbf, cutoff_fn, x_is_zero=False, *params):
        """``params`` = message_params(module, rbf) when their gradients are wanted (training pass), else nothing."""
        lib.require_hip(s, x, vec)
        s, x, vec = s.contiguous(), x.contiguous(), vec.contiguous()
        F, mul = module.node_dim, module._mul
        shat, xhat, stats, do_norm = _norm_fwd(s, x, module.norm, module.o3norm, F, mul)
        pre, h = _mlp_fwd(module.scalar_mlp, shat)
        p0, p1 = rbf.params()
        ctx.p_shapes = (p0.shape, None if p1 is None else p1.shape)
        # xhat in BT layout; behind XEmbedding x is zero, hence xhat is zero on every l > 0 column (include/xeq.h)
        xl = 1 | (lib.XHAT_HIGHER_L_ZERO if x_is_zero else 0)
        cfg = (rbf.kind, cutoff_fn.kind, module.num_basis, float(cutoff_fn.cutoff), F, mul, xl)
        s_out, x_out, saved, impl = ops.message_forward(h, xhat, vec, s, x, module.rbf_lin.weight, module.rbf_lin.bias,
                                                        p0, p1, graph, cfg, want_backward=any(ctx.needs_input_grad))
        ctx.none_mask = [t is None for t in saved]
        ctx.train = len(params) > 0
        ctx.save_for_backward(*[t for t in saved if t is not None], *((shat,) if ctx.train else ()), s, x, stats, pre)
        ctx.module, ctx.do_norm, ctx.graph, ctx.cfg, ctx.impl = module, do_norm, graph, cfg, impl
        ctx.deferral = ops.register_edge_grad(graph, impl, ctx.needs_input_grad[2])
        return s_out, x_out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_s_out, g_x_out):
        module = ctx.module
        F, mul = module.node_dim, module._mul
        t = list(ctx.saved_tensors)
        s, x, stats, pre = t[-4:]
        shat = t[-5] if ctx.train else None
        it = iter(t[: -5 if ctx.train else -4])
        msg_saved = tuple(None if is_none else next(it) for is_none in ctx.none_mask)
        node_grads = ctx.train or ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        g_h, g_xhat, g_vec, g_s_res, g_x_res = ops.message_backward(msg_saved, ctx.graph, ctx.cfg, ctx.impl, g_s_out, g_x_out,
                                                                    node_grads=node_grads, deferral=ctx.deferral)
        if not node_grads:
            # first block of a force evaluation: its node features are the embedding of the atomic numbers and zeros, neither
            # depends on the positions; only dL/dvec leaves this block (no MLP / norm reverse launches)
            return None, None, g_vec, None, None, None, None, None
        if not ctx.train:
            g_shat = _mlp_bwd(module.scalar_mlp, g_h, pre)
            g_s, g_x = _norm_bwd(s, x, module.norm, module.o3norm, stats, ctx.do_norm, F, mul, g_shat, F, g_xhat, g_s_res, g_x_res)
            return g_s, g_x, g_vec, None, None, None, None, None
        # training pass: the same reverse kernels for the input gradients, plus the parameter gradients in message_params' order
        d_wr, d_br, d_p0, d_p1 = ops.message_param_grad(msg_saved, ctx.graph, ctx.cfg, g_s_res, g_x_res)
        d_w1, d_b1, d_w2, d_b2, g_shat = _mlp_param_grads(module.scalar_mlp, shat, pre, g_h)
        g_s, g_x = _norm_bwd(s, x, module.norm, module.o3norm, stats, ctx.do_norm, F, mul, g_shat, F, g_xhat, g_s_res, g_x_res)
        p0, p1 = msg_saved[5], msg_saved[6]
        grads = list(_norm_param_grads(s, x, stats, g_shat, g_xhat, F, mul)) if ctx.do_norm else []
        grads += [d_w1, d_b1, d_w2, d_b2, d_wr, d_br, d_p0.view(ctx.p_shapes[0])]
        if p1 is not None:
            grads.append(d_p1.view(ctx.p_shapes[1]))
        return (g_s, g_x, g_vec, None, None, None, None, None, *grads)


def _packed_uv(module) -> Tuple[list, torch.Tensor]:
    """[W_U | W_V] / sqrt(mul) per l ([mul, 2 mul]) and the 0e bias pair, cached on the module."""
    wu, wv = module.update_U.weight, module.update_V.weight
    key = (wu._version, wv._version, wu.data_ptr(), wv.data_ptr(), module.update_U.bias._version, module.update_V.bias._version, wu.dtype,
           lib.pack_epoch())
    cache = getattr(module, "_uv_pack", None)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2]
    with torch.no_grad():
        packs, off = [], 0
        for mul, l, _, _ in module.node_irreps.blocks():
            Wu = wu[off : off + mul * mul].view(mul, mul)
            Wv = wv[off : off + mul * mul].view(mul, mul)
            packs.append((torch.cat([Wu, Wv], dim=1) / math.sqrt(mul)).contiguous())
            off += mul * mul
        bias = torch.cat([module.update_U.bias, module.update_V.bias]) if module.update_U.bias.numel() else None
    module._uv_pack = (key, packs, bias)
    return packs, bias


def _packed_uv_frag(module):
    """The [W_U | W_V] / sqrt(mul) blocks in matrix-core fragment order for ``xeq_update_uv_fwd`` (one per l, None for an
    absent block; the l = 0 pack carries the bias pair), or None when the kernel does not take the layout."""
    mul = module.node_irreps.mul3()
    wu = module.update_U.weight
    if wu.dtype != torch.float32 or not lib.load().xeq_update_uv_supported(lib.XEQ_F32, module.node_dim, mul3(mul)):
        return None
    packs, bias = _packed_uv(module)                     # refreshed there when a weight changes
    cache = getattr(module, "_uv_frag", None)
    if cache is not None and cache[0] is packs:
        return cache[1], cache[2], cache[3]
    frag, frag_t, it = [None, None, None], [None, None, None], iter(packs)
    for l, m in enumerate(mul):
        if m == 0:
            continue
        W = next(it)                                     # [mul, 2 mul] = [k_in][n_out]
        out = torch.empty(lib.load().xeq_mlp_packed_floats(2 * m, m), dtype=torch.float32, device=W.device)
        call("xeq_mlp_pack", ptr(W), ptr(bias if l == 0 else None), 2 * m, m, 1, ptr(out), stream())
        frag[l] = out
        out_t = torch.empty(lib.load().xeq_mlp_packed_floats(m, 2 * m), dtype=torch.float32, device=W.device)
        call("xeq_mlp_pack", ptr(W), None, m, 2 * m, 0, ptr(out_t), stream())   # reverse: the same rows as [n_out = mul][k_in = 2 mul]
        frag_t[l] = out_t
    module._uv_frag = (packs, frag, bias is not None, frag_t)
    return frag, bias is not None, frag_t


class NodeBlock(Function):
    """XPainnUpdate.forward (nn/xpainn.py:206-231) and, for every block but the last, the front half of the NEXT
    XPainnMessage.forward (nn/xpainn.py:128-139: both norms, scalar_mlp) as ONE launch per direction (csrc/xeq_nodeblock.hip).
    Outputs (s_out, x_out[, h_next, xhat_next]): the message kernel of the next block takes all four (ops.FusedMessage), so its
    reverse hands all four gradients back in one call."""

    @staticmethod
    def forward(ctx, s, x, update, message):
        from . import nodeblock

        lib.require_hip(s, x)
        s, x = s.contiguous(), x.contiguous()
        want_x = message is not None or not getattr(update, "equivariant_output_unused", False)
        o = nodeblock.node_block_fwd(s, x, update, message, want_x=want_x)
        ctx.update, ctx.message = update, message
        ctx.keys = [k for k in ("uv", "stats", "pre", "a", "ip", "s_out", "x_out", "stats2", "pre2") if o.get(k) is not None]
        ctx.save_for_backward(s, x, *[o[k] for k in ctx.keys])
        ctx.set_materialize_grads(False)
        if message is None:
            return o["s_out"], o["x_out"]
        return o["s_out"], o["x_out"], o["h2"], o["xhat2"]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_s_out, g_x_out, g_h=None, g_xhat=None):
        from . import nodeblock

        s, x = ctx.saved_tensors[:2]
        saved = dict(zip(ctx.keys, ctx.saved_tensors[2:]))
        message = ctx.message
        if g_s_out is None:
            g_s_out = torch.zeros_like(s)
        if message is not None:
            if g_h is None:
                g_h = torch.zeros((s.shape[0], message.hidden_dim), dtype=s.dtype, device=s.device)
            if g_xhat is None:
                g_xhat = torch.zeros(x.numel(), dtype=x.dtype, device=x.device)
            if g_x_out is None:
                g_x_out = torch.zeros_like(x)
        g_s, g_x = nodeblock.node_block_bwd(saved, s, x, ctx.update, message, g_s_out, g_x_out, g_h, g_xhat)
        return g_s, g_x, None, None


# xeq_update_uv_bwd can run the norms' reverse inside (113 KB of LDS: one workgroup per CU; 24 against 28 us at 1.5 k nodes) or leave
# it to xeq_norm_bwd (50 KB: three per CU; 92 against 99 us at 18 k nodes).  The two forms reduce a row in different orders, so a
# switch by node count made a node's bits depend on its batch (6e-7 in the forces between a 9 k-node batch and its 2 k-node
# chunks, found once the last library GEMM was gone): since round 3 the split form runs at every size (0 = never fuse).
# The C++ operator (csrc/xeq_torch.cpp) applies the same rule.
UV_BWD_FUSE_NORM_MAX_NODES = 0


class UpdateBlock(Function):
    """XPainnUpdate.forward (nn/xpainn.py:206-231)."""

    @staticmethod
    def forward(ctx, s, x, module, *params):
        """``params`` = update_params(module) when their gradients are wanted (training pass), else nothing."""
        lib.require_hip(s, x)
        s, x = s.contiguous(), x.contiguous()
        n, D = x.shape
        F, mul = module.node_dim, module.node_irreps.mul3()
        C = sum(mul)
        dt, dev = s.dtype, s.device
        cat = torch.empty((n, F + C), dtype=dt, device=dev)                  # [shat | v]  (nn/xpainn.py:215)
        uv = torch.empty(2 * n * D, dtype=dt, device=dev)                    # U|V pair buffer, BT layout
        p = torch.empty((n, C), dtype=dt, device=dev)
        eps = module.invariant.eps
        frag = _packed_uv_frag(module)
        if frag is not None:     # norms -> U, V -> v, p in one matrix-core launch (xeq_update.hip)
            do_norm = int(not isinstance(module.norm, torch.nn.Identity))
            stats = torch.empty((n, 4), dtype=dt, device=dev)
            lw, lb, ew, eb = ((module.norm.weight, module.norm.bias, module.o3norm.affine_weight, module.o3norm.affine_bias)
                              if do_norm else (None,) * 4)
            call("xeq_update_uv_fwd", ptr(s), ptr(x), ptr(lw), ptr(lb), ptr(ew), ptr(eb), n, F, mul3(mul), do_norm, ptr(frag[0][0]),
                 ptr(frag[0][1]), ptr(frag[0][2]), int(frag[1]), float(eps), ptr(cat), F + C, ptr(p), ptr(uv), ptr(stats), stream())
        else:
            _, xhat, stats, do_norm = _norm_fwd(s, x, module.norm, module.o3norm, F, mul, shat_out=cat, ld=F + C)
            packs, bias = _packed_uv(module)
            for (l, m, xb), (_, _, ub), W in zip(_bt_blocks(xhat, n, mul, 1), _bt_blocks(uv, n, mul, 2), packs):
                if l == 0 and bias is not None:
                    torch.addmm(bias, xb, W, out=ub)
                else:
                    torch.mm(xb, W, out=ub)
            call("xeq_uv_reduce_fwd", dtype_code(s), ptr(uv), n, mul3(mul), float(eps), ptr(cat), F + C, F, ptr(p), stream())
        pre, a, ip = mlp_and_linear_fwd(module.update_mlp, cat, module.dot_lin, p)   # a = [a_vv C | a_sv F | a_ss F]; ip = dot_lin(p)
        # (the last block in front of a scalar-only head: nobody reads its equivariant output, nn/model.py marks the module)
        s_out, x_out = torch.empty_like(s), (None if getattr(module, "equivariant_output_unused", False) else torch.empty_like(x))
        call("xeq_update_out_fwd", dtype_code(s), ptr(s), ptr(x), ptr(uv), ptr(a), ptr(ip), n, F, mul3(mul), ptr(s_out),
             ptr(x_out), stream())
        ctx.train = len(params) > 0
        ctx.save_for_backward(s, x, stats, uv, pre, a, ip, *((cat, p) if ctx.train else ()))
        ctx.module, ctx.do_norm = module, do_norm
        ctx.set_materialize_grads(False)   # an output without a consumer (the last block's x_out: the head reads s only) arrives as None
        return s_out, x_out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_s_out, g_x_out):
        module = ctx.module
        s, x, stats, uv, pre, a, ip = ctx.saved_tensors[:7]
        n, D = x.shape
        F, mul = module.node_dim, module.node_irreps.mul3()
        C = sum(mul)
        dt, dev = s.dtype, s.device
        g_s_out = torch.zeros_like(s) if g_s_out is None else g_s_out.contiguous()
        g_x_out = None if g_x_out is None else g_x_out.contiguous()   # None = zero: the kernels skip its terms
        g_a = torch.empty_like(a)
        g_ip = torch.empty_like(ip)
        # dL/dU of this stage (g_x_out a_vv) is formed inside xeq_uv_reduce_bwd: no write here, no read-modify-write there
        call("xeq_update_out_bwd", dtype_code(s), ptr(g_s_out), ptr(g_x_out), ptr(uv), ptr(a), ptr(ip), n, F, mul3(mul), ptr(g_a),
             ptr(g_ip), None, stream())
        if ctx.train:
            return UpdateBlock._backward_with_params(ctx, g_s_out, g_x_out, g_a, g_ip, linear_module_bwd(module.dot_lin, g_ip))
        g_cat, g_p = mlp_and_linear_bwd(module.update_mlp, g_a, pre, module.dot_lin, g_ip)   # [g_shat | g_v]; dL/dp
        frag = _packed_uv_frag(module)
        if frag is not None and g_cat.is_contiguous():   # dL/dU, dL/dV -> dL/dxhat -> reverse of both norms in one matrix-core launch
            lw, ew = (module.norm.weight, module.o3norm.affine_weight) if ctx.do_norm else (None, None)
            if n <= UV_BWD_FUSE_NORM_MAX_NODES:
                g_s, g_x = torch.empty_like(s), torch.empty_like(x)
                g_xhat = None
            else:
                g_s = g_x = None
                g_xhat = torch.empty(n * D, dtype=dt, device=dev)
            call("xeq_update_uv_bwd", ptr(uv), ptr(g_p), ptr(g_cat), F + C, ptr(g_x_out), ptr(g_s_out), ptr(a), a.shape[1], ptr(s), ptr(x),
                 ptr(stats), ptr(lw), ptr(ew), n, F, mul3(mul), int(ctx.do_norm), ptr(frag[2][0]), ptr(frag[2][1]), ptr(frag[2][2]),
                 float(module.invariant.eps), ptr(g_s), ptr(g_x), ptr(g_xhat), stream())
            if g_xhat is not None:
                g_s, g_x = _norm_bwd(s, x, module.norm, module.o3norm, stats, ctx.do_norm, F, mul, g_cat, F + C, g_xhat, g_s_out, g_x_out)
            return g_s, g_x, None
        g_uv = torch.empty_like(uv)
        if g_x_out is None:
            g_x_out = torch.zeros_like(x)   # the kernel chain wants the tensor
        call("xeq_uv_reduce_bwd", dtype_code(s), ptr(uv), ptr(g_p), ptr(g_cat), F + C, F, n, mul3(mul), float(module.invariant.eps),
             ptr(g_x_out), ptr(a), ptr(g_uv), stream())
        packs, _ = _packed_uv(module)
        g_xhat = torch.empty(n * D, dtype=dt, device=dev)
        for (l, m, gb), (_, _, gub), W in zip(_bt_blocks(g_xhat, n, mul, 1), _bt_blocks(g_uv, n, mul, 2), packs):
            torch.mm(gub, W.t(), out=gb)
        g_s, g_x = _norm_bwd(s, x, module.norm, module.o3norm, stats, ctx.do_norm, F, mul, g_cat, F + C, g_xhat, g_s_out, g_x_out)
        return g_s, g_x, None

    @staticmethod
    def _backward_with_params(ctx, g_s_out, g_x_out, g_a, g_ip, g_p):
        """Training pass: the unfused reverse chain (dL/dU, dL/dV exist as a tensor there), the weight gradients as GEMMs over
        node tensors, in update_params' order."""
        module = ctx.module
        s, x, stats, uv, pre, a, ip, cat, p = ctx.saved_tensors
        n, D = x.shape
        F, mul = module.node_dim, module.node_irreps.mul3()
        C = sum(mul)
        d_dot = _wgrad(g_ip, p)
        d_w1, d_b1, d_w2, d_b2, g_cat = _mlp_param_grads(module.update_mlp, cat, pre, g_a)   # g_cat = [g_shat | g_v]
        if g_x_out is None:
            g_x_out = torch.zeros_like(x)
        g_uv = torch.empty_like(uv)
        call("xeq_uv_reduce_bwd", dtype_code(s), ptr(uv), ptr(g_p), ptr(g_cat), F + C, F, n, mul3(mul), float(module.invariant.eps),
             ptr(g_x_out), ptr(a), ptr(g_uv), stream())
        _, xhat, _, _ = _norm_fwd(s, x, module.norm, module.o3norm, F, mul)     # recomputed: one launch, the forward did not keep it
        packs, bias = _packed_uv(module)
        g_xhat = torch.empty(n * D, dtype=s.dtype, device=s.device)
        d_wu, d_wv, d_bu, d_bv = [], [], None, None
        for (l, m, gb), (_, _, gub), (_, _, xb), W in zip(_bt_blocks(g_xhat, n, mul, 1), _bt_blocks(g_uv, n, mul, 2),
                                                          _bt_blocks(xhat, n, mul, 1), packs):
            torch.mm(gub, W.t(), out=gb)
            d_pack = _wgrad(xb, gub) / math.sqrt(m)              # [mul, 2 mul]: [dL/dW_U | dL/dW_V] of this l
            d_wu.append(d_pack[:, :m].reshape(-1))
            d_wv.append(d_pack[:, m:].reshape(-1))
            if l == 0 and bias is not None:
                col = gub.sum(0)
                d_bu, d_bv = col[:m], col[m:]
        g_s, g_x = _norm_bwd(s, x, module.norm, module.o3norm, stats, ctx.do_norm, F, mul, g_cat, F + C, g_xhat, g_s_out, g_x_out)
        grads = list(_norm_param_grads(s, x, stats, g_cat[:, :F], g_xhat, F, mul)) if ctx.do_norm else []
        zero_b = lambda b: torch.zeros_like(b)
        grads += [torch.cat(d_wu), d_bu if d_bu is not None else zero_b(module.update_U.bias),
                  torch.cat(d_wv), d_bv if d_bv is not None else zero_b(module.update_V.bias), d_w1, d_b1, d_w2, d_b2, d_dot]
        return (g_s, g_x, None, *grads)

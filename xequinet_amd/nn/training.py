"""Training pass of the XPaiNN energy path: parameter gradients and the double backward of a force loss.

The fused inference blocks (nn/fused.py) carry an explicit reverse pass w.r.t. node features and edge vectors only.
A training step needs dL/dtheta and, when forces enter the loss, the derivative of the force evaluation itself
(``create_graph=training``, nn/basic.py:143-159).  This module is that pass: the reference's op sequence of every block
(nn/basic.py:60-140, nn/xpainn.py:57-83, :128-161, :206-231, nn/o3layer.py:12-171, nn/output.py:114-128) written on
differentiable device tensor operations, so autograd supplies both orders.  It runs on the GPU (device memory, library
GEMMs, ATen elementwise / index kernels); it is not on the inference hot path and does not use the hand-written
kernels, and it refuses host tensors like every other entry of the package.

Layout: node features keep the reference's flat irreps layout [N, sum mul (2l+1)], channel-major inside a block; the
per-edge spherical harmonics are kept once per l ([E, 2l+1]) instead of the reference's mul-fold copies [E, 480].
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch

from .. import keys, lib

TRAIN_PASS = "_xeq_train_pass"      # data-dict flag set by BaseModel.forward: every block of this pass takes the training form
_RSH = "_xeq_train_rsh"             # per-l list of Y_l [E, 2l+1] written by the embedding
_REC = "_xeq_train_records"         # per-edge records [E, roundup(B, 4) + 12] of the kernel message (ops.training_records), or absent
# False: the message block of this pass stays on ATen tensor operations (the cross-check of the kernel form, tests/test_gpu_training.py)
NATIVE_MESSAGE = True
# False: norms, invariants and the update block's products stay on ATen tensor operations too (csrc/xeq_train_node.hip otherwise)
NATIVE_NODE = True
# True: the linear layers as training_ops.linear (xeq::linear: C++ autograd nodes closed under differentiation), i.e. every reduction over
# the N rows (weight gradients of both orders) on xeq_wgrad instead of the library.  QM9-1024 energy+force step 38.6 -> 34.4 ms.  (A first
# measurement after three warm-up steps had it 5 ms SLOWER: the variant's new buffer sizes were still growing the caching allocator;
# profiles/r04_train_step.txt)
NATIVE_LINEAR = True
# set by edge_data at the start of every training pass: forces or the virial enter the result, i.e. the pass will be differentiated twice.
# Only then do the linear layers take the xeq::linear form: an energy-only pass on this module (model.native_training = False) has ~12 ms of
# GPU work per QM9-1024 step, too little to hide the nodes' host cost behind (12.7 ms with torch.nn's layers, 18-22 ms with these).  A
# choice between two correct forms, so a plain module variable is enough.
_SECOND_ORDER_PASS = False
# data-dict flag set by BaseModel.forward for a training pass whose loss reads energies only (no forces, no virial): the blocks stay on
# the fused HIP kernels and hand their parameters to the block functions, which return the parameter gradients (nn/fused.py)
PARAM_GRADS = "_xeq_param_grads"


def native_pass_supported(model: torch.nn.Module) -> bool:
    """Every block of the model has the fused form with parameter gradients (the blocks of this package with ``fused`` on)."""
    # (the parameter-gradient kernels of the radial filter take the Bessel and Gaussian bases: xeq_message_param_grad)
    return (all(getattr(m, "fused", True) for m in model.modules())
            and all(getattr(m, "kind", "bessel") in ("bessel", "gaussian") for m in model.modules() if hasattr(m, "params") and hasattr(m, "num_basis")))


def wants_training_pass(module: torch.nn.Module) -> bool:
    """A module in training mode whose parameters ask for gradients, with autograd recording."""
    return module.training and torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


_warned_eval_params = False


def active(module: torch.nn.Module, data: Dict[str, torch.Tensor]) -> bool:
    flag = data.get(TRAIN_PASS)
    on = wants_training_pass(module) if flag is None else bool(flag)
    global _warned_eval_params
    if not on and not _warned_eval_params and not module.training and torch.is_grad_enabled():
        # eval mode with parameters that ask for gradients: the fused inference kernels differentiate with respect to the
        # positions only, so loss.backward() through this call leaves the block parameters' .grad empty -- say so once
        # (the check inside the autograd.Function of an earlier round could never fire: grad mode is off in there)
        if any(p.requires_grad for p in module.parameters()):
            import warnings

            _warned_eval_params = True
            warnings.warn("xequinet_amd: parameters require grad but the model is in eval mode: the fused inference path returns "
                          "forces / virials only, no parameter gradients.  Use model.train() for the training pass "
                          "(nn/training.py) or model.requires_grad_(False) to silence this.", stacklevel=3)
    return on


def _blocks(x: torch.Tensor, irreps) -> List[torch.Tensor]:
    """[N, mul, 2l+1] views of the flat layout, one per block."""
    return [x[:, off : off + mul * (2 * l + 1)].reshape(x.shape[0], mul, 2 * l + 1) for mul, l, off, _ in irreps.blocks()]


def _flat(parts: List[torch.Tensor]) -> torch.Tensor:
    return torch.cat([p.reshape(p.shape[0], -1) for p in parts], dim=-1)


# ---- edge geometry (nn/basic.py:60-140) ---------------------------------------------------------------------------------
def edge_data(data: Dict[str, torch.Tensor], compute_forces: bool, compute_virial: bool) -> Dict[str, torch.Tensor]:
    global _SECOND_ORDER_PASS
    _SECOND_ORDER_PASS = bool(compute_forces or compute_virial)
    pos = data[keys.POSITIONS]
    lib.require_hip(pos)
    ei = data[keys.EDGE_INDEX]
    if keys.BATCH not in data:
        data[keys.BATCH] = torch.zeros(pos.shape[0], dtype=torch.long, device=pos.device)
        data[keys.BATCH_PTR] = torch.tensor([0, pos.shape[0]], dtype=torch.long, device=pos.device)
    batch = data[keys.BATCH].long()
    n_graphs = data[keys.BATCH_PTR].numel() - 1
    has_cell = keys.CELL in data
    cell = data[keys.CELL] if has_cell else None
    if compute_forces:
        pos.requires_grad_()
    strain = torch.zeros((n_graphs, 3, 3), dtype=pos.dtype, device=pos.device)
    if compute_virial:
        strain.requires_grad_()
        sym = 0.5 * (strain + strain.transpose(1, 2))
        pos = pos + torch.bmm(pos.unsqueeze(1), sym.index_select(0, batch)).squeeze(1)
        if has_cell:
            cell = cell + torch.bmm(cell, sym)
    center, neighbor = ei[keys.CENTER_IDX].long(), ei[keys.NEIGHBOR_IDX].long()
    vectors = pos.index_select(0, center) - pos.index_select(0, neighbor)
    if has_cell:
        offsets = data[keys.CELL_OFFSETS].to(pos.dtype)
        cell_e = cell.index_select(0, batch.index_select(0, neighbor))
        vectors = vectors - torch.einsum("ni,nij->nj", offsets, cell_e)
    data.update({keys.EDGE_LENGTH: torch.linalg.norm(vectors, dim=-1), keys.EDGE_VECTOR: vectors, keys.STRAIN: strain})
    return data


# ---- radial basis, envelope, spherical harmonics (nn/rbf.py, nn/xpainn.py:66-74) ------------------------------------------
def radial_basis(rbf: torch.nn.Module, dist: torch.Tensor) -> torch.Tensor:
    """dist [E, 1] -> [E, num_basis]; the basis parameters (freq / mean, std) are trainable as in the reference."""
    if rbf.kind == "bessel":        # nn/rbf.py:148-152
        return rbf.coeff * torch.sin(rbf.freq * dist) / (dist + rbf.eps)
    if rbf.kind == "gaussian":      # nn/rbf.py:128-131
        std = rbf.std.abs() + rbf.eps
        return torch.exp(-0.5 * ((dist - rbf.mean) / std) ** 2) / (std * math.sqrt(2 * math.pi))
    if rbf.kind == "expbern":       # nn/rbf.py:186-191
        x = -torch.nn.functional.softplus(rbf._alpha) * dist
        return torch.exp(rbf.logc + rbf.n * x + rbf.v * torch.log(-torch.expm1(x)))
    if rbf.kind == "expnorm":       # nn/rbf.py:204-207
        return torch.exp(-rbf.beta * torch.square(torch.exp(-dist) - rbf.mu))
    raise NotImplementedError(f"radial basis {rbf.kind}")


def envelope(cutoff_fn: torch.nn.Module, dist: torch.Tensor) -> torch.Tensor:
    rc = cutoff_fn.cutoff
    if cutoff_fn.kind == "cosine":          # nn/rbf.py:43-57
        val = 0.5 * (torch.cos(math.pi * dist / rc) + 1.0)
    elif cutoff_fn.kind == "polynomial":    # nn/rbf.py:60-73, order 3
        r = dist / rc
        val = 1.0 - 10.0 * r**3 + 15.0 * r**4 - 6.0 * r**5
    else:
        raise NotImplementedError(f"cutoff function {cutoff_fn.kind}")
    return torch.where(dist < rc, val, torch.zeros_like(dist))


def spherical_harmonics(vectors: torch.Tensor, lmax: int) -> List[torch.Tensor]:
    """Component-normalised Y_l of the unit edge vector for l = 0..lmax in the order the reference gets from
    ``sph_harm(vectors[:, [1, 2, 0]])``: e3nn's (x, y, z) are the edge vector's (y, z, x)."""
    u = vectors / torch.linalg.norm(vectors, dim=-1, keepdim=True).clamp_min(1e-12)
    x, y, z = u[:, 1], u[:, 2], u[:, 0]
    out = [torch.ones_like(x).unsqueeze(-1)]
    if lmax >= 1:
        out.append(math.sqrt(3.0) * torch.stack([x, y, z], dim=-1))
    if lmax >= 2:
        c15, c5 = math.sqrt(15.0), math.sqrt(5.0)
        out.append(torch.stack([c15 * x * z, c15 * x * y, c5 * (y * y - 0.5 * (x * x + z * z)), c15 * y * z,
                                0.5 * c15 * (z * z - x * x)], dim=-1))
    if lmax >= 3:
        raise NotImplementedError("l > 2")
    return out


def embedding(module, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """XEmbedding.forward (nn/xpainn.py:57-83)."""
    z = data[keys.ATOMIC_NUMBERS]
    vectors = data[keys.EDGE_VECTOR]
    lib.require_hip(vectors)
    s = module.embedding(z.long() if isinstance(module.embedding, torch.nn.Embedding) else z)
    dist = data[keys.EDGE_LENGTH].unsqueeze(-1)
    data[keys.NODE_INVARIANT] = s
    data[keys.RADIAL_BASIS_FUNCTION] = radial_basis(module.rbf, dist)
    data[keys.ENVELOPE_FUNCTION] = envelope(module.cutoff_fn, dist)
    data[_RSH] = spherical_harmonics(vectors, module.node_irreps.lmax)
    if NATIVE_MESSAGE and s.is_cuda and module.node_irreps.lmax <= 2 and data[keys.RADIAL_BASIS_FUNCTION].shape[1] <= 32:
        # the three message blocks share the geometry: one record per edge, differentiable in the vectors and the basis parameters
        from .. import ops

        rsh, env = data[_RSH], data[keys.ENVELOPE_FUNCTION]
        E = env.shape[0]
        y1 = rsh[1] if len(rsh) > 1 else env.new_zeros((E, 3))
        y2 = rsh[2] if len(rsh) > 2 else env.new_zeros((E, 5))
        data[_REC] = ops.training_records(data[keys.RADIAL_BASIS_FUNCTION] * env, env, y1, y2)
    data[keys.NODE_EQUIVARIANT] = torch.zeros((s.shape[0], module.node_irreps.dim), dtype=s.dtype, device=s.device)
    return data


# ---- normalisation (nn.LayerNorm, nn/o3layer.py:145-171) -------------------------------------------------------------------
def equivariant_layer_norm(norm, x: torch.Tensor) -> torch.Tensor:
    irreps = norm.irreps
    parts = _blocks(x, irreps)
    is_scalar = [ir.l == 0 and ir.p == 1 for _, ir in irreps]
    n_scalar = sum(mul for (mul, _), sc in zip(irreps, is_scalar) if sc)
    mean = sum(p.sum(dim=(1, 2)) for p, sc in zip(parts, is_scalar) if sc) / n_scalar           # over all even scalars
    parts = [p - mean.view(-1, 1, 1) if sc else p for p, sc in zip(parts, is_scalar)]
    sq = torch.cat([(p * p).sum(-1) for p in parts], dim=-1)                                    # [N, num_irreps]
    inv_rms = torch.rsqrt(sq.mean(dim=1, keepdim=True) + norm.eps).unsqueeze(-1)
    out, ch, sc_off = [], 0, 0
    for p, sc in zip(parts, is_scalar):
        mul = p.shape[1]
        q = p * inv_rms * norm.affine_weight[ch : ch + mul].view(1, mul, 1)
        if sc:
            q = q + norm.affine_bias[sc_off : sc_off + mul].view(1, mul, 1)
            sc_off += mul
        out.append(q)
        ch += mul
    return _flat(out)


def _mlp(seq, x: torch.Tensor) -> torch.Tensor:
    """scalar_mlp / update_mlp / dot_lin / out_mlp: the module itself, or (NATIVE_LINEAR) with the weight-gradient products on
    ``xeq_wgrad`` (training_ops.LinearFn)."""
    if NATIVE_LINEAR and _SECOND_ORDER_PASS and x.is_cuda and (isinstance(seq, torch.nn.Linear) or isinstance(seq, torch.nn.Sequential)):
        from .training_ops import mlp

        return mlp(seq, x)
    return seq(x)


def _native_node(module, s: torch.Tensor):
    """(node_dim, mul) when the block's norms / products can take the kernels of csrc/xeq_train_node.hip: a device tensor, an affine
    LayerNorm + EquivariantLayerNorm pair, irreps of one block per l <= 2 in ascending order whose l = 0 block is even."""
    if not (NATIVE_NODE and s.is_cuda and s.dtype in (torch.float32, torch.float64)):
        return None
    ln, eq = module.norm, getattr(module, "o3norm", None)
    if not isinstance(ln, torch.nn.LayerNorm) or ln.weight is None or ln.bias is None or eq is None or not hasattr(eq, "affine_weight"):
        return None
    irreps = eq.irreps
    try:
        mul = tuple(irreps.mul3())
    except NotImplementedError:
        return None
    if any(ir.l == 0 and ir.p != 1 for _, ir in irreps) or eq.affine_bias.numel() != mul[0] or eq.affine_weight.numel() != sum(mul):
        return None
    return int(s.shape[1]), mul


def _norms(module, s: torch.Tensor, x: torch.Tensor, bt: bool = False):
    """(LayerNorm(s), EquivariantLayerNorm(x)); ``bt`` asks for xhat in the BT layout (a flat buffer) and is honoured by the kernel form
    only -- the caller checks ``_native_node`` first."""
    if isinstance(module.norm, torch.nn.Identity):
        return s, x
    nat = _native_node(module, s)
    if nat is not None:
        from .training_ops import NormFn

        F, mul = nat
        return NormFn.apply(s, x, module.norm.weight, module.norm.bias, module.o3norm.affine_weight, module.o3norm.affine_bias,
                            (F, mul, float(module.norm.eps), float(module.o3norm.eps), 1 if bt else 0))
    return module.norm(s), equivariant_layer_norm(module.o3norm, x)


# ---- message (nn/xpainn.py:128-161) --------------------------------------------------------------------------------------------
def message(module, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    s0, x0 = data[keys.NODE_INVARIANT], data[keys.NODE_EQUIVARIANT]
    lib.require_hip(s0)
    irreps, C = module.node_irreps, module.node_num_irreps
    ei = data[keys.EDGE_INDEX]
    center, neighbor = ei[keys.CENTER_IDX].long(), ei[keys.NEIGHBOR_IDX].long()
    s, x = _norms(module, s0, x0)
    scalar_out = _mlp(module.scalar_mlp, s)
    rec = data.get(_REC)
    if rec is not None:
        # the edge side on the kernels: the aggregation, its reverse pass and the reverse of that (ops.DiffMessage); nothing of size
        # [E, 576] outlives a launch pair
        from .. import ops
        from .basic import edge_graph

        graph = edge_graph(data)
        try:
            cfg = (int(module.rbf_lin.weight.shape[1]), int(module.node_dim), tuple(irreps.mul3()))
        except NotImplementedError:     # irreps the kernels do not take (several blocks per l, ...): the tensor form below
            cfg = None
        if cfg is not None and ops.diff_message_supported(scalar_out, graph, cfg):
            d_s, d_x = ops.DiffMessage.apply(scalar_out, x, rec, module.rbf_lin.weight, module.rbf_lin.bias, graph, cfg)
            data[keys.NODE_INVARIANT] = s0 + d_s
            data[keys.NODE_EQUIVARIANT] = x0 + d_x
            return data
    if getattr(data.get(keys.EDGE_GRAPH), "edge_count_on_device", False):
        # a capacity-sized edge list (train.GraphedTrainStep): the slots behind the true count hold stale pairs, which the tensor form
        # below would walk as edges (one row per SLOT); the kernel form above walks the row pointer and stops at the true count
        raise NotImplementedError("the tensor form of the training pass does not take a capacity-sized edge list")
    filt = module.rbf_lin(data[keys.RADIAL_BASIS_FUNCTION]) * data[keys.ENVELOPE_FUNCTION]
    filt = scalar_out.index_select(0, neighbor) * filt
    gate_state, gate_edge, msg_s = torch.split(filt, [C, C, module.node_dim], dim=-1)
    x_j = _blocks(x.index_select(0, neighbor), irreps)
    rsh = data[_RSH]
    msg_x, ch = [], 0
    for (mul, ir), xb in zip(irreps, x_j):
        gs = gate_state[:, ch : ch + mul].unsqueeze(-1)
        ge = gate_edge[:, ch : ch + mul].unsqueeze(-1)
        msg_x.append(xb * gs + rsh[ir.l].unsqueeze(1) * ge)        # l x 0e -> l, component normalisation: factor 1
        ch += mul
    data[keys.NODE_INVARIANT] = s0.index_add(0, center, msg_s)
    data[keys.NODE_EQUIVARIANT] = x0.index_add(0, center, _flat(msg_x))
    return data


# ---- update (nn/xpainn.py:206-231) -----------------------------------------------------------------------------------------------
def o3_linear(lin, x: torch.Tensor) -> List[torch.Tensor]:
    """o3.Linear(irreps, irreps, biases=True): out[w, m] = mul^-1/2 sum_u W_l[u, w] x[u, m] (+ bias on 0e), per block."""
    out, woff, boff = [], 0, 0
    for (mul, ir), xb in zip(lin.irreps_in, _blocks(x, lin.irreps_in)):
        W = lin.weight[woff : woff + mul * mul].view(mul, mul)
        woff += mul * mul
        ob = torch.einsum("uw,num->nwm", W, xb) * (1.0 / math.sqrt(mul))
        if ir.l == 0 and ir.p == 1 and lin.bias.numel() > 0:
            ob = ob + lin.bias[boff : boff + mul].view(1, mul, 1)
            boff += mul
        out.append(ob)
    return out


def _uv_weights_ok(module, mul) -> bool:
    want_w = sum(m * m for m in mul)
    for lin in (module.update_U, module.update_V):
        if lin.weight.numel() != want_w or lin.bias.numel() not in (0, mul[0]):
            return False
    return module.update_U.bias.numel() == module.update_V.bias.numel()


def _update_on_kernels(module, data, s0, x0, mul):
    """``update`` with the norms, the Invariant / channel dot and the products on csrc/xeq_train_node.hip (training_ops) and the
    o3.Linear pair as one library GEMM per l on BT rows: [N (2l+1), mul_l] x [mul_l, 2 mul_l] = (U | V)."""
    from .training_ops import UpdateOutFn, UvFn, linear

    N, C, F = s0.shape[0], module.node_num_irreps, module.node_dim
    s, x_bt = _norms(module, s0, x0, bt=True)
    uv, woff, base = [], 0, 0
    lu, lv = module.update_U, module.update_V
    for l in range(3):
        m = mul[l]
        if m == 0:
            uv.append(s0.new_zeros(0))
            continue
        rows = N * (2 * l + 1)
        xl = x_bt[N * base : N * base + rows * m].view(rows, m)
        base += (2 * l + 1) * m
        w = torch.cat([lu.weight[woff : woff + m * m].view(m, m), lv.weight[woff : woff + m * m].view(m, m)], dim=1) * (1.0 / math.sqrt(m))
        woff += m * m
        bias = torch.cat([lu.bias, lv.bias]) if l == 0 and lu.bias.numel() > 0 else None
        if NATIVE_LINEAR and _SECOND_ORDER_PASS:
            uv.append(linear(xl, w.t(), bias))
        else:
            uv.append(xl @ w if bias is None else torch.addmm(bias, xl, w))
    vd = UvFn.apply(uv[0], uv[1], uv[2], (mul, float(module.invariant.eps)))
    a = _mlp(module.update_mlp, torch.cat([s, vd[:, :C]], dim=-1))
    inner = _mlp(module.dot_lin, vd[:, C:])
    d_s, d_x = UpdateOutFn.apply(uv[0], uv[1], uv[2], a, inner, (int(F), mul))
    data[keys.NODE_INVARIANT] = s0 + d_s
    data[keys.NODE_EQUIVARIANT] = x0 + d_x
    return data


def update(module, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    s0, x0 = data[keys.NODE_INVARIANT], data[keys.NODE_EQUIVARIANT]
    lib.require_hip(s0)
    C, F = module.node_num_irreps, module.node_dim
    nat = _native_node(module, s0)
    if nat is not None and not isinstance(module.norm, torch.nn.Identity) and _uv_weights_ok(module, nat[1]):
        return _update_on_kernels(module, data, s0, x0, nat[1])
    s, x = _norms(module, s0, x0)
    U, V = o3_linear(module.update_U, x), o3_linear(module.update_V, x)
    eps = module.invariant.eps
    v_norm = torch.cat([torch.sqrt((v * v).sum(-1) + eps**2) - eps for v in V], dim=-1)          # Invariant, o3layer.py:40-44
    a_vv, a_sv, a_ss = torch.split(module.update_mlp(torch.cat([s, v_norm], dim=-1)), [C, F, F], dim=-1)
    d_x, ch = [], 0
    for u in U:
        mul = u.shape[1]
        d_x.append(u * a_vv[:, ch : ch + mul].unsqueeze(-1))
        ch += mul
    inner = module.dot_lin(torch.cat([(u * v).sum(-1) for u, v in zip(U, V)], dim=-1))           # EquivariantDot + dot_lin
    data[keys.NODE_INVARIANT] = s0 + a_sv * inner + a_ss
    data[keys.NODE_EQUIVARIANT] = x0 + _flat(d_x)
    return data


# ---- energy head (nn/output.py:114-128) --------------------------------------------------------------------------------------------
def energy_out(module, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    lib.require_hip(data[keys.NODE_INVARIANT])
    atom = _mlp(module.out_mlp, data[keys.NODE_INVARIANT]).reshape(-1)
    if keys.ATOMIC_ENERGIES in data:
        atom = data[keys.ATOMIC_ENERGIES] + atom
    n_graphs = data[keys.BATCH_PTR].numel() - 1
    total = torch.zeros(n_graphs, dtype=atom.dtype, device=atom.device).index_add(0, data[keys.BATCH].long(), atom)
    data[keys.ATOMIC_ENERGIES] = atom
    data[keys.TOTAL_ENERGY] = total
    return data

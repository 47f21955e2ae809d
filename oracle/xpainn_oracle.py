"""CPU oracle for the XPaiNN energy+force hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, autograd) restatement of the reference's
algorithm for the path named in BASELINE.json.  It is the *checker*: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product (``xequinet_amd``) never imports it
and fails loudly when its HIP library is missing.

Parity status
-------------
* e3nn-free arithmetic (edge geometry, radial bases, envelopes, PBC radius
  graph, PaiNN-twin message/update dataflow) is PINNED against outputs of the
  reference itself, generated in the build container by
  ``tests/golden/make_golden.py`` and committed under ``tests/golden/``.
* e3nn-dependent arithmetic (``o3.SphericalHarmonics``, ``ElementwiseTensorProduct``,
  ``TensorProduct('uuu')``, ``o3.Linear``; e3nn==0.5.1, environment.yaml:139) is a
  third-party dependency that is absent from /root/reference and not
  installable here.  It is restated from e3nn's published definitions and
  pinned by known-answer / property tests only: **parity unpinned** against a
  live e3nn for those four ops.

All ``file:line`` citations are relative to /root/reference/xequinet/.

Layout conventions (e3nn ``mul_ir``): features are [*, D] with blocks
[mul_0 x (2*0+1) | mul_1 x 3 | mul_2 x 5], channel-major / m-minor.
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ----------------------------------------------------------------------------
# keys (keys.py:4-50)
# ----------------------------------------------------------------------------
POSITIONS = "pos"
ATOMIC_NUMBERS = "atomic_numbers"
EDGE_INDEX = "edge_index"
CELL_OFFSETS = "cell_offsets"
CELL = "cell"
BATCH = "batch"
BATCH_PTR = "ptr"
CENTER_IDX = 0
NEIGHBOR_IDX = 1


# ----------------------------------------------------------------------------
# irreps helper (stands in for e3nn.o3.Irreps; only what the path needs)
# ----------------------------------------------------------------------------
def parse_irreps(irreps) -> List[Tuple[int, int, int]]:
    """'128x0e + 64x1o + 32x2e' -> [(128,0,+1),(64,1,-1),(32,2,+1)]."""
    if not isinstance(irreps, str):
        return [tuple(t) for t in irreps]
    out = []
    for term in irreps.split("+"):
        m = re.fullmatch(r"\s*(?:(\d+)\s*x\s*)?(\d+)\s*([eo])\s*", term)
        if m is None:
            raise ValueError(f"cannot parse irreps term {term!r}")
        out.append((int(m.group(1) or 1), int(m.group(2)), 1 if m.group(3) == "e" else -1))
    return out


def irreps_dim(irreps) -> int:
    return sum(mul * (2 * l + 1) for mul, l, _ in parse_irreps(irreps))


def irreps_num(irreps) -> int:
    return sum(mul for mul, _, _ in parse_irreps(irreps))


def _blocks(irreps):
    """yield (mul, l, flat_offset, channel_offset)."""
    off = ch = 0
    for mul, l, _ in parse_irreps(irreps):
        yield mul, l, off, ch
        off += mul * (2 * l + 1)
        ch += mul


# ----------------------------------------------------------------------------
# e3nn restatements  [3P: e3nn 0.5.1, parity unpinned]
# ----------------------------------------------------------------------------
def _sh_e3nn(lmax: int, x: Tensor, y: Tensor, z: Tensor) -> List[Tensor]:
    """Component-normalised real spherical harmonics in e3nn's axis convention
    (y is the polar axis; l=1 is sqrt(3)*(x,y,z)).  sum_m Y_lm^2 = 2l+1 on the
    unit sphere."""
    out = [torch.ones_like(x).unsqueeze(-1)]
    if lmax >= 1:
        out.append(math.sqrt(3.0) * torch.stack([x, y, z], dim=-1))
    if lmax >= 2:
        s15, s5 = math.sqrt(15.0), math.sqrt(5.0)
        out.append(
            torch.stack(
                [
                    s15 * x * z,
                    s15 * x * y,
                    s5 * (y * y - 0.5 * (x * x + z * z)),
                    s15 * y * z,
                    0.5 * s15 * (z * z - x * x),
                ],
                dim=-1,
            )
        )
    if lmax >= 3:
        raise NotImplementedError("oracle restates l <= 2 only")
    return out


def spherical_harmonics(irreps, vec: Tensor, normalize: bool = True) -> Tensor:
    """``o3.SphericalHarmonics(irreps, normalize=True, normalization='component')``
    as built at nn/xpainn.py:49-51: every Y_l is repeated ``mul`` times.
    ``vec`` is in e3nn axis order (the caller passes vec[:, [1,2,0]], xpainn.py:71-74)."""
    if normalize:
        vec = F.normalize(vec, dim=-1)  # x / max(||x||, 1e-12)
    blocks = parse_irreps(irreps)
    lmax = max(l for _, l, _ in blocks)
    ys = _sh_e3nn(lmax, vec[..., 0], vec[..., 1], vec[..., 2])
    parts = []
    for mul, l, _ in blocks:
        parts.append(ys[l].unsqueeze(-2).expand(*ys[l].shape[:-1], mul, 2 * l + 1).reshape(*ys[l].shape[:-1], mul * (2 * l + 1)))
    return torch.cat(parts, dim=-1)


def elementwise_tp(irreps, x: Tensor, g: Tensor) -> Tensor:
    """``o3.ElementwiseTensorProduct(irreps, f'{C}x0e')`` (xpainn.py:119-121):
    out[u,m] = x[u,m] * g[u]  (l x 0e -> l, component normalisation => factor 1)."""
    parts = []
    for mul, l, off, ch in _blocks(irreps):
        d = 2 * l + 1
        xb = x[..., off : off + mul * d].reshape(*x.shape[:-1], mul, d)
        gb = g[..., ch : ch + mul].unsqueeze(-1)
        parts.append((xb * gb).reshape(*x.shape[:-1], mul * d))
    return torch.cat(parts, dim=-1)


def equivariant_dot(irreps, a: Tensor, b: Tensor) -> Tensor:
    """``EquivariantDot`` (o3layer.py:79-109) = TensorProduct 'uuu' l x l -> 0e with
    path weight ir.dim and component normalisation = plain sum_m a[u,m] b[u,m]."""
    parts = []
    for mul, l, off, _ in _blocks(irreps):
        d = 2 * l + 1
        ab = a[..., off : off + mul * d].reshape(*a.shape[:-1], mul, d)
        bb = b[..., off : off + mul * d].reshape(*b.shape[:-1], mul, d)
        parts.append((ab * bb).sum(-1))
    return torch.cat(parts, dim=-1)


def invariant(irreps, x: Tensor, squared: bool = False, eps: float = 1e-5) -> Tensor:
    """``Invariant`` (o3layer.py:12-44): sqrt(sum_m x^2 + eps^2) - eps."""
    out = equivariant_dot(irreps, x, x)
    if squared:
        return out
    return torch.sqrt(out + eps**2) - eps


def o3_linear(irreps, x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """``o3.Linear(irreps, irreps, biases=True)`` (xpainn.py:186-187).
    out[w,m] = mul_in^-1/2 sum_u W_l[u,w] x[u,m]; flat weight = concat over l of
    row-major [mul_in, mul_out] blocks; bias only on 0e."""
    parts = []
    woff = 0
    boff = 0
    for mul, l, off, _ in _blocks(irreps):
        d = 2 * l + 1
        W = weight[woff : woff + mul * mul].reshape(mul, mul)
        woff += mul * mul
        xb = x[..., off : off + mul * d].reshape(*x.shape[:-1], mul, d)
        ob = torch.einsum("uw,...ui->...wi", W, xb) / math.sqrt(mul)
        if l == 0 and bias is not None:
            ob = ob + bias[boff : boff + mul].unsqueeze(-1)
            boff += mul
        parts.append(ob.reshape(*x.shape[:-1], mul * d))
    return torch.cat(parts, dim=-1)


def equivariant_layer_norm(irreps, x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """``EquivariantLayerNorm.forward`` (o3layer.py:145-171)."""
    blocks = list(_blocks(irreps))
    scalar_index = []
    for (mul, l, off, _), (_, _, p) in zip(blocks, parse_irreps(irreps)):
        if l == 0 and p == 1:
            scalar_index.extend(range(off, off + mul))
    idx = torch.tensor(scalar_index, dtype=torch.long)
    ns = len(scalar_index)
    scalar = x[:, idx]
    x = x.index_add(1, idx, -scalar.mean(dim=1, keepdim=True).repeat(1, ns))  # :150-156
    sq = invariant(irreps, x, squared=True)  # :158
    inv_rms = torch.reciprocal(torch.sqrt(sq.mean(dim=1, keepdim=True) + eps))  # :159-161
    x = x * inv_rms
    x = elementwise_tp(irreps, x, weight.unsqueeze(0))  # :164
    x = x.index_add(1, idx, bias.unsqueeze(0).repeat(x.shape[0], 1))  # :165-169
    return x


# ----------------------------------------------------------------------------
# e3nn-free pieces (pinned against the reference by tests/golden)
# ----------------------------------------------------------------------------
def bessel_rbf(dist: Tensor, freq: Tensor, cutoff: float, eps: float = 1e-5) -> Tensor:
    """``SphericalBesselj0.forward`` (nn/rbf.py:148-152). dist [E,1], freq [1,B]."""
    return math.sqrt(2.0 / cutoff) * torch.sin(freq * dist) / (dist + eps)


def cosine_cutoff(dist: Tensor, cutoff: float) -> Tensor:
    """``CosineCutoff`` through ``CutoffFunction.forward`` (nn/rbf.py:43-57)."""
    return torch.where(dist < cutoff, 0.5 * (torch.cos(math.pi * dist / cutoff) + 1.0), torch.zeros_like(dist))


def polynomial_cutoff(dist: Tensor, cutoff: float, order: int = 3) -> Tensor:
    """``PolynomialCutoff`` (nn/rbf.py:60-73)."""
    p = order
    r = dist / cutoff
    val = 1 - 0.5 * (p + 1) * (p + 2) * r**p + p * (p + 2) * r ** (p + 1) - 0.5 * p * (p + 1) * r ** (p + 2)
    return torch.where(dist < cutoff, val, torch.zeros_like(dist))


def gaussian_rbf(dist: Tensor, mean: Tensor, std: Tensor, eps: float = 1e-5) -> Tensor:
    """``GaussianSmearing.forward`` (nn/rbf.py:128-131)."""
    std = std.abs() + eps
    coeff = 1 / (std * math.sqrt(2 * math.pi))
    return coeff * torch.exp(-0.5 * ((dist - mean) / std) ** 2)


def exp_bernstein_rbf(dist: Tensor, alpha_raw: Tensor, logc: Tensor, n: Tensor, v: Tensor) -> Tensor:
    """``ExponentialBernstein.forward`` (nn/rbf.py:186-191). dist [E,1]; alpha_raw: the stored ``_alpha`` (softplus inside);
    logc / n / v: the module's buffers (log binomials, B - 1 - k, k)."""
    x = -F.softplus(alpha_raw) * dist
    return torch.exp(logc + n * x + v * torch.log(-torch.expm1(x)))


def exp_norm_rbf(dist: Tensor, beta: Tensor, mu: Tensor) -> Tensor:
    """``ExponentialNorm.forward`` (nn/rbf.py:204-207)."""
    return torch.exp(-beta * torch.square(torch.exp(-dist) - mu))


def compute_edge_data(data: Dict[str, Tensor], compute_forces: bool = True, compute_virial: bool = False) -> Dict[str, Tensor]:
    """``compute_edge_data`` (nn/basic.py:60-140) incl. the virial/strain branch (:93-107)."""
    pos = data[POSITIONS]
    edge_index = data[EDGE_INDEX]
    single_graph = False
    if BATCH not in data:
        data[BATCH] = torch.zeros(pos.shape[0], dtype=torch.long)
        data[BATCH_PTR] = torch.tensor([0, pos.shape[0]], dtype=torch.long)
        single_graph = True
    elif data[BATCH].max() == 0:
        single_graph = True
    batch = data[BATCH]
    n_graphs = data[BATCH_PTR].numel() - 1 if BATCH_PTR in data else int(batch.max()) + 1
    if compute_forces:
        pos.requires_grad_()
    strain = torch.zeros((n_graphs, 3, 3), dtype=pos.dtype)  # :93-97
    cell = data[CELL] if CELL in data else None
    if compute_virial:  # :99-107
        strain.requires_grad_()
        symm = 0.5 * (strain + strain.transpose(1, 2))
        pos = pos + torch.bmm(pos.unsqueeze(1), symm.index_select(0, batch)).squeeze(1)
        if cell is not None:
            cell = cell + torch.bmm(cell, symm)
    data["strain"] = strain
    center, neighbor = edge_index[CENTER_IDX], edge_index[NEIGHBOR_IDX]
    vec = pos.index_select(0, center) - pos.index_select(0, neighbor)  # :114-116
    if cell is not None:
        co = data[CELL_OFFSETS]
        if single_graph:
            shifts = torch.einsum("ni,ij->nj", co, cell.squeeze(0))  # :121-123
        else:
            cell_batch = cell.index_select(0, batch.index_select(0, neighbor))
            shifts = torch.einsum("ni,nij->nj", co, cell_batch)  # :125-127
        vec = vec - shifts
    data["edge_vector"] = vec
    data["edge_length"] = torch.linalg.norm(vec, dim=-1)  # :131
    return data


# ----------------------------------------------------------------------------
# XPaiNN blocks, functional on a reference-layout state dict
# ----------------------------------------------------------------------------
def _act(name: str):
    name = name.lower()
    return {
        "silu": F.silu,
        "relu": F.relu,
        "leakyrelu": F.leaky_relu,
        "softplus": F.softplus,
        "sigmoid": torch.sigmoid,
        "tanh": torch.tanh,
        "identity": lambda t: t,
    }[name]


class XPaiNNOracle:
    """Functional restatement of ``XPaiNN`` (nn/model.py:49-122) + ``BaseModel.forward``
    (nn/model.py:26-46) for the energy head.  ``sd`` is a state dict in the
    reference's key layout (``mods.embedding.*``, ``mods.message_i.*``,
    ``mods.update_i.*``, ``mods.output_energy.*``)."""

    def __init__(self, sd: Dict[str, Tensor], **kwargs):
        self.sd = sd
        self.node_dim = kwargs.get("node_dim", 128)
        self.irreps = kwargs.get("node_irreps", "128x0e + 64x1o + 32x2e")
        self.num_basis = kwargs.get("num_basis", 20)
        self.cutoff = float(kwargs.get("cutoff", 5.0))
        self.cutoff_fn = kwargs.get("cutoff_fn", "cosine")
        self.rbf_kernel = kwargs.get("rbf_kernel", "bessel")
        self.blocks = kwargs.get("action_blocks", 3)
        self.act = _act(kwargs.get("activation", "silu"))
        self.layer_norm = kwargs.get("layer_norm", True)
        self.C = irreps_num(self.irreps)
        self.D = irreps_dim(self.irreps)
        # eps of Invariant (o3layer.py:15); only the PaiNN-twin test overrides it
        self.invariant_eps = kwargs.get("invariant_eps", 1e-5)

    # -- XEmbedding.forward (xpainn.py:55-83)
    def embedding(self, data):
        p = "mods.embedding."
        sd = self.sd
        z = data[ATOMIC_NUMBERS].long()
        vec = data["edge_vector"]
        dist = data["edge_length"].unsqueeze(-1)
        if p + "embedding.0.embed_ten" in sd:
            feat = sd[p + "embedding.0.embed_ten"][z]  # basic.py:57
            s = F.linear(feat, sd[p + "embedding.1.weight"], sd[p + "embedding.1.bias"])
        else:  # one-hot: nn.Embedding(100, node_dim, padding_idx=0)
            s = sd[p + "embedding.weight"][z]
        data["node_invariant"] = s
        if self.rbf_kernel == "bessel":
            data["rbf"] = bessel_rbf(dist, sd[p + "rbf.freq"], self.cutoff)
        elif self.rbf_kernel == "gaussian":
            data["rbf"] = gaussian_rbf(dist, sd[p + "rbf.mean"], sd[p + "rbf.std"])
        elif self.rbf_kernel == "expbern":
            data["rbf"] = exp_bernstein_rbf(dist, sd[p + "rbf._alpha"], sd[p + "rbf.logc"], sd[p + "rbf.n"], sd[p + "rbf.v"])
        elif self.rbf_kernel == "expnorm":
            data["rbf"] = exp_norm_rbf(dist, sd[p + "rbf.beta"], sd[p + "rbf.mu"])
        else:
            raise NotImplementedError(self.rbf_kernel)
        if self.cutoff_fn == "cosine":
            data["fcut"] = cosine_cutoff(dist, self.cutoff)
        elif self.cutoff_fn == "polynomial":
            data["fcut"] = polynomial_cutoff(dist, self.cutoff)
        else:
            raise NotImplementedError(self.cutoff_fn)
        data["rsh"] = spherical_harmonics(self.irreps, vec[:, [1, 2, 0]])  # :71-74
        data["node_equivariant"] = torch.zeros((s.shape[0], self.D), dtype=s.dtype)  # :77-80
        return data

    def _norms(self, p, s, x):
        sd = self.sd
        if not self.layer_norm:
            return s, x
        s_hat = F.layer_norm(s, (self.node_dim,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
        x_hat = equivariant_layer_norm(self.irreps, x, sd[p + "o3norm.affine_weight"], sd[p + "o3norm.affine_bias"])
        return s_hat, x_hat

    # -- XPainnMessage.forward (xpainn.py:128-161)
    def message(self, i, data):
        p = f"mods.message_{i}."
        sd = self.sd
        s, x = data["node_invariant"], data["node_equivariant"]
        s_hat, x_hat = self._norms(p, s, x)
        center = data[EDGE_INDEX][CENTER_IDX]
        nbr = data[EDGE_INDEX][NEIGHBOR_IDX]
        h = F.linear(self.act(F.linear(s_hat, sd[p + "scalar_mlp.0.weight"], sd[p + "scalar_mlp.0.bias"])),
                     sd[p + "scalar_mlp.2.weight"], sd[p + "scalar_mlp.2.bias"])  # :139
        filt = F.linear(data["rbf"], sd[p + "rbf_lin.weight"], sd[p + "rbf_lin.bias"]) * data["fcut"]  # :140
        fo = h.index_select(0, nbr) * filt  # :142
        g_state, g_edge, m_s = torch.split(fo, [self.C, self.C, self.node_dim], dim=-1)  # :144-148
        m_x = elementwise_tp(self.irreps, x_hat.index_select(0, nbr), g_state)  # :150-152
        m_x = m_x + elementwise_tp(self.irreps, data["rsh"], g_edge)  # :153-154
        data["node_invariant"] = s.index_add(0, center, m_s)  # :158
        data["node_equivariant"] = x.index_add(0, center, m_x)  # :159
        return data

    # -- XPainnUpdate.forward (xpainn.py:206-231)
    def update(self, i, data):
        p = f"mods.update_{i}."
        sd = self.sd
        s, x = data["node_invariant"], data["node_equivariant"]
        s_hat, x_hat = self._norms(p, s, x)
        U = o3_linear(self.irreps, x_hat, sd[p + "update_U.weight"], sd[p + "update_U.bias"])  # :211
        V = o3_linear(self.irreps, x_hat, sd[p + "update_V.weight"], sd[p + "update_V.bias"])  # :212
        v = invariant(self.irreps, V, eps=self.invariant_eps)  # :214
        a = F.linear(self.act(F.linear(torch.cat([s_hat, v], -1), sd[p + "update_mlp.0.weight"], sd[p + "update_mlp.0.bias"])),
                     sd[p + "update_mlp.2.weight"], sd[p + "update_mlp.2.bias"])  # :215-216
        a_vv, a_sv, a_ss = torch.split(a, [self.C, self.node_dim, self.node_dim], dim=-1)  # :218-220
        dx = elementwise_tp(self.irreps, U, a_vv)  # :221
        ip = F.linear(equivariant_dot(self.irreps, U, V), sd[p + "dot_lin.weight"])  # :222-223
        ds = a_sv * ip + a_ss  # :224
        data["node_invariant"] = s + ds
        data["node_equivariant"] = x + dx
        return data

    # -- EnergyOut.forward (output.py:114-128)
    def energy_out(self, data):
        p = "mods.output_energy."
        sd = self.sd
        s = data["node_invariant"]
        e_atom = F.linear(self.act(F.linear(s, sd[p + "out_mlp.0.weight"], sd[p + "out_mlp.0.bias"])),
                          sd[p + "out_mlp.2.weight"], sd[p + "out_mlp.2.bias"]).reshape(-1)
        n_graphs = int(data[BATCH].max()) + 1 if data[BATCH].numel() else 0
        energy = torch.zeros(n_graphs, dtype=e_atom.dtype).index_add(0, data[BATCH], e_atom)  # scatter_sum :124
        data["atomic_energies"] = e_atom
        data["energy"] = energy
        return data

    def forward(self, data: Dict[str, Tensor], compute_forces: bool = True, compute_virial: bool = False,
                training: bool = False) -> Dict[str, Tensor]:
        """``training=True`` is ``model.train()``: the property gradients keep their graph (create_graph=training,
        basic.py:143-159) and nothing is detached, so a loss on energies / forces / virial can be differentiated w.r.t.
        the entries of ``sd``."""
        data = dict(data)
        data[POSITIONS] = data[POSITIONS].detach().clone()
        data = compute_edge_data(data, compute_forces, compute_virial)
        data = self.embedding(data)
        for i in range(self.blocks):
            data = self.message(i, data)
            data = self.update(i, data)
        data = self.energy_out(data)
        out = {"energy": data["energy"], "atomic_energies": data["atomic_energies"]}
        ones = [torch.ones_like(data["energy"])]
        if compute_forces and compute_virial:
            # compute_forces_and_virial (basic.py:181-199)
            g, gs = torch.autograd.grad([data["energy"]], [data[POSITIONS], data["strain"]], ones, create_graph=training)
            out["forces"], out["virial"] = -g, -gs
        elif compute_forces:
            # compute_forces_only (basic.py:143-159)
            (g,) = torch.autograd.grad([data["energy"]], [data[POSITIONS]], ones, create_graph=training)
            out["forces"] = -g
        elif compute_virial:
            # compute_virial_only (basic.py:162-178)
            (gs,) = torch.autograd.grad([data["energy"]], [data["strain"]], ones, create_graph=training)
            out["virial"] = -gs
        return out if training else {k: v.detach() for k, v in out.items()}

    __call__ = forward


# ----------------------------------------------------------------------------
# neighbour lists (numpy; integer outputs are compared bit-exactly)
# ----------------------------------------------------------------------------
def radius_graph_canonical(pos: np.ndarray, ptr: np.ndarray, cutoff: float) -> np.ndarray:
    """Non-PBC radius graph with the semantics of the call at data/transform.py:58-64
    (``torch_cluster.radius_graph``, [3P torch-cluster 1.6.3, parity unpinned]):
    same-graph pairs with d^2 < r^2 (strict, evaluated in the dtype of ``pos``), no
    self loops, unlimited neighbours.  torch_cluster's own emission order cannot be
    observed here, so the edge list is returned in CANONICAL order: sorted by
    (center, neighbor) = (edge_index[0], edge_index[1]).  The edge set is
    symmetric, so this is also a valid torch_cluster result up to ordering."""
    rows, cols = [], []
    r2 = pos.dtype.type(cutoff) * pos.dtype.type(cutoff)
    for g in range(len(ptr) - 1):
        a, b = int(ptr[g]), int(ptr[g + 1])
        p = pos[a:b]
        d = p[:, None, :] - p[None, :, :]
        # same association as the HIP kernel: (dx*dx + dy*dy) + dz*dz, no fma
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        m = d2 < r2
        np.fill_diagonal(m, False)
        i, j = np.nonzero(m)  # row-major => sorted by (i, j)
        rows.append(i + a)
        cols.append(j + a)
    if not rows:
        return np.zeros((2, 0), dtype=np.int64)
    return np.stack([np.concatenate(rows), np.concatenate(cols)]).astype(np.int64)


def _n_images(cell: np.ndarray, pbc: Sequence[bool], cutoff: float) -> List[int]:
    """Images per axis (data/radius_graph.py:61-89), max over the batch."""
    cross23 = np.cross(cell[:, 1], cell[:, 2])
    vol = np.sum(cell[:, 0] * cross23, axis=-1, keepdims=True)
    reps = []
    crosses = [cross23, np.cross(cell[:, 2], cell[:, 0]), np.cross(cell[:, 0], cell[:, 1])]
    for ax in range(3):
        if pbc[ax]:
            inv_min = np.linalg.norm(crosses[ax] / vol, axis=-1)
            reps.append(int(np.ceil(cutoff * inv_min).max()))
        else:
            reps.append(0)
    return reps


def radius_graph_pbc_oracle(pos: np.ndarray, n_per_graph: np.ndarray, pbc: Sequence[bool], cell: np.ndarray, cutoff: float):
    """Restatement of ``radius_graph_pbc`` (data/radius_graph.py:35-192): wrap into
    the cell (:111-116, wrap_positions :6-32), replicate over (2r+1)^3 images in
    cartesian_prod order (:93-97), keep pairs with 0.01 < D < cutoff (:124-125),
    emit center-major, then (neighbor*n_cells + cell) ascending (:177-181), and fix
    ``cell_offsets`` for the wrapping shift (:186-190).  Arithmetic is carried in
    the dtype of ``pos`` like the reference; pinned bit-exactly by
    tests/golden/radius_graph_pbc_*.npz."""
    dt = pos.dtype
    reps = _n_images(cell.astype(dt), pbc, cutoff)
    axes = [np.arange(-r, r + 1, dtype=dt) for r in reps]
    offs = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, 3)  # cartesian_prod order
    n_cells = offs.shape[0]
    ei0, ei1, cos = [], [], []
    begin = 0
    for g, n in enumerate(n_per_graph):
        n = int(n)
        p = pos[begin : begin + n]
        c = cell[g].astype(dt)
        frac = p @ np.linalg.inv(c)
        shift = np.zeros_like(frac)
        for ax in range(3):
            if pbc[ax]:
                shift[:, ax] = np.floor(frac[:, ax])
        frac = frac - shift
        pw = (frac @ c).astype(dt)
        img = (offs @ c).astype(dt)  # [n_cells,3]
        B = (pw[:, None, :] + img[None, :, :]).reshape(-1, 3)  # atom-major, cell-minor
        diff = pw[:, None, :] - B[None, :, :]
        D = np.sqrt((diff * diff).sum(-1))
        ix, iy = np.nonzero((D < cutoff) & (D > 0.01))
        c0, c1 = ix + begin, iy // n_cells + begin
        co = offs[iy % n_cells] + (shift[ix] - shift[iy // n_cells])
        ei0.append(c0)
        ei1.append(c1)
        cos.append(co)
        begin += n
    edge_index = np.stack([np.concatenate(ei0), np.concatenate(ei1)]).astype(np.int64)
    return edge_index, np.concatenate(cos).astype(dt)

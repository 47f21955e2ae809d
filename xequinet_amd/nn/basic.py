"""Edge geometry, property heads by autograd, embedding table and activation
factory -- host-side mirror of the reference's ``xequinet/nn/basic.py``."""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import keys, ops

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def get_embedding_tensor(embed_basis: str = "gfn2-xtb", aux_basis: str = "aux28") -> torch.Tensor:
    """Per-element embedding table [87, n_aux], row 0 = zeros (utils/qc.py:222-237).
    The table is numeric data shipped as ``data/{embed}_{aux}.npy`` (H..Rn, float64)."""
    path = os.path.join(_DATA, f"{embed_basis}_{aux_basis}.npy")
    if not os.path.exists(path):
        raise NotImplementedError(f"no precomputed embedding for {embed_basis}/{aux_basis} (pyscf generation is out of scope)")
    t = torch.from_numpy(np.load(path))
    t = torch.cat([torch.zeros(1, t.shape[-1], dtype=t.dtype), t])
    return t.to(torch.get_default_dtype())


class Int2c1eEmbedding(nn.Module):
    """nn/basic.py:34-57"""

    def __init__(self, embed_basis: str = "gfn2-xtb", aux_basis: str = "aux28") -> None:
        super().__init__()
        embed_ten = get_embedding_tensor(embed_basis, aux_basis)
        self.register_buffer("embed_ten", embed_ten)
        self.embed_dim = embed_ten.shape[1]

    def forward(self, at_no: torch.Tensor) -> torch.Tensor:
        return self.embed_ten[at_no.long()]


def edge_graph(data: Dict[str, torch.Tensor]) -> ops.EdgeGraph:
    """The destination-sorted CSR views of ``edge_index`` (built once per batch)."""
    g = data.get(keys.EDGE_GRAPH)
    ei = data[keys.EDGE_INDEX]
    if g is None or g.edge_index.data_ptr() != ei.contiguous().data_ptr() or g.n_edges != ei.shape[1]:
        g = ops.EdgeGraph(ei, data[keys.POSITIONS].shape[0], ptr=data.get(keys.BATCH_PTR))
        data[keys.EDGE_GRAPH] = g
    if g.ptr is None and keys.BATCH_PTR in data:
        g.ptr = data[keys.BATCH_PTR]
    return g


def compute_edge_data(
    data: Dict[str, torch.Tensor],
    compute_forces: bool = True,
    compute_virial: bool = False,
) -> Dict[str, torch.Tensor]:
    """Preprocess edge data (nn/basic.py:60-140) with the HIP edge-vector kernel."""
    pos = data[keys.POSITIONS]
    if keys.BATCH not in data:
        data[keys.BATCH] = torch.zeros(pos.shape[0], dtype=torch.long, device=pos.device)
        data[keys.BATCH_PTR] = torch.tensor([0, pos.shape[0]], dtype=torch.long, device=pos.device)
    if keys.BATCH_PTR in data:
        single_graph = data[keys.BATCH_PTR].numel() == 2
    else:
        single_graph = bool(data[keys.BATCH].max() == 0)
    n_graphs = data[keys.BATCH_PTR].numel() - 1 if keys.BATCH_PTR in data else int(data[keys.BATCH].max()) + 1

    has_cell = keys.CELL in data
    if compute_forces:
        pos.requires_grad_()
    if compute_virial:
        strain = torch.zeros((n_graphs, 3, 3), dtype=pos.dtype, device=pos.device)
    else:   # nobody differentiates with respect to it: a cached read-only zero tensor instead of a fill launch per evaluation
        from .fused import constant_vector

        strain = constant_vector(n_graphs * 9, 0.0, pos.dtype, pos.device).view(n_graphs, 3, 3)
    if compute_virial:
        # nn/basic.py:99-107 scales positions and cell by (1 + sym(strain)); every edge vector then scales the same
        # way, so the strain enters the edge-vector op (values at strain = 0 unchanged, gradient in its backward)
        strain.requires_grad_()

    graph = edge_graph(data)
    cell = data[keys.CELL] if has_cell else None
    cell_offsets = data[keys.CELL_OFFSETS].to(pos.dtype) if has_cell else None
    batch = None if (single_graph or not has_cell) else data[keys.BATCH]
    if compute_virial:
        vectors, dist = ops.EdgeVectors.apply(pos, graph, cell, cell_offsets, batch, strain, data[keys.BATCH_PTR])
    else:
        vectors, dist = ops.EdgeVectors.apply(pos, graph, cell, cell_offsets, batch)

    data.update({keys.EDGE_LENGTH: dist, keys.EDGE_VECTOR: vectors, keys.STRAIN: strain})
    return data


def _seed(energy: torch.Tensor, training: bool):
    """(grad_outputs, sign) of a force / virial evaluation.  The reference seeds the reverse pass with ones and negates the result
    (nn/basic.py:150-159).  Outside a training pass the seed is MINUS ones -- a cached constant, no fill launch -- and the reverse pass,
    linear in its seed, returns -dE/dx: one fill and one negation launch less per evaluation.  The bits are those of the negated +1
    result wherever the cotangents run through IEEE operations and the exact-f32 matrix instruction (symmetric in the sign: the message
    kernels, the small-system chain, every f64 kernel); the fused node block puts them through bf16 matrix products, which are NOT
    symmetric (profiles/r05_mfma_sign.txt: 1-2 results in 10 000 differ in the last bit), so there the forces differ from the
    reference's form in the last bits (<= 1e-6 on forces of order one) -- both fronts seed alike (csrc/xeq_torch.cpp::minus_one) and stay
    bit-identical to each other.  A training pass (create_graph) keeps the reference's form."""
    if training or not energy.is_cuda or energy.dim() != 1:
        return [torch.ones_like(energy)], -1.0
    from .fused import constant_vector

    return [constant_vector(energy.shape[0], -1.0, energy.dtype, energy.device)], None


def _grad(energy, inputs, training):
    seed, sign = _seed(energy, training)
    with ops.geometry_only_backward(energy):     # this reverse pass is asked for dE/dpos alone (ops.DiffMessage skips its parameter gradients)
        grads = torch.autograd.grad(outputs=[energy], inputs=inputs, grad_outputs=seed, retain_graph=training,
                                    create_graph=training, allow_unused=True)
    grads = [torch.zeros_like(x) if g is None else g for g, x in zip(grads, inputs)]
    return grads if sign is None else [sign * g for g in grads]


def compute_forces_only(energy: torch.Tensor, pos: torch.Tensor, training: bool = True) -> torch.Tensor:
    """nn/basic.py:143-159"""
    return _grad(energy, [pos], training)[0]


def compute_virial_only(energy: torch.Tensor, strain: torch.Tensor, training: bool = True) -> torch.Tensor:
    """nn/basic.py:162-178"""
    return _grad(energy, [strain], training)[0]


def compute_forces_and_virial(energy: torch.Tensor, pos: torch.Tensor, strain: torch.Tensor, training: bool = True):
    """nn/basic.py:181-199"""
    forces, virial = _grad(energy, [pos, strain], training)
    return forces, virial


def compute_properties(
    data: Dict[str, torch.Tensor],
    compute_forces: bool = True,
    compute_virial: bool = False,
    training: bool = True,
    extra_properties: Optional[List[str]] = None,
) -> Dict[str, torch.Tensor]:
    """nn/basic.py:202-238"""
    results = {}
    if compute_forces and compute_virial:
        results[keys.FORCES], results[keys.VIRIAL] = compute_forces_and_virial(
            energy=data[keys.TOTAL_ENERGY], pos=data[keys.POSITIONS], strain=data[keys.STRAIN], training=training)
    elif compute_virial:
        results[keys.VIRIAL] = compute_virial_only(energy=data[keys.TOTAL_ENERGY], strain=data[keys.STRAIN], training=training)
    elif compute_forces:
        results[keys.FORCES] = compute_forces_only(energy=data[keys.TOTAL_ENERGY], pos=data[keys.POSITIONS], training=training)
    # ops.EdgeGradDeferral finishes by count: if a registered message block's reverse pass never ran, every block returned None
    # and the forces above came back as zeros through `allow_unused`.  That must be an error, not a silent zero (round-5 advisor).
    graph = data.get(keys.EDGE_GRAPH)
    deferral = getattr(graph, "edge_grad_deferral", None) if graph is not None else None
    if deferral is not None and (compute_forces or compute_virial):
        if deferral.sets or deferral.sb_seen:
            n_left, deferral.sets = len(deferral.sets) + deferral.sb_seen, []
            deferral.sb_vec, deferral.sb_seen = None, 0
            raise RuntimeError(f"edge-gradient deferral not drained: {n_left} of {deferral.registered} registered message blocks "
                               "reached the reverse pass; dL/dvec was never emitted")
    if extra_properties is not None:
        results.update({k: data[k] for k in extra_properties})
    return results


def resolve_activation(activation: str, devide_x: bool = False) -> nn.Module:
    """nn/basic.py:241-262"""
    activation = activation.lower()
    activation_div_x = {"silu": "sigmoid", "relu": "identity", "leakyrelu": "identity"}
    if devide_x and activation in activation_div_x:
        activation = activation_div_x[activation]
    table = {"relu": nn.ReLU, "leakyrelu": nn.LeakyReLU, "softplus": nn.Softplus, "sigmoid": nn.Sigmoid,
             "silu": nn.SiLU, "tanh": nn.Tanh, "identity": nn.Identity}
    if activation not in table:
        raise NotImplementedError(f"Unsupported activation function {activation}")
    return table[activation]()

"""Per-irrep norm / dot / equivariant layer norm -- mirror of ``xequinet/nn/o3layer.py``
(same class names, constructor arguments, parameter and buffer names)."""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn

from .. import o3, ops


class Invariant(nn.Module):
    """Modulus of each irrep in a direct sum of irreps (nn/o3layer.py:12-44)."""

    def __init__(self, irreps_in: Iterable, squared: bool = False, eps: float = 1e-5) -> None:
        super().__init__()
        irreps_in = o3.Irreps(irreps_in).simplify()
        irreps_out = o3.Irreps([(mul, "0e") for mul, _ in irreps_in])
        instr = [(i, i, i, "uuu", False, ir.dim) for i, (mul, ir) in enumerate(irreps_in)]
        self.tp = o3.TensorProduct(irreps_in, irreps_in, irreps_out, instr, irrep_normalization="component")
        self.irreps_in = irreps_in
        self.irreps_out = irreps_out.simplify()
        self.squared = squared
        self.eps = eps

    def __repr__(self):
        return f"{self.__class__.__name__}({self.irreps_in})"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = self.tp(x, x)
        if self.squared:
            return out
        return torch.sqrt(out + self.eps**2) - self.eps


class EquivariantDot(nn.Module):
    """nn/o3layer.py:79-109"""

    def __init__(self, irreps_in: Iterable):
        super().__init__()
        irreps_in = o3.Irreps(irreps_in).simplify()
        irreps_out = o3.Irreps([(mul, "0e") for mul, _ in irreps_in])
        instr = [(i, i, i, "uuu", False, ir.dim) for i, (mul, ir) in enumerate(irreps_in)]
        self.tp = o3.TensorProduct(irreps_in, irreps_in, irreps_out, instr, irrep_normalization="component")
        self.irreps_in = irreps_in
        self.irreps_out = irreps_out.simplify()
        self.input_dim = self.irreps_in.dim

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.irreps_in})"

    def forward(self, features1: torch.Tensor, features2: torch.Tensor) -> torch.Tensor:
        assert (
            features1.shape[-1] == features2.shape[-1] == self.input_dim
        ), "Input tensor must have the same last dimension as the irreps"
        return self.tp(features1, features2)


class EquivariantLayerNorm(nn.Module):
    """nn/o3layer.py:112-171, one fused HIP kernel per direction."""

    def __init__(self, irreps, affine: bool = True, eps: float = 1e-5) -> None:
        super().__init__()
        self.irreps = o3.Irreps(irreps)
        self.dim = self.irreps.dim
        self.num_scalar = sum(mul for mul, ir in self.irreps if ir.l == 0 and ir.p == 1)
        self.num_features = self.irreps.num_irreps
        scalar_index = []
        ix = 0
        for mul, ir in self.irreps:
            if ir.l == 0 and ir.p == 1:
                scalar_index.extend(list(range(ix, ix + mul)))
            ix += ir.dim * mul
        self.register_buffer("scalar_index", torch.LongTensor(scalar_index))
        self._mul = self.irreps.mul3()
        if any(ir.l == 0 and ir.p != 1 for _, ir in self.irreps):
            raise NotImplementedError("EquivariantLayerNorm: 0o blocks are not supported by the HIP kernel")
        weight = torch.ones(self.num_features)
        bias = torch.zeros(self.num_scalar)
        if affine:
            self.affine_weight = nn.Parameter(weight)
            self.affine_bias = nn.Parameter(bias)
        else:
            self.register_buffer("affine_weight", weight)
            self.register_buffer("affine_bias", bias)
        self.eps = eps

    def forward(self, node_input: torch.Tensor) -> torch.Tensor:
        assert node_input.shape[-1] == self.dim, "Input tensor must have the same last dimension as the irreps"
        return ops.EqLayerNorm.apply(node_input, self.affine_weight, self.affine_bias, self._mul, self.eps)

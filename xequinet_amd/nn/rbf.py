"""Radial bases and envelopes -- mirror of ``xequinet/nn/rbf.py`` (same class names,
constructor arguments and parameter names), evaluated by ``xeq_radial_fwd``.

These modules only materialise ``rbf``/``fcut`` when a caller asks for them; the
fused message kernel recomputes both from the edge vector and owns the gradient,
so the outputs here are detached."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops


def resolve_rbf(rbf_kernel: str, num_basis: int, cutoff: float) -> nn.Module:
    if rbf_kernel == "bessel":
        return SphericalBesselj0(num_basis, cutoff)
    elif rbf_kernel == "gaussian":
        return GaussianSmearing(num_basis, cutoff)
    else:  # "expbern"/"expnorm" are broken or off-path in the reference
        raise NotImplementedError(f"rbf kernel {rbf_kernel} is not implemented")


def resolve_cutoff(cutoff_fn: str, cutoff: float, **kwargs) -> nn.Module:
    if cutoff_fn == "cosine":
        return CosineCutoff(cutoff)
    elif cutoff_fn == "polynomial":
        return PolynomialCutoff(cutoff, **kwargs)
    else:  # "exponential"/"flat" cannot be constructed in the reference (nn/rbf.py:76-91)
        raise NotImplementedError(f"cutoff function {cutoff_fn} is not implemented")


class CutoffFunction(nn.Module):
    kind = "cosine"

    def __init__(self, cutoff: float) -> None:
        super().__init__()
        self.cutoff = cutoff

    def forward(self, dist: torch.Tensor) -> torch.Tensor:
        assert dist.dim() == 2 and dist.size(1) == 1, "Distance tensor must be [nedge, 1]."
        _, fcut = ops.radial_basis(dist, "bessel", self.kind, 1, self.cutoff, None, want_rbf=False)
        return fcut.view(-1, 1)


class CosineCutoff(CutoffFunction):
    kind = "cosine"


class PolynomialCutoff(CutoffFunction):
    kind = "polynomial"

    def __init__(self, cutoff: float, order: int = 3) -> None:
        super().__init__(cutoff=cutoff)
        if order != 3:
            raise NotImplementedError("PolynomialCutoff: only the default order 3 is built")
        self.order = order


class GaussianSmearing(nn.Module):
    kind = "gaussian"

    def __init__(self, num_basis: int, cutoff: float, eps: float = 1e-5) -> None:
        super().__init__()
        self.num_basis = num_basis
        self.cutoff = cutoff
        self.eps = eps
        self.mean = torch.nn.Parameter(torch.linspace(0, cutoff, num_basis).view(1, -1))
        self.std = torch.nn.Parameter(torch.ones(num_basis).view(1, -1))

    def params(self):
        return self.mean, self.std

    def forward(self, dist: torch.Tensor) -> torch.Tensor:
        rbf, _ = ops.radial_basis(dist, "gaussian", "cosine", self.num_basis, self.cutoff, self.mean, self.std, want_fcut=False)
        return rbf


class SphericalBesselj0(nn.Module):
    kind = "bessel"

    def __init__(self, num_basis: int, cutoff: float, eps: float = 1e-5) -> None:
        super().__init__()
        self.num_basis = num_basis
        self.cutoff = cutoff
        freq = math.pi * torch.arange(1, num_basis + 1) / cutoff
        self.freq = torch.nn.Parameter(freq.view(1, -1))
        if eps != 1e-5:
            raise NotImplementedError("SphericalBesselj0: eps is fixed to the reference default 1e-5 in the kernels")
        self.eps = eps
        self.coeff = math.sqrt(2 / self.cutoff)

    def params(self):
        return self.freq, None

    def forward(self, dist: torch.Tensor) -> torch.Tensor:
        rbf, _ = ops.radial_basis(dist, "bessel", "cosine", self.num_basis, self.cutoff, self.freq, want_fcut=False)
        return rbf

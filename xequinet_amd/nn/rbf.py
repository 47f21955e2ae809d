"""Radial bases and envelopes -- mirror of ``xequinet/nn/rbf.py`` (same class names,
constructor arguments and parameter names), evaluated by ``xeq_radial_fwd``.

These modules only materialise ``rbf``/``fcut`` when a caller asks for them; the
fused message kernel recomputes both from the edge vector and owns the gradient,
so the outputs here are detached."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def resolve_rbf(rbf_kernel: str, num_basis: int, cutoff: float) -> nn.Module:
    if rbf_kernel == "bessel":
        return SphericalBesselj0(num_basis, cutoff)
    elif rbf_kernel == "gaussian":
        return GaussianSmearing(num_basis, cutoff)
    elif rbf_kernel == "expbern":   # nn/rbf.py:14-15: the reference hands the CUTOFF over as alpha; so does this
        return ExponentialBernstein(num_basis, cutoff)
    elif rbf_kernel == "expnorm":
        return ExponentialNorm(num_basis, cutoff)
    else:
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


def softplus_inverse(x) -> torch.Tensor:
    """nn/rbf.py:155-158"""
    if not isinstance(x, torch.Tensor):
        x = torch.tensor(x)
    return x + torch.log(-torch.expm1(-x))


class ExponentialBernstein(nn.Module):
    """nn/rbf.py:161-191: rho_k(d) = C(B-1, k) e^{-alpha d (B-1-k)} (1 - e^{-alpha d})^k in logarithms, alpha = softplus(_alpha).
    Same buffers (logc, n, v) and the same parameter (_alpha) as the reference, so its checkpoints load."""

    kind = "expbern"

    def __init__(self, num_basis: int, alpha: float = 0.5) -> None:
        super().__init__()
        self.num_basis = num_basis
        self.alpha = alpha
        dt = torch.get_default_dtype()
        k = torch.arange(num_basis, dtype=torch.float64)
        top = float(num_basis - 1)
        # log C(B - 1, k) through lgamma (the reference sums logarithms; the same numbers to 1e-15)
        logc = math.lgamma(top + 1.0) - torch.lgamma(k + 1.0) - torch.lgamma(top - k + 1.0)
        self.register_buffer("logc", logc.to(dt))          # buffer and parameter names: the reference's state dict
        self.register_buffer("n", (top - k).to(dt))
        self.register_buffer("v", k.to(dt))
        self._alpha = nn.Parameter(torch.tensor(1.0, dtype=dt))
        self.reset_parameters()
        self._p0 = None

    def reset_parameters(self) -> None:
        nn.init.constant_(self._alpha, softplus_inverse(self.alpha))

    def params(self):
        """(softplus(_alpha) once per basis function, logc): what the kernels take as p0 / p1 (include/xeq.h, XEQ_RBF_EXPBERN);
        the first is formed once per version of _alpha."""
        key = (self._alpha._version, self._alpha.data_ptr(), self._alpha.dtype)
        if self._p0 is None or self._p0[0] != key:
            with torch.no_grad():
                self._p0 = (key, F.softplus(self._alpha).reshape(1, 1).expand(1, self.num_basis).contiguous())
        return self._p0[1], self.logc.view(1, -1)

    def forward(self, dist: torch.Tensor) -> torch.Tensor:
        p0, p1 = self.params()
        rbf, _ = ops.radial_basis(dist, "expbern", "cosine", self.num_basis, 1.0, p0, p1, want_fcut=False)
        return rbf


class ExponentialNorm(nn.Module):
    """nn/rbf.py:194-207: rho_k(d) = exp(-beta_k (e^{-d} - mu_k)^2)."""

    kind = "expnorm"

    def __init__(self, num_basis: int, cutoff: float) -> None:
        super().__init__()
        self.num_basis = num_basis
        self.cutoff = cutoff
        inv_beta = torch.square(2 * (1 - math.exp(-cutoff)) / torch.arange(1, num_basis + 1))
        self.beta = torch.nn.Parameter(torch.reciprocal(inv_beta))
        self.mu = torch.nn.Parameter(torch.linspace(1, math.exp(-cutoff), num_basis))

    def params(self):
        return self.beta.view(1, -1), self.mu.view(1, -1)

    def forward(self, dist: torch.Tensor) -> torch.Tensor:
        rbf, _ = ops.radial_basis(dist, "expnorm", "cosine", self.num_basis, self.cutoff, self.beta, self.mu, want_fcut=False)
        return rbf

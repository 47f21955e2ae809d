"""CPU oracle of the general Clebsch-Gordan tensor product  --  TEST INFRASTRUCTURE ONLY (see xpainn_oracle.py's header).

Restates e3nn 0.5.1's ``o3.TensorProduct`` forward (its published definition: one einsum per instruction against the
real Wigner-3j table, 'component' irrep normalisation, 'element' path normalisation) for the instruction lists the
reference builds in ``nn/tp.py:20-107``; e3nn itself is absent from /root/reference and not installable: **parity
unpinned** against a live e3nn.  The real Wigner-D matrices below (for equivariance tests) are built independently of the
3j tables: from the angular-momentum generators carried to the real basis m = -l..l.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

N_ELEMENTS = {"uvw": lambda a, b: a * b, "uvu": lambda a, b: b, "uvv": lambda a, b: a, "uuw": lambda a, b: a,
              "uuu": lambda a, b: 1, "uvuv": lambda a, b: 1}
EINSUM = {"uvw": ("uvw,ijk,zui,zvj->zwk", "zuvw,ijk,zui,zvj->zwk"), "uvu": ("uv,ijk,zui,zvj->zuk", "zuv,ijk,zui,zvj->zuk"),
          "uvv": ("uv,ijk,zui,zvj->zvk", "zuv,ijk,zui,zvj->zvk"), "uuw": ("uw,ijk,zui,zuj->zwk", "zuw,ijk,zui,zuj->zwk"),
          "uuu": ("u,ijk,zui,zuj->zuk", "zu,ijk,zui,zuj->zuk"), "uvuv": ("uv,ijk,zui,zvj->zuvk", "zuv,ijk,zui,zvj->zuvk")}
NOWEIGHT = {"uvw": None, "uvu": "ijk,zui,zvj->zuk", "uvv": "ijk,zui,zvj->zvk", "uuw": None, "uuu": "ijk,zui,zuj->zuk",
            "uvuv": "ijk,zui,zvj->zuvk"}


def tensor_product(irreps1: Sequence, irreps2: Sequence, irreps_out: Sequence, instructions: Sequence, w3j, x: torch.Tensor,
                   y: torch.Tensor, weight: Optional[torch.Tensor], shared_weights: bool = True) -> torch.Tensor:
    """irreps*: lists of (mul, l); instructions: (i1, i2, io, mode, has_weight, path_weight); w3j(l1, l2, l3) -> table."""
    def offs(irr):
        o, out = 0, []
        for mul, l in irr:
            out.append(o)
            o += mul * (2 * l + 1)
        return out, o

    o1, d1 = offs(irreps1)
    o2, d2 = offs(irreps2)
    oo, do = offs(irreps_out)
    z = x.shape[0]
    out = torch.zeros(z, do, dtype=x.dtype)
    woff = 0
    for ins in instructions:
        i1, i2, io, mode, has_w = ins[:5]
        pw = ins[5] if len(ins) > 5 else 1.0
        (m1, l1), (m2, l2), (mo, l3) = irreps1[i1], irreps2[i2], irreps_out[io]
        xx = x[:, o1[i1]:o1[i1] + m1 * (2 * l1 + 1)].reshape(z, m1, 2 * l1 + 1)
        yy = y[:, o2[i2]:o2[i2] + m2 * (2 * l2 + 1)].reshape(z, m2, 2 * l2 + 1)
        C = w3j(l1, l2, l3).to(x.dtype)
        alpha = (2 * l3 + 1) / sum(N_ELEMENTS[k[3]](irreps1[k[0]][0], irreps2[k[1]][0]) for k in instructions if k[2] == io)
        coeff = math.sqrt(alpha * pw)
        if has_w:
            shape = {"uvw": (m1, m2, mo), "uvu": (m1, m2), "uvv": (m1, m2), "uuw": (m1, mo), "uuu": (m1,), "uvuv": (m1, m2)}[mode]
            n = int(np.prod(shape))
            if shared_weights:
                w = weight[woff:woff + n].reshape(shape)
                r = torch.einsum(EINSUM[mode][0], w, C, xx, yy)
            else:
                w = weight[:, woff:woff + n].reshape((z,) + shape)
                r = torch.einsum(EINSUM[mode][1], w, C, xx, yy)
            woff += n
        else:
            if mode == "uvw":
                r = torch.einsum("ijk,zui,zvj->zk", C, xx, yy).unsqueeze(1).expand(z, mo, 2 * l3 + 1)
            elif mode == "uuw":
                r = torch.einsum("ijk,zui,zuj->zk", C, xx, yy).unsqueeze(1).expand(z, mo, 2 * l3 + 1)
            else:
                r = torch.einsum(NOWEIGHT[mode], C, xx, yy)
        out[:, oo[io]:oo[io] + mo * (2 * l3 + 1)] += coeff * r.reshape(z, -1)
    return out


def _real_to_complex(l: int) -> np.ndarray:
    U = np.zeros((2 * l + 1, 2 * l + 1), dtype=complex)
    s2 = 1 / math.sqrt(2)
    for m in range(-l, l + 1):
        if m < 0:
            U[m + l, l + abs(m)] = s2
            U[m + l, l - abs(m)] = -1j * s2
        elif m == 0:
            U[l, l] = 1
        else:
            U[m + l, l + m] = (-1) ** m * s2
            U[m + l, l - m] = 1j * (-1) ** m * s2
    return U


def generators_real(l: int) -> np.ndarray:
    """The three real antisymmetric generators of rotations about x, y, z on the real basis of degree l (m = -l..l)."""
    J = np.zeros((3, 2 * l + 1, 2 * l + 1), dtype=complex)
    for m in range(-l, l + 1):
        J[2, m + l, m + l] = m
        if m < l:
            c = math.sqrt(l * (l + 1) - m * (m + 1))
            J[0, m + 1 + l, m + l] += 0.5 * c
            J[1, m + 1 + l, m + l] += -0.5j * c
            J[0, m + l, m + 1 + l] += 0.5 * c
            J[1, m + l, m + 1 + l] += 0.5j * c
    U = _real_to_complex(l)
    X = np.stack([(U.conj().T @ (-1j * J[a]) @ U) for a in range(3)])
    assert np.abs(X.imag).max() < 1e-12
    return X.real


def wigner_D_real(l: int, w: Sequence[float]) -> np.ndarray:
    """D^l of the rotation exp(w . L) (axis-angle vector w) on the real basis m = -l..l."""
    from scipy.linalg import expm

    X = generators_real(l)
    # with the phase convention of _real_to_complex the real generators are (-L_x, +L_y, -L_z) of the Cartesian rotation
    # exp(w . L) below (checked at l = 1 against rotation_matrix in tests/test_tp.py): flip the two signs
    sgn = (-1.0, 1.0, -1.0)
    return expm(sum(sg * wi * X[a] for a, (wi, sg) in enumerate(zip(w, sgn))))


def rotation_matrix(w: Sequence[float]) -> np.ndarray:
    """The 3 x 3 rotation exp(w . L) in Cartesian (x, y, z)."""
    from scipy.linalg import expm

    wx, wy, wz = w
    return expm(np.array([[0, -wz, wy], [wz, 0, -wx], [-wy, wx, 0]], dtype=float))

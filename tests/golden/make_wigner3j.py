"""Generate the real Wigner-3j tables shipped as xequinet_amd/data/wigner3j_lmax4.npz  (run by hand; numpy only).

e3nn 0.5.1 (environment.yaml:139) is absent from /root/reference and not installable here, so the tables are built from the
published definitions: complex Clebsch-Gordan coefficients by Racah's formula, carried to the REAL spherical-harmonics
basis the path uses (m = -l..l; sin-type components for m < 0, cos-type for m > 0; SURVEY 8a row a5 shows that e3nn's Y_1,
Y_2 in the reference's ORIGINAL axis order are exactly these), normalised to unit Frobenius norm like e3nn's
``o3.wigner_3j``.  The overall SIGN of each (l1, l2, l3) table is a convention e3nn fixes in code that cannot be run here
(parity unpinned): this script makes the first non-zero entry in (i, j, k) lexicographic order positive.

Checks run before writing: every table is invariant under the real Wigner-D matrices of random rotations (1e-12),
l1 x l2 -> 0 is the scaled dot product and 1 x 1 -> 1 the Levi-Civita tensor / sqrt(6).
"""
import math
import os
import sys

import numpy as np

LMAX = 4


def _f(n):
    return math.factorial(n)


def clebsch_gordan(j1, m1, j2, m2, j3, m3):
    """<j1 m1 j2 m2 | j3 m3>, Racah's formula (integer spins)."""
    if m1 + m2 != m3 or not (abs(j1 - j2) <= j3 <= j1 + j2):
        return 0.0
    pref = math.sqrt((2 * j3 + 1) * _f(j3 + j1 - j2) * _f(j3 - j1 + j2) * _f(j1 + j2 - j3) / _f(j1 + j2 + j3 + 1))
    pref *= math.sqrt(_f(j3 + m3) * _f(j3 - m3) * _f(j1 - m1) * _f(j1 + m1) * _f(j2 - m2) * _f(j2 + m2))
    s = 0.0
    for k in range(0, j1 + j2 - j3 + 1):
        d = [k, j1 + j2 - j3 - k, j1 - m1 - k, j2 + m2 - k, j3 - j2 + m1 + k, j3 - j1 - m2 + k]
        if min(d) < 0:
            continue
        s += (-1) ** k / np.prod([float(_f(x)) for x in d])
    return pref * s


def real_to_complex(l):
    """U with Y_complex[m] = sum_m' U[m, m'] Y_real[m']  (Condon-Shortley phase; rows / columns m = -l..l)."""
    U = np.zeros((2 * l + 1, 2 * l + 1), dtype=complex)
    s2 = 1 / math.sqrt(2)
    for m in range(-l, l + 1):
        if m < 0:
            U[m + l, l + abs(m)] = s2                      # cos-type component |m|
            U[m + l, l - abs(m)] = -1j * s2                # sin-type component
        elif m == 0:
            U[l, l] = 1
        else:
            U[m + l, l + m] = (-1) ** m * s2
            U[m + l, l - m] = 1j * (-1) ** m * s2
    return U


def wigner_3j_real(l1, l2, l3):
    C = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            m3 = m1 + m2
            if abs(m3) <= l3:
                C[m1 + l1, m2 + l2, m3 + l3] = clebsch_gordan(l1, m1, l2, m2, l3, m3)
    U1, U2, U3 = real_to_complex(l1), real_to_complex(l2), real_to_complex(l3)
    # complex-basis invariant:  Y3[m3] ~ sum C[m1,m2,m3] Y1[m1] Y2[m2];  real basis: contract with U1, U2 and conj(U3)
    T = np.einsum("abc,ai,bj,ck->ijk", C.astype(complex), U1, U2, U3.conj())
    re, im = np.abs(T.real).max(), np.abs(T.imag).max()
    T = T.real if re >= im else T.imag                    # real for even l1+l2+l3, imaginary for odd
    assert min(re, im) < 1e-12 * max(re, im, 1e-300) or min(re, im) < 1e-14
    T = T / np.linalg.norm(T)
    nz = np.flatnonzero(np.abs(T) > 1e-12)
    if T.flat[nz[0]] < 0:
        T = -T
    T[np.abs(T) < 1e-14] = 0.0
    return T


def generators_real(l):
    """Real antisymmetric generators (X_x, X_y, X_z) of the rotation group on the real basis of degree l."""
    J = np.zeros((3, 2 * l + 1, 2 * l + 1), dtype=complex)
    for m in range(-l, l + 1):
        J[2, m + l, m + l] = m
        if m < l:
            c = math.sqrt(l * (l + 1) - m * (m + 1))
            J[0, m + 1 + l, m + l] += 0.5 * c          # J+ = Jx + i Jy
            J[1, m + 1 + l, m + l] += -0.5j * c
            J[0, m + l, m + 1 + l] += 0.5 * c          # J-
            J[1, m + l, m + 1 + l] += 0.5j * c
    U = real_to_complex(l)
    X = np.stack([(U.conj().T @ (-1j * J[a]) @ U) for a in range(3)])
    assert np.abs(X.imag).max() < 1e-12
    return X.real


def wigner_D_real(l, axis_angle):
    """D^l(R) on the real basis for the rotation exp(sum_a w_a L_a)."""
    from scipy.linalg import expm

    X = generators_real(l)
    return expm(sum(w * X[a] for a, w in enumerate(axis_angle)))


def main():
    rng = np.random.default_rng(0)
    tables = {}
    for l1 in range(LMAX + 1):
        for l2 in range(LMAX + 1):
            for l3 in range(abs(l1 - l2), min(LMAX, l1 + l2) + 1):
                T = wigner_3j_real(l1, l2, l3)
                for _ in range(3):
                    w = rng.normal(size=3)
                    D1, D2, D3 = (wigner_D_real(l, w) for l in (l1, l2, l3))
                    T2 = np.einsum("ijk,ai,bj,ck->abc", T, D1, D2, D3)
                    assert np.abs(T2 - T).max() < 1e-11, (l1, l2, l3, np.abs(T2 - T).max())
                tables[f"{l1}_{l2}_{l3}"] = T
    eps = np.zeros((3, 3, 3))
    for a, b, c in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        eps[a, b, c], eps[a, c, b] = 1, -1
    # real l = 1 components are (y, z, x): the Levi-Civita tensor in that order is the same tensor (cyclic relabelling)
    assert np.allclose(np.abs(tables["1_1_1"]), np.abs(eps) / math.sqrt(6))
    for l in range(LMAX + 1):
        assert np.allclose(tables[f"{l}_{l}_0"][:, :, 0], np.eye(2 * l + 1) / math.sqrt(2 * l + 1) * np.sign(tables[f"{l}_{l}_0"][0, 0, 0]))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "xequinet_amd", "data",
                       f"wigner3j_lmax{LMAX}.npz")
    np.savez_compressed(out, **tables)
    print("wrote", out, len(tables), "tables")


if __name__ == "__main__":
    sys.exit(main())

"""Seeded synthetic workloads of SURVEY 8d (pure numpy; no structure files ship with the reference).

Owned by the product: ``bench.py``, ``__graft_entry__.smoke()`` and the tests all draw their inputs here, so that
``oracle/`` is purely the checker.  ``WORKLOADS`` / ``make_workload`` name the BASELINE.json configurations.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np


def synth_molecule(rng: np.random.Generator, n: int) -> Tuple[np.ndarray, np.ndarray]:
    """Chain-growth geometry: atom k at U(1.0,1.55) A in a random
    direction from a random earlier atom, rejected if < 0.95 A from any atom."""
    pos = np.zeros((n, 3))
    k = 1
    while k < n:
        parent = rng.integers(0, k)
        v = rng.normal(size=3)
        v /= np.linalg.norm(v)
        cand = pos[parent] + v * rng.uniform(1.0, 1.55)
        if np.all(np.linalg.norm(pos[:k] - cand, axis=1) >= 0.95):
            pos[k] = cand
            k += 1
    nh = (n + 1) // 2
    z = np.ones(n, dtype=np.int32)
    z[:nh] = rng.choice([6, 7, 8, 9], size=nh, p=[0.72, 0.12, 0.15, 0.01])
    return pos, z


def synth_qm9_batch(n_mol: int, seed: int = 1234):
    """QM9-shape batch: n = clip(round(N(18,3)),3,29) atoms per molecule."""
    rng = np.random.default_rng(seed)
    P, Z, ptr = [], [], [0]
    for _ in range(n_mol):
        n = int(np.clip(np.rint(rng.normal(18, 3)), 3, 29))
        p, z = synth_molecule(rng, n)
        P.append(p)
        Z.append(z)
        ptr.append(ptr[-1] + n)
    return np.concatenate(P), np.concatenate(Z), np.asarray(ptr, dtype=np.int64)


def synth_aspirin(seed: int = 7):
    """Aspirin-shaped C9H8O4: chain-growth geometry, 21 atoms."""
    rng = np.random.default_rng(seed)
    pos, _ = synth_molecule(rng, 21)
    z = np.array([6] * 9 + [8] * 4 + [1] * 8, dtype=np.int32)
    return pos, z, np.asarray([0, 21], dtype=np.int64)


def synth_water_box(n_side: int = 8, seed: int = 5):
    """Bulk-water-density cubic box: n_side^3 molecules, 0.0334 / A^3."""
    rng = np.random.default_rng(seed)
    nmol = n_side**3
    L = (nmol / 0.0334) ** (1.0 / 3.0)
    a = L / n_side
    grid = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3)
    O = (grid + 0.5) * a + rng.uniform(-0.25, 0.25, size=(nmol, 3))
    pos, z = [], []
    half = math.radians(104.5) / 2
    for o in O:
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        w = np.cross(u, rng.normal(size=3))
        w /= np.linalg.norm(w)
        h1 = o + 0.96 * (math.cos(half) * u + math.sin(half) * w)
        h2 = o + 0.96 * (math.cos(half) * u - math.sin(half) * w)
        pos += [o, h1, h2]
        z += [8, 1, 1]
    cell = np.eye(3) * L
    return np.asarray(pos), np.asarray(z, dtype=np.int32), np.asarray([0, 3 * nmol], dtype=np.int64), cell[None]


def synth_md17_frames(n_frames: int = 4096, seed: int = 11, sigma: float = 0.05):
    """MD17-aspirin trajectory batch: the aspirin geometry + iid N(0, sigma A) displacement per frame."""
    p0, z0, _ = synth_aspirin()
    rng = np.random.default_rng(seed)
    n = p0.shape[0]
    pos = (p0[None] + rng.normal(0, sigma, size=(n_frames, n, 3))).reshape(-1, 3)
    return pos, np.tile(z0, n_frames), np.arange(0, (n_frames + 1) * n, n, dtype=np.int64)


WORKLOADS = {
    "qm9_1024": "1024 QM9-shape synthetic molecules",
    "qm9_64": "64 QM9-shape synthetic molecules",
    "qm9_8192": "8192 QM9-shape synthetic molecules (the per-GPU share of QM9-65k on 8 GPUs, SURVEY 8d-5)",
    "qm9_65536": "65536 QM9-shape synthetic molecules, ONE batch sharded by molecule over the GPUs (SURVEY 8d-5)",
    "qm9_8192_sharded": "8192 QM9-shape synthetic molecules, ONE batch sharded by molecule over the GPUs (rehearsal size of SURVEY 8d-5)",
    "md17_4096": "4096 perturbed aspirin frames (MD17 shape)",
    "water_512": "one periodic box of 512 water molecules",
}


def make_workload(name: str, seed: int = 1234):
    """(pos[N,3] f64, z[N] i32, ptr[G+1] i64, cell[1,3,3] | None) of a named workload."""
    if name.startswith("qm9_"):
        pos, z, ptr = synth_qm9_batch(int(name.split("_")[1]), seed=seed)   # ("qm9_8192_sharded": the same batch as "qm9_8192")
        return pos, z, ptr, None
    if name == "md17_4096":
        pos, z, ptr = synth_md17_frames(4096, seed=11 + seed)
        return pos, z, ptr, None
    if name == "water_512":
        pos, z, ptr, cell = synth_water_box(8, seed=5 + seed)
        return pos, z, ptr, cell
    raise ValueError(name)

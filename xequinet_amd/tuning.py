"""Library-GEMM selection for the dense node-side contractions.

The dense contractions of the path (scalar_mlp, update_mlp, o3.Linear, dot_lin, energy MLP: SURVEY 8a rows a9, a14,
a15) are plain library GEMMs on hipBLASLt / rocBLAS.  Their default heuristics pick slow kernels for the skinny shapes
of this model ([N, 32..576] x [32..576]): on QM9-1024 the 54 GEMMs of one evaluation take 1.48 ms by default and
1.05 ms with each shape timed once against the libraries' candidate kernels.  `enable_gemm_autotune()` switches that
selection on through PyTorch's TunableOp (a one-off cost of ~50 ms per distinct GEMM shape at its first call, i.e.
during warm-up); the chosen kernels are fp32 GEMMs of the same libraries, so results stay within fp32 rounding.
"""
from __future__ import annotations

import os

import torch

import contextlib

__all__ = ["enable_gemm_autotune", "gemm_autotune_enabled", "gemm_autotune_scope"]


def enable_gemm_autotune(max_tuning_ms: int = 100, results_file: str | None = None) -> bool:
    """Time every new GEMM shape once and keep the fastest library kernel.  Returns False (and does nothing) when no
    HIP device is present.  With `results_file` naming an existing file of an earlier run, its selections are
    replayed and nothing is timed (what a profiled run wants: no tuning launches in the trace)."""
    if not torch.cuda.is_available():
        return False
    t = torch.cuda.tunable
    t.enable(True)
    replay = results_file is not None and os.path.exists(results_file) and os.path.getsize(results_file) > 0
    t.tuning_enable(not replay)
    t.set_max_tuning_duration(int(max_tuning_ms))
    if results_file is None:
        # per-process scratch file: nothing is written into the caller's working directory
        results_file = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"xeq_tunableop_{os.getpid()}.csv")
    t.set_filename(results_file, insert_device_ordinal=False)
    if replay:
        t.read_file(results_file)
    return True


def gemm_autotune_enabled() -> bool:
    return torch.cuda.is_available() and torch.cuda.tunable.is_enabled()


@contextlib.contextmanager
def gemm_autotune_scope(active: bool = True):
    """Time new GEMM shapes inside the block only, then put PyTorch's process-wide TunableOp switches back the way the
    host application had them (a library must not leave every other GEMM of the process tuned at ~50 ms per shape)."""
    if not active or not torch.cuda.is_available() or gemm_autotune_enabled():
        yield
        return
    t = torch.cuda.tunable
    was_enabled, was_tuning = t.is_enabled(), t.tuning_is_enabled()
    enable_gemm_autotune()
    try:
        yield
    finally:
        # the lookup stays on (a selection is only applied while TunableOp is enabled: the captured graph and later
        # eager calls use the kernels timed here), timing of NEW shapes goes back to what the application had: off
        t.tuning_enable(was_tuning if was_enabled else False)

"""``torch_scatter``-style entry points on the path (nn/output.py:7,73,124)."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


def scatter(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size: Optional[int] = None,
            reduce: str = "sum", ptr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``torch_scatter.scatter(src, index, dim=0, reduce='sum')``.

    With ``ptr`` (the CSR form of a sorted ``index``, e.g. PyG's ``batch``/``ptr``)
    the reduction is a deterministic segmented sum with a gradient; without it the
    forward-only atomic kernel is used."""
    if dim != 0:
        raise NotImplementedError("scatter: only dim=0 is on the XPaiNN path")
    if reduce not in ("sum", "add"):
        raise NotImplementedError(f"scatter: reduce={reduce!r} is not on the XPaiNN path")
    if ptr is not None:
        return ops.SegmentSum.apply(src, ptr, index if (index is not None and index.dim() == 1) else None)
    if src.requires_grad:
        raise NotImplementedError("scatter without ptr is forward-only; pass ptr for a differentiable segmented sum")
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    return ops.scatter_add(src, index, dim_size)


def scatter_sum(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size: Optional[int] = None,
                ptr: Optional[torch.Tensor] = None) -> torch.Tensor:
    return scatter(src, index, dim=dim, dim_size=dim_size, reduce="sum", ptr=ptr)

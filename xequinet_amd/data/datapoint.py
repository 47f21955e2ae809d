"""Minimal batch container with the attribute surface the hot path reads from the
reference's ``XequiData`` / PyG ``Batch`` (data/datapoint.py:7-115): ``pos``,
``atomic_numbers``, ``ptr``, ``batch``, ``num_graphs``, optional ``pbc``/``cell``,
``edge_index``/``cell_offsets``; ``.to(device)`` and ``.to_dict()``."""
from __future__ import annotations

from typing import Dict, Optional

import torch


class XequiBatch:
    _TENSORS = ("pos", "atomic_numbers", "ptr", "batch", "pbc", "cell", "edge_index", "cell_offsets")

    def __init__(self, pos: torch.Tensor, atomic_numbers: torch.Tensor, ptr: Optional[torch.Tensor] = None,
                 pbc: Optional[torch.Tensor] = None, cell: Optional[torch.Tensor] = None,
                 edge_index: Optional[torch.Tensor] = None, cell_offsets: Optional[torch.Tensor] = None) -> None:
        assert pos.dim() == 2 and pos.shape[1] == 3
        assert atomic_numbers.shape[0] == pos.shape[0]
        n = pos.shape[0]
        if ptr is None:
            ptr = torch.tensor([0, n], dtype=torch.int64, device=pos.device)
        ptr = ptr.to(torch.int64)
        self.pos = pos
        self.atomic_numbers = atomic_numbers.to(torch.int32)  # data/datapoint.py:52-55
        self.ptr = ptr
        counts = ptr[1:] - ptr[:-1]
        # output_size: no device-to-host read-back for the length
        self.batch = torch.repeat_interleave(torch.arange(counts.numel(), device=ptr.device), counts, output_size=n)
        self.num_graphs = int(ptr.numel() - 1)
        if (pbc is None) != (cell is None):
            raise ValueError("PBC and cell must be both defined or both undefined.")
        if pbc is not None:
            self.pbc = pbc.reshape(-1, 3)
            self.cell = cell.reshape(-1, 3, 3)
        self.edge_index = edge_index
        self.cell_offsets = cell_offsets

    def to(self, device) -> "XequiBatch":
        for k in self._TENSORS:
            v = getattr(self, k, None)
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        return self

    def to_dict(self) -> Dict[str, torch.Tensor]:
        out = {}
        for k in self._TENSORS:
            v = getattr(self, k, None)
            if isinstance(v, torch.Tensor):
                out[k] = v
        for k in ("_xeq_edge_graph",):
            if hasattr(self, k):
                out[k] = getattr(self, k)
        return out

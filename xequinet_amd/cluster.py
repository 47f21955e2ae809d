"""``torch_cluster.radius_graph`` drop-in (call site data/transform.py:58-64)."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


def radius_graph(x: torch.Tensor, r: float, batch: Optional[torch.Tensor] = None, loop: bool = False,
                 max_num_neighbors: int = 32, flow: str = "source_to_target", num_workers: int = 1,
                 batch_size: Optional[int] = None, ptr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All same-graph pairs with ``d^2 < r^2`` (strict), no self loops.

    Differences from torch_cluster, by design: neighbours are never truncated
    (the reference passes ``max_num_neighbors = sum n_g^2``, i.e. unlimited,
    data/transform.py:57) and the result is emitted in canonical order, sorted by
    (row 0, row 1) -- the edge set is symmetric, so this is a re-ordering of
    torch_cluster's result and is the destination-sorted layout the fused message
    kernel consumes directly.  ``batch`` must be sorted (PyG batches are);
    ``ptr`` may be given instead to skip a device->host sync."""
    if loop:
        raise NotImplementedError("radius_graph(loop=True) is not used by the reference")
    if flow != "source_to_target":
        raise NotImplementedError("radius_graph: only flow='source_to_target' is used by the reference")
    n = x.shape[0]
    if ptr is None:
        if batch is None:
            ptr = torch.tensor([0, n], dtype=torch.int64, device=x.device)
        else:
            n_graphs = batch_size if batch_size is not None else (int(batch.max()) + 1 if n else 0)
            ptr = ops.csr_rowptr(batch.to(torch.int64).contiguous(), n_graphs).to(torch.int64)
    edge_index, _ = ops.radius_graph_raw(x, ptr, r)
    return edge_index

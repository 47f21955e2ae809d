"""``NeighborTransform`` -- mirror of ``xequinet/data/transform.py:21-69``."""
from __future__ import annotations

import os

import torch

from .. import keys, ops
from .radius_graph import radius_graph_pbc


class NeighborTransform:
    def __init__(self, cutoff: float) -> None:
        self.cutoff = cutoff

    def __call__(self, data):
        device = data.pos.device
        num_graphs = data.num_graphs if hasattr(data, keys.NUM_GRAPHS) else 1
        if num_graphs > 1:
            assert hasattr(data, keys.BATCH)
            ptr = data.ptr
        else:
            ptr = torch.tensor([0, data.pos.shape[0]], dtype=torch.int64, device=device)

        has_pbc = hasattr(data, keys.PBC) and bool(data.pbc.any())
        has_cell = hasattr(data, keys.CELL)

        if has_pbc and has_cell:
            if getattr(data, "edge_index", None) is not None and getattr(data, "cell_offsets", None) is not None:
                return data
            n_nodes_per_graph = ptr[1:] - ptr[:-1]     # (only the periodic search takes the counts)
            edge_index, cell_offsets = radius_graph_pbc(
                pos=data.pos, n_nodes_per_graph=n_nodes_per_graph, cell=data.cell, pbc=data.pbc, cutoff=self.cutoff, ptr=ptr,
            )
            data.edge_index = edge_index
            data.cell_offsets = cell_offsets
        elif not has_pbc and not has_cell:
            if getattr(data, "edge_index", None) is not None:
                return data
            # unlimited neighbours, like max_num_neighbors = sum n_g^2 (data/transform.py:57); same result as
            # cluster.radius_graph, whose row pointer is kept: the list is symmetric with ascending neighbours, so the
            # model's neighbour-sorted view is the reverse-edge map (no sort)
            data.edge_index, rowptr = ops.radius_graph_raw(data.pos, ptr, self.cutoff)
            setattr(data, keys.EDGE_GRAPH, ops.EdgeGraph(data.edge_index, data.pos.shape[0], center_sorted=True, ptr=ptr,
                                                         c_rowptr=rowptr, symmetric=os.environ.get("XEQ_REVERSE_EDGE_MAP", "1") != "0"))
            return data
        else:
            raise ValueError("PBC and cell must be both defined or both undefined.")
        # center-sorted edges in (neighbor, image) order, every edge with its mirror image (up to a rounding at the cutoff): hand the CSR
        # view and the mirror map to the model (ops.EdgeGraph: no sort by neighbor, the reverse pass walks the forward plan)
        setattr(data, keys.EDGE_GRAPH, ops.EdgeGraph(data.edge_index, data.pos.shape[0], center_sorted=True, ptr=ptr, symmetric=True,
                                                     cell_offsets=data.cell_offsets))
        return data

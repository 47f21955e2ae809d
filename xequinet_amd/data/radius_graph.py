"""Periodic neighbour list -- mirror of ``xequinet/data/radius_graph.py``.

Host-side preparation follows the reference line by line in meaning (image counts
:61-89, image table :93-104, wrapping :6-32/:111-116) using torch ops on the
device, so that the per-pair arithmetic the HIP search kernel sees is the
reference's; the O(n^2 n_cells) cdist/nonzero search and the Python loop over
graphs (:118-181) are replaced by ``xeq_radius_graph_pbc_{count,fill}``."""
from __future__ import annotations

from typing import List, Tuple

import torch

from .. import ops


def wrap_positions(pos: torch.Tensor, cell: torch.Tensor, n_nodes_per_graph: torch.Tensor, pbc: List[bool]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Wrap positions into the unit cell (data/radius_graph.py:6-32)."""
    if not any(pbc):
        return pos, torch.zeros_like(pos)
    cell_per_atom = cell.repeat_interleave(n_nodes_per_graph, dim=0)
    cell_inv = torch.linalg.inv(cell_per_atom)
    fractional = torch.bmm(pos.unsqueeze(1), cell_inv).squeeze(1)
    shift = torch.zeros_like(pos)
    for i, periodic in enumerate(pbc):
        if periodic:
            shift[:, i] = torch.floor(fractional[:, i])
    fractional = fractional - shift
    pos_wrap = torch.bmm(fractional.unsqueeze(1), cell_per_atom).squeeze(1)
    return pos_wrap, shift


_PRUNE_MARGIN = 1e-3  # on |f_a - n_a|: far above the rounding of f, far below any lattice spacing
_CELL_LIST_MIN_ATOMS = 6144  # average atoms per graph from which the bin grid replaces the O(n_g^2) pair sweep (water boxes on
                             # MI355X, scratch/nl_bench.py: 3 993 atoms 0.43 vs 0.53 ms, 8 232 atoms 0.72 vs 0.65 ms)
_CELL_LIST_MAX_BINS = 64     # per axis


def _use_cell_list(n_atoms: int, n_graphs: int) -> bool:
    import os

    flag = os.environ.get("XEQ_PBC_CELL_LIST")
    if flag is not None:
        return flag not in ("0", "", "false")
    return n_atoms >= _CELL_LIST_MIN_ATOMS * max(1, n_graphs)


def _with_bins(prune, pbc: List[bool]):
    """prune + bins per axis: floor(1 / thr_a) on periodic axes (bin width >= cutoff across lattice planes), 1 on open ones."""
    recip, thr, reps = prune
    nb = torch.floor(1.0 / thr).clamp_(1, _CELL_LIST_MAX_BINS).to(torch.int32)
    for ax in range(3):
        if not pbc[ax]:
            nb[:, ax] = 1
    return recip, thr, reps, nb


def _image_counts(cell: torch.Tensor, pbc: List[bool], cutoff: float, with_prune: bool = False):
    """Images per axis (data/radius_graph.py:61-89): ceil(rc * |a_j x a_k| / V), max over the batch.
    with_prune: also the reciprocal rows recip[G,3,3] = a_j x a_k / V and thr[G,3] = rc |recip_a| + margin of the
    image-pruned kernels (a pair can only be within rc through images n with |f_a - n_a| <= rc |recip_a|)."""
    cross_a2a3 = torch.cross(cell[:, 1], cell[:, 2], dim=-1)
    cell_vol = torch.sum(cell[:, 0] * cross_a2a3, dim=-1, keepdim=True)
    crosses = [cross_a2a3, torch.cross(cell[:, 2], cell[:, 0], dim=-1), torch.cross(cell[:, 0], cell[:, 1], dim=-1)]
    reps, inv = [], []
    for ax in range(3):
        inv_min_dist = torch.norm(crosses[ax] / cell_vol, p=2, dim=-1)
        inv.append(inv_min_dist)
        if pbc[ax]:
            reps.append(torch.ceil(cutoff * inv_min_dist).max())
        else:
            reps.append(cell.new_zeros(()))
    reps = [int(v) for v in torch.stack(reps).tolist()]  # one host sync (the reference does three .item())
    if not with_prune:
        return reps
    recip = torch.stack([c / cell_vol for c in crosses], dim=1)           # [G, 3(axis), 3]
    thr = cutoff * torch.stack(inv, dim=1) + _PRUNE_MARGIN                 # [G, 3]
    return reps, (recip, thr, reps)


def _host_cell_tables(cell: torch.Tensor, pbc_: List[bool], cutoff: float, with_inverse: bool):
    """Everything the periodic search derives from the cells alone, formed ON THE HOST in one round trip and uploaded as one
    buffer: image counts per axis (data/radius_graph.py:61-89), the image table and its Cartesian offsets per graph (:93-104),
    the reciprocal rows / thresholds of the image-pruned kernels and (for wrapping, :6-32) the inverse cells.  The cells are
    9 numbers per graph; as device tensor operations this was ~40 launches and three round trips (0.5 ms in front of a 0.07 ms
    search).  The tables come from the C ABI's host functions (``xeq_pbc_image_counts`` / ``xeq_pbc_tables_host``: the cells'
    own dtype, every operation rounded once, in the reference's order of operations) -- the same functions the registered
    operator ``xeq::radius_graph_pbc`` calls, so the scripted and the Python models search with the same bits.

    -> reps [3] (ints), n_cells, dict of device tensors: cell_offsets [n_cells, 3], pbc_offsets [G, n_cells, 3], recip [G, 3, 3],
       thr [G, 3], cell_inv [G, 3, 3] (or None)."""
    import numpy as np

    c = np.ascontiguousarray(cell.detach().cpu().numpy())             # the round trip
    reps, n_cells, flat_h = host_cell_tables_np(c, pbc_, cutoff, with_inverse, dtype_code_of=cell)
    flat = torch.from_numpy(flat_h).to(cell.device)                   # one upload
    return reps, n_cells, split_cell_tables(flat, c.shape[0], n_cells, with_inverse)


def host_cell_tables_np(c, pbc_: List[bool], cutoff: float, with_inverse: bool, dtype_code_of: torch.Tensor):
    """The host half of _host_cell_tables on a numpy copy of the cells [G, 3, 3]: -> (reps [3], n_cells, flat host array).  The
    flat array is what is uploaded (split_cell_tables names its pieces); a caller that keeps the tables in static device memory
    (runtime.GraphedStepPBC) copies it there when the cell changes."""
    import ctypes

    import numpy as np

    from ..lib import call, dtype_code, mul3

    dt = c.dtype
    G = c.shape[0]
    code = dtype_code(dtype_code_of)
    reps_c = (ctypes.c_int32 * 3)()
    call("xeq_pbc_image_counts", code, ctypes.c_void_p(c.ctypes.data), G, mul3([int(v) for v in pbc_]), float(cutoff), reps_c)
    reps = [int(v) for v in reps_c]
    n_cells = (2 * reps[0] + 1) * (2 * reps[1] + 1) * (2 * reps[2] + 1)
    n_tab = (3 + 3 * G) * n_cells + 12 * G
    flat_h = np.empty(n_tab + (9 * G if with_inverse else 0), dtype=dt)
    call("xeq_pbc_tables_host", code, ctypes.c_void_p(c.ctypes.data), G, reps_c, float(cutoff), ctypes.c_void_p(flat_h.ctypes.data), n_tab)
    if with_inverse:
        flat_h[n_tab:] = np.linalg.inv(c).astype(dt).ravel()
    return reps, n_cells, flat_h


def split_cell_tables(flat: torch.Tensor, G: int, n_cells: int, with_inverse: bool):
    """Views of the uploaded table buffer: cell_offsets [n_cells, 3], pbc_offsets [G, n_cells, 3], recip [G, 3, 3], thr [G, 3],
    cell_inv [G, 3, 3] (or None)."""
    import numpy as np

    out, o = {}, 0
    for name, shape in (("cell_offsets", (n_cells, 3)), ("pbc_offsets", (G, n_cells, 3)), ("recip", (G, 3, 3)), ("thr", (G, 3)),
                        ("cell_inv", (G, 3, 3))):
        if name == "cell_inv" and not with_inverse:
            out[name] = None
            continue
        n = int(np.prod(shape))
        out[name] = flat[o:o + n].view(shape)
        o += n
    return out


def _wrap_on_device(pos: torch.Tensor, ptr: torch.Tensor, cell: torch.Tensor, cell_inv: torch.Tensor, pbc_: List[bool]):
    """wrap_positions in one launch (xeq_pbc_wrap)."""
    from ..lib import call, dtype_code, mul3, ptr as p_, stream

    pos = pos.detach().contiguous()
    pos_wrap, shift = torch.empty_like(pos), torch.empty_like(pos)
    call("xeq_pbc_wrap", dtype_code(pos), p_(pos), p_(ptr), ptr.numel() - 1, pos.shape[0], p_(cell.contiguous()), p_(cell_inv),
         mul3([int(v) for v in pbc_]), p_(pos_wrap), p_(shift), stream())
    return pos_wrap, shift


@torch.no_grad()
def radius_graph_pbc(pos: torch.Tensor, n_nodes_per_graph: torch.Tensor, pbc: torch.Tensor, cell: torch.Tensor,
                     cutoff: float, return_rowptr: bool = False, ptr: torch.Tensor = None):
    """Same signature and outputs as the reference (:35-192): ``edge_index`` [2,E] int64
    center-major, then (neighbor * n_cells + cell) ascending; ``cell_offsets`` [E,3].  ``ptr``: the graph pointer when
    the caller has it (it is the cumulative sum of ``n_nodes_per_graph``)."""
    ops.lib.require_hip(pos, cell)
    device, dtype = pos.device, pos.dtype
    batch_size = n_nodes_per_graph.shape[0]
    assert pbc.dim() == 2 and pbc.shape[1] == 3, "Invalid pbc shape"
    pbc_cpu = pbc.detach().cpu()
    assert torch.all(pbc_cpu[0] == pbc_cpu), "PBC must be the same for all graphs"
    pbc_ = pbc_cpu[0].tolist()
    cell = cell.to(dtype)

    max_rep, n_cells, tab = _host_cell_tables(cell, pbc_, cutoff, with_inverse=any(pbc_))
    if ptr is None:
        ptr = torch.zeros(batch_size + 1, dtype=torch.int64, device=device)
        ptr[1:] = torch.cumsum(n_nodes_per_graph.to(device), dim=0)
    ptr = ptr.to(torch.int64).contiguous()
    if any(pbc_):
        pos_wrap, shift = _wrap_on_device(pos, ptr, cell, tab["cell_inv"], pbc_)
    else:
        pos_wrap, shift = pos, torch.zeros_like(pos)
    prune = (tab["recip"], tab["thr"], max_rep)
    if _use_cell_list(pos.shape[0], batch_size):
        prune = _with_bins(prune, pbc_)
    edge_index, offsets, rowptr = ops.radius_graph_pbc_raw(pos_wrap, ptr, tab["pbc_offsets"], tab["cell_offsets"], shift, cutoff, prune=prune)
    if return_rowptr:
        return edge_index, offsets, rowptr
    return edge_index, offsets


def single_radius_graph(pos: torch.Tensor, pbc: torch.Tensor, cell: torch.Tensor, cutoff: float, return_rowptr: bool = False):
    """Single-graph variant (:195-275): no wrapping, cell [3,3], pbc [3].  ``return_rowptr``: also the CSR row pointer of the
    (center-sorted) list, which the search has anyway."""
    ops.lib.require_hip(pos, cell)
    device = pos.device
    pbc_ = [bool(v) for v in pbc.detach().cpu().tolist()]
    max_rep, n_cells, tab = _host_cell_tables(cell.to(pos.dtype).unsqueeze(0), pbc_, cutoff, with_inverse=False)
    ptr = torch.tensor([0, pos.shape[0]], dtype=torch.int64, device=device)
    # positions are NOT wrapped here (:195-275), so the bin grid (fractional coordinates in [0, 1)) does not apply
    edge_index, offsets, rowptr = ops.radius_graph_pbc_raw(pos.detach(), ptr, tab["pbc_offsets"], tab["cell_offsets"], torch.zeros_like(pos),
                                                           cutoff, prune=(tab["recip"], tab["thr"], max_rep))
    if return_rowptr:
        return edge_index, offsets, rowptr
    return edge_index, offsets

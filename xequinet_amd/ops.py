"""Autograd-aware Python front of the C ABI (include/xeq.h).

Every op here launches a hand-written HIP kernel from libxeq_hip.so on the
current HIP stream.  Backward passes are explicit HIP kernels too (first order
only: the force evaluation ``-dE/dpos`` of nn/basic.py:143-159 in eval mode; a
training pass -- parameter gradients, ``create_graph=True`` -- takes the differentiable form of
the blocks in nn/training.py and does not come through these ops).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import lib
from .lib import call, dtype_code, mul3, ptr, require_hip, stream


class _KernelTimer:
    """Optional HIP-event timing of the fused message kernels (bench.py's roofline leg).
    Events are recorded on torch's current stream, which is the stream the kernels are
    launched on (lib.stream()).  Disabled by default: zero overhead."""

    def __init__(self) -> None:
        self.enabled = False
        self.events = {}

    def reset(self, enabled: bool) -> None:
        self.enabled = enabled
        self.events = {}

    def launch(self, name: str, *args, label: Optional[str] = None) -> None:
        """``label``: the key the launch is booked under (default: the entry point's name)."""
        if not self.enabled:
            call(name, *args)
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        call(name, *args)
        b.record()
        self.events.setdefault(label or name, []).append((a, b))

    def summary(self):
        torch.cuda.synchronize()
        return {k: {"launches": len(v), "total_ms": sum(a.elapsed_time(b) for a, b in v)} for k, v in self.events.items()}


KERNEL_TIMER = _KernelTimer()


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.contiguous()


# --------------------------------------------------------------------------- graph
_SCAN_BYTES = {}


_SCAN_ONE_LAUNCH_MAX = None


def _scan_one_launch_max() -> int:
    global _SCAN_ONE_LAUNCH_MAX
    if _SCAN_ONE_LAUNCH_MAX is None:
        _SCAN_ONE_LAUNCH_MAX = int(lib.load().xeq_rowptr_from_degrees_max())
    return _SCAN_ONE_LAUNCH_MAX


def _exclusive_scan(deg: torch.Tensor, n: int, rowptr: torch.Tensor) -> None:
    """rowptr[0..n] = exclusive prefix sum of deg[0..n) (rowptr[n] = total): one launch of one workgroup up to 36 864 entries, the
    grid-wide scan (xeq_exclusive_scan_i32_ws) above."""
    if n <= _scan_one_launch_max():      # one workgroup, one launch (csrc/xeq_graph.hip: k_rowptr_from_degrees); the library's scan is two
        call("xeq_rowptr_from_degrees", ptr(deg), n, -1, ptr(rowptr), None, None, stream())
        return
    nbytes = _SCAN_BYTES.get(n)
    if nbytes is None:      # a pure function of n: asked once per size (the neighbour lists sit in front of every evaluation)
        nbytes = _SCAN_BYTES[n] = lib.load().xeq_exclusive_scan_i32_workspace(n)
        if len(_SCAN_BYTES) > 4096:
            _SCAN_BYTES.clear()
    work = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=deg.device)
    call("xeq_exclusive_scan_i32_ws", ptr(deg), n, ptr(rowptr), ptr(work), int(nbytes), stream())


def csr_by_key(keys: torch.Tensor, n_rows: int, n_valid: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(rowptr[n_rows+1], perm[E]) of an unsorted int64 index row: stable radix sort + row pointer in one library call.
    ``n_valid`` (int32 device tensor, first element): the keys are a capacity-sized array whose first n_valid entries count; the
    rest sorts behind every row (xeq_csr_by_key_bounded) -- no size is read back."""
    require_hip(keys)
    keys = keys.contiguous()
    n = keys.numel()
    nbytes = lib.load().xeq_csr_by_key_workspace(n, n_rows + (0 if n_valid is None else 1))
    if nbytes < 0:
        raise ValueError("csr_by_key: sizes out of range")
    work = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=keys.device)
    rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=keys.device)
    perm = torch.empty(n, dtype=torch.int32, device=keys.device)
    if n_valid is None:
        call("xeq_csr_by_key", ptr(keys), n, n_rows, ptr(work), int(nbytes), ptr(rowptr), ptr(perm), stream())
    else:
        assert n_valid.dtype == torch.int32 and n_valid.is_cuda and n_valid.numel() >= 1
        call("xeq_csr_by_key_bounded", ptr(keys), n, n_rows, ptr(n_valid), ptr(work), int(nbytes), ptr(rowptr), ptr(perm), stream())
    return rowptr, perm


def csr_rowptr(sorted_keys: torch.Tensor, n_rows: int) -> torch.Tensor:
    require_hip(sorted_keys)
    rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=sorted_keys.device)
    call("xeq_csr_rowptr", ptr(sorted_keys), sorted_keys.numel(), n_rows, ptr(rowptr), stream())
    return rowptr


class EdgeGraph:
    """Destination-sorted views of ``edge_index`` shared by all message blocks.

    ``c_*``: CSR over centers (edge_index[0], keys.py:16) -- forward aggregation.
    ``n_*``: CSR over neighbors (edge_index[1], keys.py:17) -- reverse pass.
    ``perm`` is None when the edge list is already sorted by that row.
    Index plumbing is one library call per unsorted row (``xeq_csr_by_key``: stable radix sort +
    row pointer); a sorted row only needs ``xeq_csr_rowptr`` (or the builder's own ``c_rowptr``).
    ``symmetric=True`` is the promise of the open-boundary neighbour-list builders of this package: center-sorted,
    neighbours ascending and unique per center, (i, j) present iff (j, i) is.  Then the neighbour-sorted order is
    the reverse-edge map (``xeq_reverse_edge_map``, one binary search per edge) and needs no sort; the result is
    the same permutation the stable sort gives.
    ``symmetric=True`` WITH ``cell_offsets`` is the promise of this package's PERIODIC builders (radius_graph_pbc*, runtime.GraphedStepPBC):
    center-sorted, a center's edges ascending in (neighbor, image), and (i, j, o) present iff (j, i, -o) is -- up to a rounding at the
    cutoff (include/xeq.h, xeq_reverse_edge_map_pbc).  Then ``mirror_map`` (position of the mirror edge, -1 where the list holds none)
    lets the wq reverse kernel walk the forward plan as for an open system: no sort, no second plan, no second set of records.  It is
    not a permutation, so the neighbor-sorted view proper (``n_rowptr`` / ``n_perm``) is built only when somebody asks for it (the sb /
    generic kernel families, the training kernels).  XEQ_PBC_MIRROR=0 switches the periodic mirror map off.
    """

    def __init__(self, edge_index: torch.Tensor, n_nodes: int, center_sorted: Optional[bool] = None,
                 ptr: Optional[torch.Tensor] = None, c_rowptr: Optional[torch.Tensor] = None,
                 symmetric: bool = False, capacity_form: bool = False, cell_offsets: Optional[torch.Tensor] = None) -> None:
        require_hip(edge_index)
        self.ptr = ptr          # graph boundaries [G+1] (keys.BATCH_PTR), when the caller knows them
        self._wq = None
        self._basis = None   # per-edge records of the current geometry (edge_basis)
        assert edge_index.dim() == 2 and edge_index.shape[0] == 2 and edge_index.dtype == torch.int64
        self.edge_index = edge_index = edge_index.contiguous()
        self.n_nodes = int(n_nodes)
        self.n_edges = E = int(edge_index.shape[1])
        center, nbr = edge_index[0], edge_index[1]
        if center_sorted is None:
            center_sorted = bool((center[1:] >= center[:-1]).all()) if E > 1 else True
        if center_sorted:
            self.c_perm = None
            if c_rowptr is not None:
                assert c_rowptr.dtype == torch.int32 and c_rowptr.numel() == self.n_nodes + 1
                self.c_rowptr = c_rowptr.contiguous()
            else:
                self.c_rowptr = csr_rowptr(center, self.n_nodes)
        else:
            self.c_rowptr, self.c_perm = csr_by_key(center, self.n_nodes)
        import os

        periodic_mirror = bool(symmetric and center_sorted and cell_offsets is not None and os.environ.get("XEQ_PBC_MIRROR", "1") != "0")
        exact_mirror = bool(symmetric and center_sorted and cell_offsets is None)
        # symmetric, center-sorted list: the reverse wq kernel walks the FORWARD plan (every slot stands for its mirror edge) and
        # mirror_map tells where a slot's edge gradient goes (include/xeq.h, XEQ_WQ_MIRROR_WALK)
        self.mirror_walk = exact_mirror or periodic_mirror
        # the KIND of list, fixed at construction (runtime.GraphedModel re-creates a list of the same kind for its capture; what has
        # been built lazily since -- the sorted view -- says nothing about it)
        self.periodic = cell_offsets is not None
        self.cell_offsets = cell_offsets
        self.mirror_map: Optional[torch.Tensor] = None
        # set by callers whose edge_index is a capacity-sized buffer (runtime.GraphedStep*, train.GraphedTrainStep): the true edge count
        # is c_rowptr[N] on the device; kernels that walk edges by index rather than by row pointer are handed that pointer
        self.edge_count_on_device = False
        self._capacity_form = bool(capacity_form)
        self._n_view = None                      # (n_rowptr, n_perm), built on demand
        if capacity_form:
            assert center_sorted and c_rowptr is not None
        if exact_mirror:
            rev = torch.empty(E, dtype=torch.int32, device=edge_index.device)
            call("xeq_reverse_edge_map", lib.ptr(edge_index), E, self.n_nodes, lib.ptr(self.c_rowptr), lib.ptr(rev), stream())
            self.mirror_map = rev
            self._n_view = (self.c_rowptr, rev)   # a permutation: the stable sort by neighbor gives the same one
        elif periodic_mirror:
            off = cell_offsets.contiguous()
            assert off.shape == (E, 3) and off.is_floating_point()
            rev = torch.empty(E, dtype=torch.int32, device=edge_index.device)
            call("xeq_reverse_edge_map_pbc", dtype_code(off), lib.ptr(edge_index), lib.ptr(off), E, self.n_nodes, lib.ptr(self.c_rowptr),
                 lib.ptr(rev), stream())
            self.mirror_map = rev
        else:
            self._sorted_neighbor_view()

    def _sorted_neighbor_view(self):
        if self._n_view is None:
            nbr = self.edge_index[1]
            if self._capacity_form:
                # edge_index is a capacity-sized buffer, the edge count sits in c_rowptr[N] on the device (radius_graph_pbc_capacity):
                # the neighbor-sorted view skips the slots behind it
                self._n_view = csr_by_key(nbr, self.n_nodes, n_valid=self.c_rowptr[self.n_nodes:])
            else:
                self._n_view = csr_by_key(nbr, self.n_nodes)
        return self._n_view

    @property
    def n_rowptr(self) -> torch.Tensor:
        return self._sorted_neighbor_view()[0]

    @property
    def n_perm(self) -> torch.Tensor:
        return self._sorted_neighbor_view()[1]

    def reverse_view(self):
        """(row pointer, edge per slot) of the edges by NEIGHBOR for kernels that only sum over them (xeq_edge_vectors_bwd): the mirror map
        over the center rows where there is one (a missing mirror, -1, is skipped), else the sorted view."""
        if self.mirror_map is not None:
            return self.c_rowptr, self.mirror_map
        return self._sorted_neighbor_view()

    def wq_plan(self, reverse: bool, edges_per_stream: int = 128):
        """Walk plan of the wave / quad message kernels (xeq_message_wq_plan): every node's edge list padded to whole
        quads (four slots), per padded slot the gathered node and the edge id, per quad the owner and its first / last
        flags, stream boundaries on quads.  Depends on the graph only, not on the positions."""
        key = (bool(reverse), int(edges_per_stream))
        if self._wq is None:
            self._wq = {}
        plan = self._wq.get(key)
        if plan is None:
            L = lib.load()
            N, E, dev = self.n_nodes, self.n_edges, self.edge_index.device
            pcap = int(L.xeq_message_wq_pcap(N, E))
            n_ranges = max(1, -(-E // (2 * edges_per_stream)))
            i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)
            plan = {"reverse": bool(reverse), "n_ranges": n_ranges, "pcap": pcap, "qptr": i32(N + 1), "pgath": i32(pcap),
                    "peid": i32(pcap), "qinfo": i32(pcap // 4), "sq": i32(2 * n_ranges + 1), "sn": i32(2 * n_ranges + 1),
                    "win": i32(int(L.xeq_message_wq_win_ints(n_ranges))),
                    "work": torch.empty(max(int(L.xeq_message_wq_plan_workspace(N)), 1), dtype=torch.uint8, device=dev),
                    "records": None}
            self._wq[key] = plan
            self._build_wq_plan(plan)
        return plan

    def _build_wq_plan(self, plan) -> None:
        rev = plan["reverse"]
        rowptr, perm = (self.n_rowptr, self.n_perm) if rev else (self.c_rowptr, self.c_perm)
        owner, gather = (self.edge_index[1], self.edge_index[0]) if rev else (self.edge_index[0], self.edge_index[1])
        plan["rowptr"] = rowptr
        call("xeq_message_wq_plan", ptr(rowptr), ptr(perm), ptr(owner), ptr(gather), self.n_nodes, self.n_edges, plan["n_ranges"],
             ptr(plan["work"]), plan["work"].numel(), ptr(plan["qptr"]), ptr(plan["pgath"]), ptr(plan["peid"]), ptr(plan["qinfo"]),
             ptr(plan["sq"]), ptr(plan["sn"]), ptr(plan["win"]), stream())

    def refresh_plans(self) -> None:
        """Recompute the cached stream tables / walk plans in place after the CSR arrays were overwritten (HIP-graph
        replay: same sizes, new contents)."""
        for plan in (self._wq or {}).values():
            self._build_wq_plan(plan)



_CELL_LIST_MIN_ATOMS = 512   # average atoms per graph from which the bin grid replaces the O(n_g^2) sweep
_CELL_LIST_MAX_BINS = 64     # per axis


def _use_cell_list(n_atoms: int, n_graphs: int) -> bool:
    import os

    flag = os.environ.get("XEQ_CELL_LIST")
    if flag is not None:
        return flag not in ("0", "", "false")
    return n_atoms >= _CELL_LIST_MIN_ATOMS * max(1, n_graphs)


def _radius_graph_cell_list(pos, ptr_, cutoff):
    """Open-boundary neighbour list through a per-graph bin grid (xeq_radius_graph_*_cl): O(N) instead of O(n_g^2)."""
    N, G = pos.shape[0], ptr_.numel() - 1
    dev, dt = pos.device, dtype_code(pos)
    counts = ptr_[1:] - ptr_[:-1]
    gidx = torch.repeat_interleave(torch.arange(G, device=dev), counts, output_size=N).unsqueeze(1).expand(N, 3)
    lo = torch.full((G, 3), float("inf"), dtype=pos.dtype, device=dev).scatter_reduce_(0, gidx, pos, "amin", include_self=True)
    hi = torch.full((G, 3), float("-inf"), dtype=pos.dtype, device=dev).scatter_reduce_(0, gidx, pos, "amax", include_self=True)
    ext = (hi - lo).clamp_(min=0).nan_to_num_(nan=0.0, posinf=0.0, neginf=0.0)      # empty graphs: one bin
    lo = lo.nan_to_num_(nan=0.0, posinf=0.0, neginf=0.0)
    nb = torch.floor(ext / (cutoff * (1.0 + 1e-4))).clamp_(1, _CELL_LIST_MAX_BINS).to(torch.int32)   # bin width >= cutoff
    inv_w = (nb.to(pos.dtype) / ext.clamp(min=1e-30)).contiguous()
    bin_base = torch.zeros(G + 1, dtype=torch.int32, device=dev)
    bin_base[1:] = torch.cumsum(nb.prod(dim=1), 0)
    n_bins = int(bin_base[-1].item())
    keys_ = torch.empty(N, dtype=torch.int64, device=dev)
    lo, nb = lo.contiguous(), nb.contiguous()
    call("xeq_radius_graph_bin_ids", dt, ptr(pos), ptr(ptr_), G, N, ptr(lo), ptr(inv_w), ptr(nb), ptr(bin_base), ptr(keys_), stream())
    bin_start, bin_atom = csr_by_key(keys_, n_bins)
    deg = torch.empty(N, dtype=torch.int32, device=dev)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    call("xeq_radius_graph_count_cl", dt, ptr(pos), ptr(ptr_), G, N, float(cutoff), ptr(lo), ptr(inv_w), ptr(nb), ptr(bin_base),
         ptr(bin_start), ptr(bin_atom), ptr(deg), stream())
    _exclusive_scan(deg, N, rowptr)
    E = int(rowptr[-1].item()) if N > 0 else 0
    edge_index = torch.empty((2, E), dtype=torch.int64, device=dev)
    tmp_keys = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    call("xeq_radius_graph_fill_cl", dt, ptr(pos), ptr(ptr_), G, N, float(cutoff), ptr(lo), ptr(inv_w), ptr(nb), ptr(bin_base),
         ptr(bin_start), ptr(bin_atom), ptr(rowptr), E, ptr(tmp_keys), ptr(edge_index), stream())
    return edge_index, rowptr


def radius_graph_raw(pos: torch.Tensor, ptr_: torch.Tensor, cutoff: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """Non-PBC neighbour list; returns (edge_index[2,E] sorted by (center, neighbor), rowptr[N+1])."""
    require_hip(pos, ptr_)
    pos = pos.detach().contiguous()
    ptr_ = ptr_.to(torch.int64).contiguous()
    N, G = pos.shape[0], ptr_.numel() - 1
    if N > 0 and G > 0 and _use_cell_list(N, G):
        return _radius_graph_cell_list(pos, ptr_, cutoff)
    dev = pos.device
    deg = torch.empty(N, dtype=torch.int32, device=dev)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    dt = dtype_code(pos)
    call("xeq_radius_graph_count", dt, ptr(pos), ptr(ptr_), G, N, float(cutoff), ptr(deg), stream())
    _exclusive_scan(deg, N, rowptr)
    E = int(rowptr[-1].item()) if N > 0 else 0  # one host sync, as in the reference's nonzero()
    edge_index = torch.empty((2, E), dtype=torch.int64, device=dev)
    call("xeq_radius_graph_fill", dt, ptr(pos), ptr(ptr_), G, N, float(cutoff), ptr(rowptr), E, ptr(edge_index), stream())
    return edge_index, rowptr


def radius_graph_capacity(pos: torch.Tensor, ptr_: torch.Tensor, cutoff: float, edge_index: torch.Tensor, running_total: Optional[torch.Tensor] = None):
    """Non-PBC neighbour list into a caller-owned ``edge_index`` [2, capacity] WITHOUT reading the edge count back: returns
    (row pointer [N + 1], count [1]) on the device.  ``count`` is the true edge count; a list that outgrew the capacity is replaced
    by an EMPTY one (row pointer all zeros: no kernel walks past a buffer, and the symmetric shortcuts downstream never see a cut,
    asymmetric list) -- ``count > capacity`` tells the caller, who reads it next to the results.  Slots behind the count keep their old contents, which must
    be valid node ids (zero-initialise the buffer once).  Every kernel downstream bounds its walk by the row pointer, so the
    whole evaluation can sit in one captured HIP graph (runtime.GraphedStep).  ``running_total`` (int64 [1], optional) += the true count,
    inside the same launch.  The pair sweep only (graphs of many atoms go
    through the cell list, whose bin count is a host value)."""
    require_hip(pos, ptr_, edge_index)
    pos = pos.detach().contiguous()
    ptr_ = ptr_.to(torch.int64).contiguous()
    assert edge_index.dim() == 2 and edge_index.shape[0] == 2 and edge_index.dtype == torch.int64 and edge_index.is_contiguous()
    N, G, cap = pos.shape[0], ptr_.numel() - 1, int(edge_index.shape[1])
    dev = pos.device
    deg = torch.empty(N, dtype=torch.int32, device=dev)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    dt = dtype_code(pos)
    call("xeq_radius_graph_count", dt, ptr(pos), ptr(ptr_), G, N, float(cutoff), ptr(deg), stream())
    count = torch.empty(1, dtype=torch.int32, device=dev)
    if N <= _scan_one_launch_max():     # scan, total and capacity guard by one workgroup in one launch (three before)
        call("xeq_rowptr_from_degrees", ptr(deg), N, cap, ptr(rowptr), ptr(count), ptr(running_total), stream())
    else:
        raw = torch.empty(N + 1, dtype=torch.int32, device=dev)
        _exclusive_scan(deg, N, raw)
        call("xeq_rowptr_guard", ptr(raw), N, cap, ptr(rowptr), ptr(count), stream())
        if running_total is not None:
            running_total.add_(count)
    call("xeq_radius_graph_fill", dt, ptr(pos), ptr(ptr_), G, N, float(cutoff), ptr(rowptr), cap, ptr(edge_index), stream())
    return rowptr, count


def radius_graph_pbc_raw(pos_wrap, ptr_, img, cells, shift, cutoff, prune=None):
    """PBC neighbour search over precomputed images (see xeq_radius_graph_pbc_* in xeq.h).
    prune = (recip[G,3,3], thr[G,3], reps) selects the image-pruned kernels (same edges, same order)."""
    require_hip(pos_wrap, ptr_, img, cells, shift)
    pos_wrap, img, cells, shift = (t.contiguous() for t in (pos_wrap, img, cells, shift))
    if prune is not None and len(prune) == 4:
        # cell list (graphs of many atoms): bins in fractional coordinates, see xeq_radius_graph_pbc_*_cl
        recip, thr, reps, nbins = prune
        recip, thr = recip.to(pos_wrap.dtype).contiguous(), thr.to(pos_wrap.dtype).contiguous()
        nbins = nbins.to(torch.int32).contiguous()
        ptr_ = ptr_.to(torch.int64).contiguous()
        N, G, n_cells = pos_wrap.shape[0], ptr_.numel() - 1, cells.shape[0]
        dev, dt = pos_wrap.device, dtype_code(pos_wrap)
        per_graph = nbins.prod(dim=1)
        bin_base = torch.zeros(G + 1, dtype=torch.int32, device=dev)
        bin_base[1:] = torch.cumsum(per_graph, 0)
        n_bins = int(bin_base[-1].item())
        keys_ = torch.empty(N, dtype=torch.int64, device=dev)
        call("xeq_radius_graph_pbc_bin_ids", dt, ptr(pos_wrap), ptr(ptr_), G, N, ptr(recip), ptr(nbins), ptr(bin_base), ptr(keys_),
             stream())
        bin_start, bin_atom = csr_by_key(keys_, n_bins)
        deg = torch.empty(N, dtype=torch.int32, device=dev)
        rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        call("xeq_radius_graph_pbc_count_cl", dt, ptr(pos_wrap), ptr(ptr_), G, N, ptr(img), n_cells, float(cutoff), ptr(recip),
             ptr(thr), mul3(reps), ptr(nbins), ptr(bin_base), ptr(bin_start), ptr(bin_atom), ptr(deg), stream())
        _exclusive_scan(deg, N, rowptr)
        E = int(rowptr[-1].item()) if N > 0 else 0
        edge_index = torch.empty((2, E), dtype=torch.int64, device=dev)
        cell_offsets = torch.empty((E, 3), dtype=pos_wrap.dtype, device=dev)
        tmp_keys = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
        call("xeq_radius_graph_pbc_fill_cl", dt, ptr(pos_wrap), ptr(ptr_), G, N, ptr(img), ptr(cells), ptr(shift), n_cells,
             float(cutoff), ptr(recip), ptr(thr), mul3(reps), ptr(nbins), ptr(bin_base), ptr(bin_start), ptr(bin_atom), ptr(rowptr),
             E, ptr(tmp_keys), ptr(edge_index), ptr(cell_offsets), stream())
        return edge_index, cell_offsets, rowptr
    if prune is not None:
        recip, thr, reps = prune
        recip, thr = recip.to(pos_wrap.dtype).contiguous(), thr.to(pos_wrap.dtype).contiguous()
        ptr_ = ptr_.to(torch.int64).contiguous()
        N, G, n_cells = pos_wrap.shape[0], ptr_.numel() - 1, cells.shape[0]
        dev, dt = pos_wrap.device, dtype_code(pos_wrap)
        deg = torch.empty(N, dtype=torch.int32, device=dev)
        rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        call("xeq_radius_graph_pbc_count_pruned", dt, ptr(pos_wrap), ptr(ptr_), G, N, ptr(img), n_cells, float(cutoff), ptr(recip),
             ptr(thr), mul3(reps), ptr(deg), stream())
        _exclusive_scan(deg, N, rowptr)
        E = int(rowptr[-1].item()) if N > 0 else 0
        edge_index = torch.empty((2, E), dtype=torch.int64, device=dev)
        cell_offsets = torch.empty((E, 3), dtype=pos_wrap.dtype, device=dev)
        call("xeq_radius_graph_pbc_fill_pruned", dt, ptr(pos_wrap), ptr(ptr_), G, N, ptr(img), ptr(cells), ptr(shift), n_cells,
             float(cutoff), ptr(recip), ptr(thr), mul3(reps), ptr(rowptr), E, ptr(edge_index), ptr(cell_offsets), stream())
        return edge_index, cell_offsets, rowptr
    ptr_ = ptr_.to(torch.int64).contiguous()
    N, G, n_cells = pos_wrap.shape[0], ptr_.numel() - 1, cells.shape[0]
    dev = pos_wrap.device
    dt = dtype_code(pos_wrap)
    deg = torch.empty(N, dtype=torch.int32, device=dev)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    call("xeq_radius_graph_pbc_count", dt, ptr(pos_wrap), ptr(ptr_), G, N, ptr(img), n_cells, float(cutoff), ptr(deg), stream())
    _exclusive_scan(deg, N, rowptr)
    E = int(rowptr[-1].item()) if N > 0 else 0
    edge_index = torch.empty((2, E), dtype=torch.int64, device=dev)
    cell_offsets = torch.empty((E, 3), dtype=pos_wrap.dtype, device=dev)
    call("xeq_radius_graph_pbc_fill", dt, ptr(pos_wrap), ptr(ptr_), G, N, ptr(img), ptr(cells), ptr(shift), n_cells,
         float(cutoff), ptr(rowptr), E, ptr(edge_index), ptr(cell_offsets), stream())
    return edge_index, cell_offsets, rowptr


def radius_graph_pbc_capacity(pos_wrap, ptr_, img, cells, shift, cutoff, prune, edge_index: torch.Tensor, cell_offsets: torch.Tensor):
    """The image-pruned periodic search of radius_graph_pbc_raw into caller-owned ``edge_index`` [2, capacity] / ``cell_offsets``
    [capacity, 3] WITHOUT reading the edge count back: returns (row pointer [N + 1], count [1]) on the device; a count above the
    capacity means the list was cut (the row pointer is cut with it) -- the caller checks it when it reads its results.  Same edges in the same
    order as the sized form.  The pair sweep only (the bin grid's size is a host value)."""
    require_hip(pos_wrap, ptr_, img, cells, shift, edge_index, cell_offsets)
    recip, thr, reps = prune
    pos_wrap, img, cells, shift = (t.contiguous() for t in (pos_wrap, img, cells, shift))
    recip, thr = recip.to(pos_wrap.dtype).contiguous(), thr.to(pos_wrap.dtype).contiguous()
    ptr_ = ptr_.to(torch.int64).contiguous()
    assert edge_index.dim() == 2 and edge_index.shape[0] == 2 and edge_index.dtype == torch.int64 and edge_index.is_contiguous()
    N, G, n_cells, cap = pos_wrap.shape[0], ptr_.numel() - 1, cells.shape[0], int(edge_index.shape[1])
    assert cell_offsets.shape == (cap, 3) and cell_offsets.dtype == pos_wrap.dtype and cell_offsets.is_contiguous()
    dev, dt = pos_wrap.device, dtype_code(pos_wrap)
    deg = torch.empty(N, dtype=torch.int32, device=dev)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    call("xeq_radius_graph_pbc_count_pruned", dt, ptr(pos_wrap), ptr(ptr_), G, N, ptr(img), n_cells, float(cutoff), ptr(recip),
         ptr(thr), mul3(reps), ptr(deg), stream())
    _exclusive_scan(deg, N, rowptr)
    # every kernel downstream walks by this row pointer: a list that outgrew the capacity must not lead them past the buffers.  The
    # true count is kept aside for the caller's check, the row pointer is cut at the capacity (rows behind it become empty; the fill
    # writes nothing at or past `cap`): the step's results are then wrong but every access stays in bounds
    count = rowptr[N:].clone()
    rowptr.clamp_(max=cap)
    call("xeq_radius_graph_pbc_fill_pruned", dt, ptr(pos_wrap), ptr(ptr_), G, N, ptr(img), ptr(cells), ptr(shift), n_cells,
         float(cutoff), ptr(recip), ptr(thr), mul3(reps), ptr(rowptr), cap, ptr(edge_index), ptr(cell_offsets), stream())
    return rowptr, count


# ------------------------------------------------------------------- edge geometry
class EdgeVectors(Function):
    """vec = pos[c] - pos[n] - cell_offsets @ cell[batch[n]], dist = |vec| (nn/basic.py:110-131).

    ``strain`` [G,3,3] is the zero tensor of the virial branch (nn/basic.py:93-107): positions and cell are scaled
    by (1 + sym(strain)), hence vec -> vec (1 + sym(strain)).  At strain = 0 the forward values do not change;
    the backward pass returns dE/dstrain = sym(sum_{e in graph} vec_e (x) dE/dvec_e), a deterministic segmented
    sum over the center-sorted edges (``edge_graph_ptr``: first edge of every graph)."""

    @staticmethod
    def forward(ctx, pos, graph: EdgeGraph, cell, cell_offsets, batch, strain=None, graph_ptr=None):
        require_hip(pos)
        pos_c = pos.contiguous()
        E = graph.n_edges
        vec = torch.empty((E, 3), dtype=pos.dtype, device=pos.device)
        dist = torch.empty((E,), dtype=pos.dtype, device=pos.device)
        cell, cell_offsets, batch = _c(cell), _c(cell_offsets), _c(batch)
        call("xeq_edge_vectors_fwd", dtype_code(pos), ptr(pos_c), ptr(graph.edge_index), E, ptr(cell), ptr(cell_offsets),
             ptr(batch), ptr(vec), ptr(dist), stream())
        graph._basis = None   # records of an earlier geometry on this graph
        for plan in (graph._wq or {}).values():
            plan["records"] = None
        ctx.graph = graph
        ctx.graph_ptr = graph_ptr
        ctx.n_graphs = None if strain is None else strain.shape[0]
        ctx.save_for_backward(vec, dist)
        # an output nobody differentiated through arrives as None in backward, not as a zero tensor: the fused blocks read `vec`
        # only, and a materialised zero dL/ddist cost four elementwise launches per evaluation to add nothing
        ctx.set_materialize_grads(False)
        return vec, dist

    @staticmethod
    @once_differentiable
    def backward(ctx, g_vec, g_dist):
        vec, dist = ctx.saved_tensors
        graph = ctx.graph
        g = g_vec if g_vec is not None else torch.zeros_like(vec)
        if g_dist is not None:
            g = g + (g_dist / dist.clamp_min(torch.finfo(dist.dtype).tiny)).unsqueeze(-1) * vec
        g = g.contiguous()
        grad_pos = None
        if ctx.needs_input_grad[0]:
            grad_pos = torch.empty((graph.n_nodes, 3), dtype=vec.dtype, device=vec.device)
            n_rowptr, n_perm = graph.reverse_view()
            call("xeq_edge_vectors_bwd", dtype_code(vec), ptr(g), graph.n_nodes, ptr(graph.c_rowptr), ptr(graph.c_perm),
                 ptr(n_rowptr), ptr(n_perm), ptr(grad_pos), stream())
        grad_strain = None
        if ctx.n_graphs is not None and ctx.needs_input_grad[5]:
            outer = (vec.unsqueeze(2) * g.unsqueeze(1)).reshape(-1, 9)            # vec_e (x) dE/dvec_e
            if graph.c_perm is not None:
                outer = outer.index_select(0, graph.c_perm.long())                 # walk the edges center-sorted
            eptr = graph.c_rowptr.long().index_select(0, ctx.graph_ptr.long())     # first edge of every graph
            m = SegmentSum.apply(outer.contiguous(), eptr).view(ctx.n_graphs, 3, 3)
            grad_strain = 0.5 * (m + m.transpose(1, 2))
        return grad_pos, None, None, None, None, grad_strain, None


# ------------------------------------------------------------- e3nn-style operators
class SphHarm(Function):
    @staticmethod
    def forward(ctx, vec, mul, normalize):
        require_hip(vec)
        vec = vec.contiguous()
        n = vec.shape[0]
        D = mul[0] + 3 * mul[1] + 5 * mul[2]
        out = torch.empty((n, D), dtype=vec.dtype, device=vec.device)
        call("xeq_sph_harm_fwd", dtype_code(vec), ptr(vec), n, mul3(mul), int(normalize), ptr(out), stream())
        ctx.save_for_backward(vec)
        ctx.mul, ctx.normalize = mul, normalize
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (vec,) = ctx.saved_tensors
        g = g.contiguous()
        gv = torch.empty_like(vec)
        call("xeq_sph_harm_bwd", dtype_code(vec), ptr(vec), ptr(g), vec.shape[0], mul3(ctx.mul), int(ctx.normalize), ptr(gv), stream())
        return gv, None, None


def _etp(x, g, mul):
    n = x.shape[0]
    out = torch.empty_like(x)
    call("xeq_elementwise_tp_fwd", dtype_code(x), ptr(x), ptr(g), n, g.shape[0], mul3(mul), ptr(out), stream())
    return out


def _cdot(a, b, mul):
    n = a.shape[0]
    out = torch.empty((n, sum(mul)), dtype=a.dtype, device=a.device)
    call("xeq_channel_dot_fwd", dtype_code(a), ptr(a), ptr(b), n, mul3(mul), ptr(out), stream())
    return out


class ElementwiseTP(Function):
    """out[n,u,m] = x[n,u,m] * g[n,u]; g may be a single broadcast row."""

    @staticmethod
    def forward(ctx, x, g, mul):
        require_hip(x, g)
        x, g = x.contiguous(), g.contiguous()
        if g.dim() == 1:
            g = g.unsqueeze(0)
        ctx.save_for_backward(x, g)
        ctx.mul = mul
        return _etp(x, g, mul)

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        x, g = ctx.saved_tensors
        go = go.contiguous()
        gx = _etp(go, g, ctx.mul) if ctx.needs_input_grad[0] else None
        gg = None
        if ctx.needs_input_grad[1]:
            gg = _cdot(go, x, ctx.mul)
            if g.shape[0] == 1:
                gg = gg.sum(0, keepdim=True)
        return gx, gg, None


class ChannelDot(Function):
    """out[n,u] = sum_m a[n,u,m] b[n,u,m]."""

    @staticmethod
    def forward(ctx, a, b, mul):
        require_hip(a, b)
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        ctx.mul = mul
        return _cdot(a, b, mul)

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        a, b = ctx.saved_tensors
        go = go.contiguous()
        ga = _etp(b, go, ctx.mul) if ctx.needs_input_grad[0] else None
        gb = _etp(a, go, ctx.mul) if ctx.needs_input_grad[1] else None
        return ga, gb, None


class EqLayerNorm(Function):
    """EquivariantLayerNorm.forward (nn/o3layer.py:145-171); gradient w.r.t. x only."""

    @staticmethod
    def forward(ctx, x, weight, bias, mul, eps):
        require_hip(x, weight, bias)
        x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
        out = torch.empty_like(x)
        call("xeq_eqln_fwd", dtype_code(x), ptr(x), ptr(weight), ptr(bias), x.shape[0], mul3(mul), float(eps), ptr(out), stream())
        ctx.save_for_backward(x, weight)
        ctx.mul, ctx.eps = mul, eps
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise NotImplementedError("xequinet_amd: this operator has no parameter gradients (the training pass is nn/training.py); "
                                      "call model.requires_grad_(False) / model.eval()")
        x, weight = ctx.saved_tensors
        go = go.contiguous()
        gx = torch.empty_like(x)
        call("xeq_eqln_bwd", dtype_code(x), ptr(x), ptr(weight), ptr(go), x.shape[0], mul3(ctx.mul), float(ctx.eps), ptr(gx), stream())
        return gx, None, None, None, None


class SegmentSum(Function):
    """out[g] = sum_{i in [ptr[g], ptr[g+1])} src[i]  (scatter_sum over a sorted batch index)."""

    @staticmethod
    def forward(ctx, src, ptr_, index=None):
        """``index`` (optional): the sorted segment index per row (keys.BATCH) -- the reverse pass is then one gather."""
        require_hip(src, ptr_)
        src = src.contiguous()
        ptr_ = ptr_.to(torch.int64).contiguous()
        ctx.index = index
        G = ptr_.numel() - 1
        width = 1
        for d in src.shape[1:]:
            width *= int(d)
        out = torch.empty((G,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        call("xeq_segment_sum", dtype_code(src), ptr(src), ptr(ptr_), G, width, ptr(out), stream())
        ctx.save_for_backward(ptr_)
        ctx.n = src.shape[0]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        (ptr_,) = ctx.saved_tensors
        if ctx.index is not None and ctx.index.shape[0] == ctx.n:
            return go.index_select(0, ctx.index.long()), None, None   # (repeat_interleave: subtract, scan, search, gather)
        counts = ptr_[1:] - ptr_[:-1]
        return torch.repeat_interleave(go, counts, dim=0, output_size=ctx.n), None, None


def scatter_add(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """Unsorted scatter-sum (float atomics; forward only)."""
    require_hip(src, index)
    src = src.contiguous()
    width = 1
    for d in src.shape[1:]:
        width *= int(d)
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    call("xeq_scatter_add", dtype_code(src), ptr(src), ptr(index.to(torch.int64).contiguous()), src.shape[0], width,
         ptr(out), dim_size, stream())
    return out


def radial_basis(dist, rbf_kind: str, cutoff_kind: str, num_basis: int, cutoff: float, p0, p1=None,
                 want_rbf=True, want_fcut=True):
    """rbf[E,B] and fcut[E] from dist[E] (forward only; the fused message op owns the gradient)."""
    require_hip(dist)
    d = dist.detach().reshape(-1).contiguous()
    n = d.numel()
    rbf = torch.empty((n, num_basis), dtype=d.dtype, device=d.device) if want_rbf else None
    fcut = torch.empty((n,), dtype=d.dtype, device=d.device) if want_fcut else None
    p0 = None if p0 is None else p0.detach().reshape(-1).to(d.dtype).contiguous()
    p1 = None if p1 is None else p1.detach().reshape(-1).to(d.dtype).contiguous()
    call("xeq_radial_fwd", dtype_code(d), ptr(d), n, lib.RBF_KINDS[rbf_kind], lib.CUTOFF_KINDS[cutoff_kind], num_basis,
         float(cutoff), ptr(p0), ptr(p1), ptr(rbf), ptr(fcut), stream())
    return rbf, fcut


# ------------------------------------------------------------------- fused message
_MESSAGE_IMPLS = ("auto", "wq", "sb", "generic")


def _message_impl() -> str:
    import os

    impl = os.environ.get("XEQ_MESSAGE_IMPL", "auto")
    if impl == "valu":   # earlier name of the generic form
        impl = "generic"
    if impl not in _MESSAGE_IMPLS:
        raise ValueError(f"XEQ_MESSAGE_IMPL={impl!r}: expected " + " | ".join(_MESSAGE_IMPLS))
    return impl


def dtype_code_of(dtype) -> int:
    if dtype == torch.float32:
        return lib.XEQ_F32
    if dtype == torch.float64:
        return lib.XEQ_F64
    raise TypeError(f"xequinet_amd supports float32/float64, got {dtype}")


def prefers_sb(n_nodes: int, n_edges: int) -> bool:
    """Where the scalar-broadcast kernels beat the matrix-core ones (measured, MI355X): graphs of a few thousand edges at most (one
    small molecule: the step is launch-bound and sb needs no walk plan; aspirin replay 0.80 against 0.86 ms).  Round 2 also sent
    dense neighbourhoods (>= 40 edges per atom) here; with the split-bf16 filter of round 3 the wq kernels win there too (water-512:
    message kernels 0.49 against 0.74 ms per evaluation, step 1.42 against 1.60 ms).  The rule is stated once, in the C ABI
    (``xeq_message_auto_family``: every front calls it); this is its first clause for callers that only want to know."""
    return n_edges < 4096


def select_message_impl(dtype, n_nodes: int, n_edges: int, num_basis: int, node_dim: int, mul) -> str:
    """Kernel family of the fused message for this configuration AND these sizes.  ``auto``: the scalar-broadcast form where it is
    the faster one (``prefers_sb``), else the wave / quad matrix-core
    form (f32, multiplicities in multiples of 32, num_basis <= 31, 32-bit byte offsets: ~1.8 M atoms / ~28 M padded edge
    slots with the default model), else the scalar-broadcast form (f32 / f64, at most 256 channels, 32-bit element offsets), else the generic
    form (64-bit offsets).  An explicit XEQ_MESSAGE_IMPL is taken as is: its kernels raise when they do not fit."""
    impl = _message_impl()
    if impl != "auto":
        if impl == "wq" and not (dtype == torch.float32 and lib.load().xeq_message_wq_supported(int(num_basis), int(node_dim), mul3(mul))):
            raise RuntimeError("XEQ_MESSAGE_IMPL=wq: this configuration does not fit the matrix-core kernels "
                               "(needs f32, node_dim == mul[0], multiplicities in multiples of 32, num_basis <= 31)")
        return impl
    # the rule itself lives in the C ABI (xeq_message_auto_family), shared with the registered operator
    code = lib.load().xeq_message_auto_family(dtype_code_of(dtype), int(n_nodes), int(n_edges), int(num_basis), int(node_dim), mul3(mul))
    return {0: "wq", 1: "sb", 3: "generic"}[int(code)]


def _wq_edges_per_stream(n_edges: int, n_nodes: int) -> int:
    """Edges per half-wave stream of the wq kernels.  A step (what the four waves of a workgroup walk together: eight
    streams) should gather from few enough nodes for its window to fit LDS: 64 edges per stream is ~30 owner nodes and
    a window of two to four QM9-size molecules.  Small systems get shorter streams so that the launch still spreads over the chip.  XEQ_WQ_EDGES_PER_STREAM
    fixes it."""
    import os

    env = os.environ.get("XEQ_WQ_EDGES_PER_STREAM")
    if env:
        return max(16, int(env))
    return int(lib.load().xeq_message_wq_edges_per_stream(int(n_nodes), int(n_edges)))   # (the C ABI states the rule)


def _basis_cache_hit(cached, vec, key):
    """A cached record set is valid only for the very tensor it was computed from.  ``vec`` is written by a raw
    HIP kernel into a fresh allocation, so its ``_version`` never moves and the caching allocator can hand the next
    evaluation's vec the same address: identity of the tensor object is the test (the cache holds a reference, which
    also keeps that address from being reused while the entry lives)."""
    return cached is not None and cached[0] is vec and cached[1] == key


def edge_basis_wq(vec, plan, n_nodes, rbf_kind, cutoff_kind, num_basis, cutoff, p0, p1, deriv: bool):
    """Per-edge records of the wave / quad kernels in the PADDED WALK ORDER of `plan` (xeq_edge_basis_wq), once per
    evaluation and direction, cached on the plan: value records for the forward walk, value + d/dd records for the
    reverse walk."""
    key = (rbf_kind, cutoff_kind, num_basis, float(cutoff), p0.data_ptr(), p0._version)
    cached = plan["records"]
    if _basis_cache_hit(cached, vec, key) and (cached[3] is not None or not deriv):   # records with derivatives serve both requests
        return cached[2], cached[3]
    E = vec.shape[0]
    width = int(lib.load().xeq_message_wq_record_floats_for(int(num_basis)))
    basis = torch.empty((plan["pcap"], width), dtype=vec.dtype, device=vec.device)
    dbasis = torch.empty((plan["pcap"], width), dtype=vec.dtype, device=vec.device) if deriv else None
    call("xeq_edge_basis_wq", ptr(vec), n_nodes, E, ptr(plan["qptr"]), ptr(plan["peid"]), lib.RBF_KINDS[rbf_kind],
         lib.CUTOFF_KINDS[cutoff_kind], num_basis, float(cutoff), ptr(p0), ptr(p1), ptr(basis), ptr(dbasis), stream())
    plan["records"] = (vec, key, basis, dbasis)
    return basis, dbasis


def edge_basis(vec, graph: EdgeGraph, rbf_kind, cutoff_kind, num_basis, cutoff, p0, p1):
    """Per-edge radial / angular records (xeq_edge_basis), computed once per evaluation and cached
    on the graph: the three message blocks and both directions share them."""
    key = (vec.dtype, rbf_kind, cutoff_kind, num_basis, float(cutoff), p0.data_ptr(), p0._version)
    cached = getattr(graph, "_basis", None)
    if _basis_cache_hit(cached, vec, key):
        return cached[2], cached[3]
    width = lib.load().xeq_edge_basis_width(num_basis)
    E = vec.shape[0]
    basis = torch.empty((E, width), dtype=vec.dtype, device=vec.device)
    dbasis = torch.empty((E, width), dtype=vec.dtype, device=vec.device)
    call("xeq_edge_basis", dtype_code(vec), ptr(vec), E, lib.RBF_KINDS[rbf_kind], lib.CUTOFF_KINDS[cutoff_kind], num_basis,
         float(cutoff), ptr(p0), ptr(p1), ptr(basis), ptr(dbasis), stream())
    graph._basis = (vec, key, basis, dbasis)
    return basis, dbasis


def copy_many(pairs) -> None:
    """dst.copy_(src) for a list of (dst, src) device tensors in ONE launch (xeq_copy_many) -- same dtype, same number of
    elements, both contiguous; anything else takes Tensor.copy_."""
    import ctypes

    fast = []
    for dst, src in pairs:
        if (dst.is_cuda and src.is_cuda and dst.device == src.device and dst.dtype == src.dtype and dst.numel() == src.numel()
                and dst.is_contiguous() and src.is_contiguous() and (dst.numel() * dst.element_size()) % 4 == 0):
            if dst.numel() > 0 and dst.data_ptr() != src.data_ptr():
                fast.append((dst, src))
        else:
            dst.copy_(src, non_blocking=True)
    cap = lib.COPY_MANY_MAX
    for i in range(0, len(fast), cap):
        grp = fast[i : i + cap]
        n = len(grp)
        srcs = (ctypes.c_void_p * n)(*[t.data_ptr() for _, t in grp])
        dsts = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in grp])
        sizes = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t, _ in grp])
        call("xeq_copy_many", n, srcs, dsts, sizes, stream())


_DIFFER_FLAGS = {}


def any_differs(pairs) -> bool:
    """Whether any (a, b) of a few pairs of device tensors differs, with ONE launch and ONE read-back (xeq_compare_many) -- same dtype,
    shape and device, contiguous, whole 4-byte words; anything else takes torch.equal."""
    import ctypes

    fast = []
    for a, b in pairs:
        if a.shape != b.shape or a.dtype != b.dtype:
            return True
        if (a.is_cuda and b.is_cuda and a.device == b.device and a.is_contiguous() and b.is_contiguous()
                and (a.numel() * a.element_size()) % 4 == 0 and len(fast) < lib.COPY_MANY_MAX):
            if a.numel() > 0 and a.data_ptr() != b.data_ptr():
                fast.append((a, b))
        elif not torch.equal(a, b):
            return True
    if not fast:
        return False
    dev = fast[0][0].device
    st = _DIFFER_FLAGS.get(dev)
    if st is None:
        st = _DIFFER_FLAGS[dev] = [torch.zeros(1, dtype=torch.int32, device=dev), 0]
    st[1] = st[1] % 2_000_000_000 + 1          # never the value the flag holds (it holds an older one, or 0)
    n = len(fast)
    pa = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in fast])
    pb = (ctypes.c_void_p * n)(*[t.data_ptr() for _, t in fast])
    sizes = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t, _ in fast])
    call("xeq_compare_many", n, pa, pb, sizes, st[1], ptr(st[0]), stream())
    return int(st[0].item()) == st[1]


_WQ_WEIGHT_PACKS = {}


def wq_packed_weights(w_rbf: torch.Tensor, b_rbf: torch.Tensor, num_basis: int, node_dim: int, mul) -> torch.Tensor:
    """rbf_lin's rows in the wq kernels' LDS layout, per unit (xeq_message_wq_pack_weights), cached per weight version and pack epoch.
    The kernels' own staging reads W with one row per lane -- 64 cache lines per wave instruction -- between every two workgroups of
    a CU slot; from this copy it is a coalesced 16-byte copy (~11 us per launch).  The cache holds the weight tensors themselves, so an
    entry's address cannot be reused by another tensor while the entry lives (at most 32 entries)."""
    key = (w_rbf.data_ptr(), w_rbf._version, b_rbf.data_ptr(), b_rbf._version, int(num_basis), int(node_dim), tuple(int(m) for m in mul),
           lib.pack_epoch())
    hit = _WQ_WEIGHT_PACKS.get(key)
    if hit is not None:
        return hit[0]
    n = int(lib.load().xeq_message_wq_packed_weight_floats(int(num_basis), int(node_dim), mul3(mul)))
    packed = torch.empty(n, dtype=torch.float32, device=w_rbf.device)
    call("xeq_message_wq_pack_weights", ptr(w_rbf), ptr(b_rbf), int(num_basis), int(node_dim), mul3(mul), ptr(packed), stream())
    if len(_WQ_WEIGHT_PACKS) >= 32:
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()   # (rare: another stream -- a step in flight -- may still read the evicted pack; round-5 advisor)
        _WQ_WEIGHT_PACKS.pop(next(iter(_WQ_WEIGHT_PACKS)))
    _WQ_WEIGHT_PACKS[key] = (packed, w_rbf, b_rbf)
    return packed


def message_forward(h, xhat, vec, s, x, w_rbf, b_rbf, p0, p1, graph: EdgeGraph, cfg, want_backward: bool = False):
    """Launch the fused message kernel.  cfg = (rbf_kind, cutoff_kind, num_basis, cutoff, node_dim, mul[, xhat_layout]).
    Returns (s_out, x_out, saved, impl): `saved` is what message_backward needs.  ``want_backward``: a reverse pass will follow
    (wq on a symmetric list: the derivative records are then written by the same launch as the value records, the reverse kernel
    walks this plan)."""
    rbf_kind, cutoff_kind, num_basis, cutoff, node_dim, mul = cfg[:6]
    xl = int(cfg[6]) if len(cfg) > 6 else 0  # layout of xhat / grad_xhat: 0 e3nn, 1 BT
    require_hip(h, xhat, vec, s, x, w_rbf, b_rbf, p0)
    h, xhat, vec, s, x = (t.contiguous() for t in (h, xhat, vec, s, x))
    w_rbf, b_rbf, p0 = w_rbf.contiguous(), b_rbf.contiguous(), p0.reshape(-1).contiguous()
    p1 = None if p1 is None else p1.reshape(-1).contiguous()
    N, E = graph.n_nodes, graph.n_edges
    C, D = sum(mul), mul[0] + 3 * mul[1] + 5 * mul[2]
    assert h.shape == (N, node_dim + 2 * C) and xhat.numel() == N * D and vec.shape == (E, 3)
    assert s.shape == (N, node_dim) and x.shape == (N, D) and w_rbf.shape == (node_dim + 2 * C, num_basis)
    s_out, x_out = torch.empty_like(s), torch.empty_like(x)
    impl = select_message_impl(h.dtype, N, E, num_basis, node_dim, mul)
    if impl == "wq":
        plan = graph.wq_plan(False, _wq_edges_per_stream(E, N))
        basis, _ = edge_basis_wq(vec, plan, N, rbf_kind, cutoff_kind, num_basis, cutoff, p0, p1,
                                 deriv=bool(want_backward and getattr(graph, "mirror_walk", False)))
        wp = wq_packed_weights(w_rbf, b_rbf, num_basis, node_dim, mul)
        KERNEL_TIMER.launch("xeq_message_fwd_wq", N, E, plan["n_ranges"], ptr(plan["sq"]), ptr(plan["sn"]), ptr(plan["win"]), ptr(plan["rowptr"]),
                            ptr(plan["pgath"]), ptr(plan["qinfo"]), ptr(basis), ptr(h), ptr(xhat), ptr(s), ptr(x), ptr(wp),
                            None, num_basis, node_dim, mul3(mul), ptr(s_out), ptr(x_out), xl | lib.WQ_PACKED_WEIGHTS, stream(),
                            label="xeq_message_fwd_wq_first" if xl & lib.XHAT_HIGHER_L_ZERO else None)   # the first-block form moves fewer bytes
        return s_out, x_out, (h, xhat, vec, w_rbf, b_rbf, p0, p1, None, None), impl
    if impl == "sb":
        basis, dbasis = edge_basis(vec, graph, rbf_kind, cutoff_kind, num_basis, cutoff, p0, p1)
        KERNEL_TIMER.launch("xeq_message_fwd_sb", dtype_code(h), N, E, ptr(graph.c_rowptr), ptr(graph.c_perm),
                            ptr(graph.edge_index[1]), ptr(basis), ptr(h), ptr(xhat), ptr(s), ptr(x), ptr(w_rbf), ptr(b_rbf),
                            num_basis, node_dim, mul3(mul), ptr(s_out), ptr(x_out), xl, stream())
        return s_out, x_out, (h, xhat, vec, w_rbf, b_rbf, p0, p1, basis, dbasis), impl
    # generic form: 64-bit offsets, recomputes the per-edge quantities from vec
    KERNEL_TIMER.launch("xeq_message_fwd", dtype_code(h), N, E, ptr(graph.c_rowptr), ptr(graph.c_perm),
                        ptr(graph.edge_index[1]), ptr(vec), ptr(h), ptr(xhat), ptr(s), ptr(x), ptr(w_rbf), ptr(b_rbf),
                        ptr(p0), ptr(p1), lib.RBF_KINDS[rbf_kind], lib.CUTOFF_KINDS[cutoff_kind], num_basis, float(cutoff),
                        node_dim, mul3(mul), ptr(s_out), ptr(x_out), xl, stream())
    return s_out, x_out, (h, xhat, vec, w_rbf, b_rbf, p0, p1, None, None), impl


class EdgeGradDeferral:
    """dL/dvec of ALL message blocks of one evaluation in one launch (round 5).  Every wq block of a force evaluation leaves per-unit
    partials of dL/dd and dL/dY_lm by padded slot; the chain rule to dL/dvec is linear in them, so the blocks' partials are added first
    and the chain rule runs once (xeq_message_wq_edge_grad_sum) when the LAST block of the reverse pass -- the model's first -- has
    run: that block returns the total as its dL/dvec, the others return None (autograd's sum then has one term).  Before: three
    edge-gradient launches and two elementwise adds per evaluation.

    The model creates one per forward call and hangs it on the evaluation's EdgeGraph (nn/model.py); blocks register in their forward
    when their edge vectors ask for a gradient.  Operator-level callers (no model) have none and keep one launch per block."""

    def __init__(self) -> None:
        self.registered = 0
        self.sets = []
        self.sb_vec = None      # scalar-broadcast blocks: the one dL/dvec buffer they add into (XEQ_SB_ACCUM_VEC)
        self.sb_seen = 0

    def register(self) -> bool:
        if self.registered >= lib.WQ_MAX_PART_SETS:
            return False
        self.registered += 1
        return True

    def add(self, parts, vec, graph, plan, mirror: bool, mul):
        """One block's partials; -> dL/dvec [E, 3] from the last registered block, None from the others."""
        self.sets.append(parts)
        if len(self.sets) < self.registered:
            return None
        sets, self.sets = self.sets, []          # (a second reverse pass over a retained graph collects afresh)
        g_vec = torch.empty_like(vec)
        arr = (ctypes.c_void_p * len(sets))(*[t.data_ptr() for t in sets])
        call("xeq_message_wq_edge_grad_sum", ptr(vec), graph.n_nodes, graph.n_edges, ptr(plan["qptr"]), ptr(plan["peid"]),
             ptr(graph.mirror_map if mirror else None), mul3(mul), len(sets), arr, ptr(g_vec), stream())
        return g_vec


    def next_sb(self, vec):
        """A scalar-broadcast block's turn: -> (buffer its kernel writes dL/dvec to, whether it ADDS, whether this block returns it)."""
        first = self.sb_vec is None
        if first:
            self.sb_vec = torch.empty_like(vec)
        self.sb_seen += 1
        buf, last = self.sb_vec, self.sb_seen >= self.registered
        if last:                      # (a second reverse pass over a retained graph starts afresh)
            self.sb_vec, self.sb_seen = None, 0
        return buf, not first, last


def message_backward(saved, graph: EdgeGraph, cfg, impl: str, g_s, g_x, node_grads: bool = True, deferral: Optional[EdgeGradDeferral] = None):
    """Reverse pass of the fused message: (grad_h, grad_xhat, grad_vec, grad_s, grad_x).  node_grads=False: only grad_vec is
    wanted (the first block of a force evaluation); the wq kernel then stores no node gradients and None is returned for them.
    ``deferral`` (wq only): grad_vec is None unless this is the last block the deferral waits for (EdgeGradDeferral)."""
    h, xhat, vec, w_rbf, b_rbf, p0, p1, basis, dbasis = saved
    rbf_kind, cutoff_kind, num_basis, cutoff, node_dim, mul = cfg[:6]
    xl = int(cfg[6]) if len(cfg) > 6 else 0
    D = mul[0] + 3 * mul[1] + 5 * mul[2]
    g_s = torch.zeros((graph.n_nodes, node_dim), dtype=h.dtype, device=h.device) if g_s is None else g_s.contiguous()
    g_x = torch.zeros((graph.n_nodes, D), dtype=h.dtype, device=h.device) if g_x is None else g_x.contiguous()
    skip = impl == "wq" and not node_grads
    g_h, g_xhat = (None, None) if skip else (torch.empty_like(h), torch.empty_like(xhat))
    g_vec = None if (impl in ("wq", "sb") and deferral is not None) else torch.empty_like(vec)  # written at the edge's own position by every kernel family
    if impl == "wq":
        N, E = graph.n_nodes, graph.n_edges
        mirror = getattr(graph, "mirror_walk", False)      # symmetric list: the forward plan and its records serve both directions
        plan = graph.wq_plan(not mirror, _wq_edges_per_stream(E, N))
        basis, dbasis = edge_basis_wq(vec, plan, N, rbf_kind, cutoff_kind, num_basis, cutoff, p0, p1, deriv=True)
        xl_bwd = xl | (lib.WQ_MIRROR_WALK if mirror else 0) | lib.WQ_PACKED_WEIGHTS
        wp = wq_packed_weights(w_rbf, b_rbf, num_basis, node_dim, mul)
        parts = torch.empty(max(1, lib.load().xeq_message_wq_parts_floats(N, E, mul3(mul))), dtype=h.dtype, device=h.device)
        KERNEL_TIMER.launch("xeq_message_bwd_wq", N, E, plan["n_ranges"], ptr(plan["sq"]), ptr(plan["sn"]), ptr(plan["win"]), ptr(plan["rowptr"]),
                            ptr(plan["pgath"]), ptr(plan["qinfo"]), ptr(basis), ptr(dbasis), ptr(h), ptr(xhat), ptr(g_s), ptr(g_x),
                            ptr(wp), None, num_basis, node_dim, mul3(mul), ptr(g_h), ptr(g_xhat), ptr(parts), xl_bwd, stream(),
                            label="xeq_message_bwd_wq_first" if (skip and xl & lib.XHAT_HIGHER_L_ZERO) else None)
        if deferral is not None:
            g_vec = deferral.add(parts, vec, graph, plan, mirror, mul)
        else:
            call("xeq_message_wq_edge_grad", ptr(vec), N, E, ptr(plan["qptr"]), ptr(plan["peid"]), ptr(graph.mirror_map if mirror else None),
                 mul3(mul), ptr(parts), ptr(g_vec), stream())
    elif impl == "sb":
        # with a deferral the blocks of the evaluation share ONE dL/dvec buffer: the first to run (the model's last block) stores, the
        # others add -- the sum autograd would form from their separate results, in its order, without its elementwise launches
        buf, accum, emit = deferral.next_sb(vec) if deferral is not None else (g_vec, False, True)
        KERNEL_TIMER.launch("xeq_message_bwd_sb", dtype_code(h), graph.n_nodes, graph.n_edges, ptr(graph.n_rowptr),
                            ptr(graph.n_perm), ptr(graph.edge_index[0]), ptr(basis), ptr(dbasis), ptr(h), ptr(xhat), ptr(g_s),
                            ptr(g_x), ptr(w_rbf), ptr(b_rbf), num_basis, node_dim, mul3(mul), ptr(g_h), ptr(g_xhat),
                            ptr(buf), xl | (lib.SB_ACCUM_VEC if accum else 0), stream())
        g_vec = buf if emit else None
    else:
        KERNEL_TIMER.launch("xeq_message_bwd", dtype_code(h), graph.n_nodes, graph.n_edges, ptr(graph.n_rowptr),
                            ptr(graph.n_perm), ptr(graph.edge_index[0]), ptr(vec), ptr(h), ptr(xhat), ptr(g_s), ptr(g_x),
                            ptr(w_rbf), ptr(b_rbf), ptr(p0), ptr(p1), lib.RBF_KINDS[rbf_kind], lib.CUTOFF_KINDS[cutoff_kind],
                            num_basis, float(cutoff), node_dim, mul3(mul), ptr(g_h), ptr(g_xhat), ptr(g_vec), xl, stream())
    return g_h, g_xhat, g_vec, g_s, g_x


def message_param_grad(saved, graph: EdgeGraph, cfg, g_s, g_x):
    """Parameter gradients of the radial filter of one message block (training pass): (dL/dW_rbf [H, B], dL/db_rbf [H], dL/dp0 [B],
    dL/dp1 [B] or None) from what message_forward saved and dL/ds_out, dL/dx_out -- ``xeq_message_param_grad`` (one launch, per-workgroup
    partial sums) and the sum over its parts."""
    h, xhat, vec, w_rbf, b_rbf, p0, p1 = saved[:7]
    rbf_kind, cutoff_kind, num_basis, cutoff, node_dim, mul = cfg[:6]
    xl = int(cfg[6]) if len(cfg) > 6 else 0
    N, E, B, H = graph.n_nodes, graph.n_edges, num_basis, h.shape[1]
    L = lib.load()
    if h.is_cuda and L.xeq_message_param_grad_mc_supported(dtype_code(h), lib.RBF_KINDS[rbf_kind], B, node_dim, mul3(mul)):
        # matrix-core form (csrc/xeq_train.hip): per-edge rows once per geometry and parameter state, shared by the blocks
        key = (rbf_kind, cutoff_kind, B, float(cutoff), p0.data_ptr(), p0._version, None if p1 is None else p1._version)
        cached = getattr(graph, "_param_basis", None)
        if _basis_cache_hit(cached, vec, key):
            tab = cached[2]
        else:
            tab = torch.empty((E, int(L.xeq_param_basis_width(lib.RBF_KINDS[rbf_kind], B))), dtype=h.dtype, device=h.device)
            call("xeq_param_basis", ptr(vec), E, lib.RBF_KINDS[rbf_kind], lib.CUTOFF_KINDS[cutoff_kind], B, float(cutoff), ptr(p0), ptr(p1),
                 ptr(tab), stream())
            graph._param_basis = (vec, key, tab)
        n_parts = int(L.xeq_message_param_grad_mc_parts(E, node_dim, mul3(mul)))
        parts = torch.empty((n_parts, H, 64), dtype=h.dtype, device=h.device)
        g_x_bt = torch.empty(g_x.numel(), dtype=h.dtype, device=h.device)     # dL/dx_out rows contiguous over the channels
        call("xeq_to_bt", ptr(g_x.contiguous()), N, mul3(mul), ptr(g_x_bt), stream())
        KERNEL_TIMER.launch("xeq_message_param_grad_mc", N, E, ptr(graph.edge_index[0]), ptr(graph.edge_index[1]), ptr(tab), ptr(h), ptr(xhat),
                            ptr(g_s.contiguous()), ptr(g_x_bt), lib.RBF_KINDS[rbf_kind], B, node_dim, mul3(mul), xl & 1, 1,
                            ptr(graph.c_rowptr[N:]) if getattr(graph, "edge_count_on_device", False) else None, n_parts, ptr(parts), stream())
        total = parts.sum(0)
        d_p1 = None if p1 is None else (w_rbf * total[:, 2 * B + 1 : 3 * B + 1]).sum(0)
        return total[:, :B], total[:, B], (w_rbf * total[:, B + 1 : 2 * B + 1]).sum(0), d_p1
    n_parts = int(L.xeq_message_param_grad_parts(N))
    parts = torch.empty((n_parts, H, 3 * B + 1), dtype=h.dtype, device=h.device)
    KERNEL_TIMER.launch("xeq_message_param_grad", dtype_code(h), N, E, ptr(graph.n_rowptr), ptr(graph.n_perm), ptr(graph.edge_index[0]),
                        ptr(vec), ptr(h), ptr(xhat), ptr(g_s.contiguous()), ptr(g_x.contiguous()), ptr(p0), ptr(p1),
                        lib.RBF_KINDS[rbf_kind], lib.CUTOFF_KINDS[cutoff_kind], B, float(cutoff), node_dim, mul3(mul), xl & 1, n_parts,
                        ptr(parts), stream())
    total = parts.sum(0)
    d_w, d_b = total[:, :B], total[:, B]
    d_p0 = (w_rbf * total[:, B + 1 : 2 * B + 1]).sum(0)
    d_p1 = None if p1 is None else (w_rbf * total[:, 2 * B + 1 :]).sum(0)
    return d_w, d_b, d_p0, d_p1


def register_edge_grad(graph: EdgeGraph, impl: str, vec_needs_grad: bool) -> Optional[EdgeGradDeferral]:
    """The evaluation's EdgeGradDeferral for a block that runs the wq or the scalar-broadcast kernels and owes a dL/dvec, registered;
    else None.  (All blocks of an evaluation run the same family: it follows from the sizes.)"""
    d = getattr(graph, "edge_grad_deferral", None)
    if d is None or impl not in ("wq", "sb") or not vec_needs_grad or not d.register():
        return None
    return d


class FusedMessage(Function):
    """nn/xpainn.py:140-159 in one kernel; see xeq_message_fwd / xeq_message_bwd."""

    @staticmethod
    def forward(ctx, h, xhat, vec, s, x, w_rbf, b_rbf, p0, p1, graph: EdgeGraph, cfg):
        s_out, x_out, saved, impl = message_forward(h, xhat, vec, s, x, w_rbf, b_rbf, p0, p1, graph, cfg,
                                                    want_backward=any(ctx.needs_input_grad))
        ctx.p_shapes = (p0.shape, None if p1 is None else p1.shape)
        ctx.save_for_backward(*saved)
        ctx.graph, ctx.cfg, ctx.impl = graph, cfg, impl
        ctx.deferral = register_edge_grad(graph, impl, ctx.needs_input_grad[2])
        return s_out, x_out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_s, g_x):
        saved = ctx.saved_tensors
        # no node input asks for a gradient: the model's first block in a force evaluation (its node features do not depend on the
        # positions, nn/xpainn.py first-block table) -- only dL/dvec is formed
        node_grads = any(ctx.needs_input_grad[i] for i in (0, 1, 3, 4))
        g_h, g_xhat, g_vec, g_s, g_x = message_backward(saved, ctx.graph, ctx.cfg, ctx.impl, g_s, g_x, node_grads=node_grads,
                                                        deferral=ctx.deferral)
        d_w = d_b = d_p0 = d_p1 = None
        if any(ctx.needs_input_grad[5:9]):   # first order only: a loss on forces differentiates the reverse pass itself (nn/training.py)
            d_w, d_b, d_p0, d_p1 = message_param_grad(saved, ctx.graph, ctx.cfg, g_s, g_x)
            d_p0 = d_p0.view(ctx.p_shapes[0])
            d_p1 = None if d_p1 is None else d_p1.view(ctx.p_shapes[1])
        return g_h, g_xhat, g_vec, g_s, g_x, d_w, d_b, d_p0, d_p1, None, None


# ---- the message block of a TRAINING pass with forces in the loss: differentiable twice ---------------------------------------------
def training_records(rbf_env: torch.Tensor, env: torch.Tensor, y1: torch.Tensor, y2: torch.Tensor) -> torch.Tensor:
    """Per-edge records in xeq_edge_basis's layout -- [f rho_k (B) | 0 to a multiple of 4 | f | Y_1 (3) | Y_2 (5) | 0 0 0] -- formed from
    differentiable tensors (nn/training.py: radial basis x envelope [E, B], envelope [E, 1], harmonics), so autograd carries both orders
    of the geometry and of the trainable basis parameters through them."""
    E, B = rbf_env.shape
    bp = (B + 3) & ~3
    z = rbf_env.new_zeros
    return torch.cat([rbf_env, z((E, bp - B)), env.reshape(E, 1), y1, y2, z((E, 3))], dim=1).contiguous()


def _diff_sizes(cfg):
    num_basis, node_dim, mul = cfg
    C, D = sum(mul), mul[0] + 3 * mul[1] + 5 * mul[2]
    return num_basis, node_dim, mul, C, D, node_dim + 2 * C, (num_basis + 3) & ~3


def _diff_fwd(h, xhat, rec, w, b, graph: "EdgeGraph", cfg, y0_zero: bool = False, s_in=None, x_in=None):
    """(s_in + sum_e msg_s, x_in + sum_e msg_x); s_in / x_in None: the aggregates alone."""
    B, F, mul, C, D, H, bp = _diff_sizes(cfg)
    N, E = graph.n_nodes, graph.n_edges
    ds, dx = torch.empty((N, F), dtype=h.dtype, device=h.device), torch.empty((N, D), dtype=h.dtype, device=h.device)
    KERNEL_TIMER.launch("xeq_message_fwd_sb", dtype_code(h), N, E, ptr(graph.c_rowptr), ptr(graph.c_perm), ptr(graph.edge_index[1]),
                        ptr(rec), ptr(h), ptr(xhat), ptr(s_in), ptr(x_in), ptr(w), ptr(b), B, F, mul3(mul), ptr(ds), ptr(dx),
                        lib.SB_Y0_ZERO if y0_zero else 0, stream())
    return ds, dx


def _diff_bwd(h, xhat, rec, w, b, g_s, g_x, graph: "EdgeGraph", cfg, q=None, y0_zero: bool = False, want_gy: bool = True):
    """(grad_h, grad_xhat, q, gy) of xeq_message_bwd_sbq; ``q`` given: the per-edge products are added to it."""
    B, F, mul, C, D, H, bp = _diff_sizes(cfg)
    N, E = graph.n_nodes, graph.n_edges
    g_h, g_xh = torch.empty_like(h), torch.empty_like(xhat)
    flags = (lib.SB_Y0_ZERO if y0_zero else 0) | (lib.SB_Q_ACCUMULATE if q is not None else 0) | (0 if want_gy else lib.SB_NO_GY)
    if q is None:
        q = torch.empty((E, H), dtype=h.dtype, device=h.device)
    # every edge of an exact list is walked and writes its row; only the slots behind the count of a capacity-sized list need zeros
    gy = None if not want_gy else (torch.zeros if getattr(graph, "edge_count_on_device", False) else torch.empty)((E, 8), dtype=h.dtype, device=h.device)
    KERNEL_TIMER.launch("xeq_message_bwd_sbq", dtype_code(h), N, E, ptr(graph.n_rowptr), ptr(graph.n_perm), ptr(graph.edge_index[0]),
                        ptr(rec), ptr(h), ptr(xhat), ptr(g_s), ptr(g_x), ptr(w), ptr(b), B, F, mul3(mul), ptr(g_h), ptr(g_xh),
                        ptr(q), ptr(gy), flags, stream())
    if getattr(graph, "edge_count_on_device", False) and not flags & lib.SB_Q_ACCUMULATE:
        # capacity-sized list: the slots behind the true count were not walked, and the products over all rows of q must read zeros there
        call("xeq_zero_rows_from", ptr(q), H * (q.element_size() // 4), E, ptr(graph.c_rowptr[N:]), stream())
    return g_h, g_xh, q, gy


def _diff_fwd_pair(h, u_h, xhat, u_xh, rec, rec_u, w, b, graph: "EdgeGraph", cfg, s_in=None, x_in=None):
    """Passes A (h <- u_h) and B ((xhat, Y) <- (u_xh, harmonics of rec_u)) of the second order in one walk (xeq_message_fwd_sb_pair):
    (s_in + A's scalar aggregate, x_in + both equivariant aggregates)."""
    B, F, mul, C, D, H, bp = _diff_sizes(cfg)
    N, E = graph.n_nodes, graph.n_edges
    ds, dx = torch.empty((N, F), dtype=h.dtype, device=h.device), torch.empty((N, D), dtype=h.dtype, device=h.device)
    KERNEL_TIMER.launch("xeq_message_fwd_sb_pair", dtype_code(h), N, E, ptr(graph.c_rowptr), ptr(graph.c_perm), ptr(graph.edge_index[1]),
                        ptr(rec), ptr(rec_u), ptr(h), ptr(u_h), ptr(xhat), ptr(u_xh), ptr(s_in), ptr(x_in), ptr(w), ptr(b), B, F, mul3(mul),
                        ptr(ds), ptr(dx), 0, stream())
    return ds, dx


def _diff_bwd_pair(h, u_h, xhat, u_xh, rec, rec_u, w, b, g_s, g_x, graph: "EdgeGraph", cfg):
    """The same two passes' reverse walk (xeq_message_bwd_sbq_pair): (B's dL/dh, A's dL/dxhat, q_A + q_B, A's dL/dY)."""
    B, F, mul, C, D, H, bp = _diff_sizes(cfg)
    N, E = graph.n_nodes, graph.n_edges
    capacity = getattr(graph, "edge_count_on_device", False)
    g_h, g_xh = torch.empty_like(h), torch.empty_like(xhat)
    q = torch.empty((E, H), dtype=h.dtype, device=h.device)
    gy = (torch.zeros if capacity else torch.empty)((E, 8), dtype=h.dtype, device=h.device)
    KERNEL_TIMER.launch("xeq_message_bwd_sbq_pair", dtype_code(h), N, E, ptr(graph.n_rowptr), ptr(graph.n_perm), ptr(graph.edge_index[0]),
                        ptr(rec), ptr(rec_u), ptr(h), ptr(u_h), ptr(xhat), ptr(u_xh), ptr(g_s), ptr(g_x), ptr(w), ptr(b), B, F, mul3(mul),
                        ptr(g_h), ptr(g_xh), ptr(q), ptr(gy), 0, stream())
    if capacity:
        call("xeq_zero_rows_from", ptr(q), H * (q.element_size() // 4), E, ptr(graph.c_rowptr[N:]), stream())
    return g_h, g_xh, q, gy


def diff_message_supported(h: torch.Tensor, graph: "EdgeGraph", cfg) -> bool:
    B, F, mul, C, D, H, bp = _diff_sizes(cfg)
    return bool(h.is_cuda and h.dtype in (torch.float32, torch.float64)
                and lib.load().xeq_message_sb_fits(graph.n_nodes, graph.n_edges, B, F, mul3(mul)))


_GEOMETRY_ONLY_TASKS = set()     # ids of autograd graph tasks that were started for gradients w.r.t. the geometry only


class geometry_only_backward:
    """``with geometry_only_backward(energy): autograd.grad(energy, [pos, strain], ...)`` -- the force / virial evaluation of nn/basic.py:
    143-199.  That reverse pass wants no parameter gradient, but a Python ``Function`` cannot see it (``needs_input_grad`` says "the parameter
    requires grad", not "this pass wants it"), and ``DiffMessage.backward`` would form dL/d[W | b] three times per evaluation for the engine
    to drop (0.9 ms of a QM9-1024 training step).  The pass is identified by its GRAPH TASK: a pre-hook on the root node notes the task's id
    when the engine starts it, ``DiffMessage.backward`` looks its own task up -- exact under concurrent backward passes of other threads
    (the engine runs all of them on one device thread; a process-wide flag would leak into theirs).  Without the private id call
    (``torch._C._current_graph_task_id``) nothing is skipped."""

    def __init__(self, root: torch.Tensor) -> None:
        self.node = root.grad_fn if torch.is_tensor(root) else None
        self.ids, self.handle = [], None

    def _note(self, _grads):
        tid = torch._C._current_graph_task_id()
        if tid >= 0:
            self.ids.append(tid)
            _GEOMETRY_ONLY_TASKS.add(tid)

    def __enter__(self):
        if self.node is not None and hasattr(torch._C, "_current_graph_task_id"):
            self.handle = self.node.register_prehook(self._note)
        return self

    def __exit__(self, *exc):
        if self.handle is not None:
            self.handle.remove()
        for tid in self.ids:
            _GEOMETRY_ONLY_TASKS.discard(tid)
        return False


def _in_geometry_only_task() -> bool:
    return bool(_GEOMETRY_ONLY_TASKS) and torch._C._current_graph_task_id() in _GEOMETRY_ONLY_TASKS


class DiffMessage(Function):
    """nn/xpainn.py:140-159 without the residual: (sum_e msg_s, sum_e msg_x) from h = scalar_mlp(s), xhat, the per-edge records
    (``training_records``) and rbf_lin's weight [2C+F, B] / bias.  First derivatives w.r.t. all five; the reverse pass is itself a
    differentiable node (``DiffMessageGrad``), so a loss on forces (nn/basic.py:143-159, create_graph=training) stays on the kernels."""

    @staticmethod
    def forward(ctx, h, xhat, rec, w, b, graph, cfg):
        require_hip(h, xhat, rec, w, b)
        h, xhat, rec, w, b = (t.contiguous() for t in (h, xhat, rec, w, b))
        ds, dx = _diff_fwd(h, xhat, rec, w, b, graph, cfg)
        ctx.save_for_backward(h, xhat, rec, w, b)
        ctx.graph, ctx.cfg = graph, cfg
        return ds, dx

    @staticmethod
    def backward(ctx, g_s, g_x):
        h, xhat, rec, w, b = ctx.saved_tensors
        B, F, mul, C, D, H, bp = _diff_sizes(ctx.cfg)
        N = ctx.graph.n_nodes
        g_s = h.new_zeros((N, F)) if g_s is None else g_s
        g_x = h.new_zeros((N, D)) if g_x is None else g_x
        want = (ctx.needs_input_grad[2], (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]) and not _in_geometry_only_task())
        g_h, g_xh, g_rec, g_w, g_b = DiffMessageGrad.apply(h, xhat, rec, w, b, g_s, g_x, ctx.graph, ctx.cfg, want)
        return g_h, g_xh, g_rec, g_w, g_b, None, None


def _rec_from_q(q, gy, w, b, bp):
    """dL/drecord [E, bp + 12] from the per-edge products: head = q [W | b], harmonics from the kernel."""
    B = w.shape[1]
    head = q @ torch.cat([w, b.unsqueeze(1)], dim=1)       # [E, B + 1]
    z = q.new_zeros
    return torch.cat([head[:, :B], z((q.shape[0], bp - B)), head[:, B:], gy, z((q.shape[0], 3))], dim=1)


def _wb_from_q(q, rec, cfg):
    """dL/dW [H, B], dL/db [H] = q^T (record head | envelope column): xeq_message_q_wgrad and the sum over its chunks."""
    B, F, mul, C, D, H, bp = _diff_sizes(cfg)
    E = q.shape[0]
    n_chunks = int(lib.load().xeq_message_q_wgrad_chunks(E))
    parts = torch.empty((max(n_chunks, 1), H, bp + 1), dtype=q.dtype, device=q.device)
    if n_chunks == 0:
        parts.zero_()
    KERNEL_TIMER.launch("xeq_message_q_wgrad", dtype_code(q), ptr(q), ptr(rec), E, B, F, mul3(mul), n_chunks, ptr(parts), stream())
    g = parts.sum(0)
    return g[:, :B].contiguous(), g[:, bp].contiguous()


class DiffMessageGrad(Function):
    """The reverse pass of ``DiffMessage`` as a differentiable node: (h, xhat, rec, w, b, g_s, g_x) -> (dh, dxhat, drec, dw, db).
    The message is multilinear in its operands, so this node's own reverse pass is three pairs of the same two kernels with one operand
    replaced by the incoming cotangent (include/xeq.h, xeq_message_bwd_sbq).  Cotangents of dw / db (a third order) are refused."""

    @staticmethod
    def forward(ctx, h, xhat, rec, w, b, g_s, g_x, graph, cfg, want):
        B, F, mul, C, D, H, bp = _diff_sizes(cfg)
        g_s, g_x = g_s.contiguous(), g_x.contiguous()
        g_h, g_xh, q, gy = _diff_bwd(h, xhat, rec, w, b, g_s, g_x, graph, cfg)
        g_rec = _rec_from_q(q, gy, w, b, bp) if want[0] else None
        g_w, g_b = _wb_from_q(q, rec, cfg) if want[1] else (None, None)
        ctx.save_for_backward(h, xhat, rec, w, b, g_s, g_x)
        ctx.graph, ctx.cfg = graph, cfg
        ctx.set_materialize_grads(False)
        return g_h, g_xh, g_rec, g_w, g_b

    @staticmethod
    @once_differentiable
    def backward(ctx, u_h, u_xh, u_rec, u_w, u_b):
        if u_w is not None or u_b is not None:
            raise NotImplementedError("DiffMessageGrad: derivatives of the filter's parameter gradients are not implemented")
        h, xhat, rec, w, b, g_s, g_x = ctx.saved_tensors
        graph, cfg = ctx.graph, ctx.cfg
        B, F, mul, C, D, H, bp = _diff_sizes(cfg)
        N, E = graph.n_nodes, graph.n_edges
        need = ctx.needs_input_grad
        zeros = lambda *shape: h.new_zeros(shape)
        d_h = d_xh = d_gs = d_gx = None
        gy_sum = None
        q_ab = None        # products of the passes whose record head is the true one: they meet [W | b] and the true head
        d_w = d_b = None

        def add(acc, t):
            return t if acc is None else acc + t

        if u_rec is not None:                    # C: record head <- its cotangent (first: its products are dropped before the others exist)
            rec_c = torch.cat([u_rec[:, : bp + 1], rec[:, bp + 1 :]], dim=1)
            if bp > B:
                rec_c[:, B:bp] = 0
            d_gs, d_gx = _diff_fwd(h, xhat, rec_c, w, b, graph, cfg)
            gh, gxh, q_c, gy = _diff_bwd(h, xhat, rec_c, w, b, g_s, g_x, graph, cfg)
            d_h, d_xh, gy_sum = add(d_h, gh), add(d_xh, gxh), add(gy_sum, gy)
            if need[3] or need[4]:
                d_w, d_b = _wb_from_q(q_c, rec_c, cfg)
            del q_c
        u_y = None if u_rec is None else u_rec[:, bp + 1 : bp + 9]
        has_b = u_xh is not None or u_rec is not None
        if has_b:       # the operands of pass B: (xhat, Y) <- (u_xhat, u_Y)
            ux = zeros(*xhat.shape) if u_xh is None else u_xh.contiguous()
            rec_b = torch.cat([rec[:, : bp + 1], zeros(E, 8) if u_y is None else u_y, zeros(E, 3)], dim=1)
        if u_h is not None and has_b:            # A (h <- u_h) and B share the filter and the gathered rows: one walk each way for both
            u_h = u_h.contiguous()
            d_gs, d_gx = _diff_fwd_pair(h, u_h, xhat, ux, rec, rec_b, w, b, graph, cfg, s_in=d_gs, x_in=d_gx)
            gh, gxh, q_ab, gy = _diff_bwd_pair(h, u_h, xhat, ux, rec, rec_b, w, b, g_s, g_x, graph, cfg)
            d_h, d_xh, gy_sum = add(d_h, gh), add(d_xh, gxh), add(gy_sum, gy)
        elif u_h is not None:                    # A alone
            u_h = u_h.contiguous()
            d_gs, d_gx = _diff_fwd(u_h, xhat, rec, w, b, graph, cfg, s_in=d_gs, x_in=d_gx)     # the kernel adds to the sums so far
            _, gxh, q_ab, gy = _diff_bwd(u_h, xhat, rec, w, b, g_s, g_x, graph, cfg)
            d_xh, gy_sum = add(d_xh, gxh), add(gy_sum, gy)
        elif has_b:                              # B alone (the first block: its h does not depend on the positions); dL/ds_out plays no part
            _, d_gx = _diff_fwd(h, ux, rec_b, w, b, graph, cfg, y0_zero=True, x_in=d_gx)      # (its scalar aggregate is not a term)
            gh, _, q_ab, _ = _diff_bwd(h, ux, rec_b, w, b, zeros(N, F), g_x, graph, cfg, y0_zero=True, want_gy=False)
            d_h = add(d_h, gh)
        d_rec = None
        if q_ab is not None:
            if need[2]:
                d_rec = _rec_from_q(q_ab, gy_sum if gy_sum is not None else zeros(E, 8), w, b, bp)
            if need[3] or need[4]:
                gw, gb = _wb_from_q(q_ab, rec, cfg)
                d_w, d_b = add(d_w, gw), add(d_b, gb)
        elif gy_sum is not None and need[2]:
            d_rec = torch.cat([zeros(E, bp + 1), gy_sum, zeros(E, 3)], dim=1)
        return d_h, d_xh, d_rec, d_w, d_b, d_gs, d_gx, None, None, None

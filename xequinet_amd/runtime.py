"""HIP-graph replay of one energy(+force, +virial) evaluation.

An evaluation is ~170 kernel launches; for small systems (one molecule, an MD frame) the host needs longer to
enqueue them (~3.5 ms) than the MI355X needs to run them (<1 ms).  ``GraphedModel`` captures the model's forward
pass and its force backward once per input *signature* -- (atoms, edges, graphs, dtype, periodic or not) -- into a
HIP graph (``torch.cuda.CUDAGraph``; every kernel of ``libxeq_hip.so`` is launched on the capturing stream) and
afterwards replays it: inputs are copied into the captured buffers, one graph launch runs the whole evaluation.
Results are bitwise those of the eager path (same kernels, same order).

The neighbour list is NOT captured: its edge count has to come back to the host to size the edge arrays (as in the
reference: ``nonzero`` / torch_cluster).  It runs eagerly before the replay; a new edge count is a new signature and
is captured on first sight (a few ms), which suits fixed-topology work (a trajectory of one system mostly keeps E for
many frames only at small cutoffs; batches of recurring shapes always do).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import keys, lib, ops


class _Captured:
    def __init__(self) -> None:
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.inputs: Dict[str, torch.Tensor] = {}
        self.edge_graph: Optional[ops.EdgeGraph] = None
        self.derived_batch = False
        self.outputs: Dict[str, torch.Tensor] = {}
        # the same evaluation WITHOUT the walk-plan rebuild, for replays on an unchanged list (captured on the first such replay, in
        # the first graph's memory pool: the two are never in flight together)
        self.static: Dict[str, object] = {}
        self.graph_same: Optional[torch.cuda.CUDAGraph] = None
        self.outputs_same: Dict[str, torch.Tensor] = {}


class GraphedModel:
    """``GraphedModel(model)(data) -> {energy, atomic_energies, forces[, virial]}`` with HIP-graph replay.

    ``data`` is the dict the model takes (``XequiBatch.to_dict()`` after ``NeighborTransform``); it must carry
    ``edge_index`` and ``ptr``.  The returned tensors are owned by the captured graph and are overwritten by the
    next call with the same signature: clone what must outlive it."""

    _TENSOR_KEYS = (keys.POSITIONS, keys.ATOMIC_NUMBERS, keys.EDGE_INDEX, keys.BATCH, keys.BATCH_PTR, keys.CELL, keys.CELL_OFFSETS)

    def __init__(self, model: torch.nn.Module, compute_forces: bool = True, compute_virial: bool = False,
                 max_graphs: int = 8, warmup: int = 2, tune_gemms: bool = True, reuse_unchanged_topology: bool = False) -> None:
        """``tune_gemms``: time the library GEMM candidates of every new shape once during the warm-up runs
        (``tuning.enable_gemm_autotune``; PyTorch TunableOp, a process-wide switch).  The libraries' default picks
        for few-hundred-row operands are tiles of 128-256 rows on one or two workgroups (35 us per GEMM at 192
        atoms); the timed picks take ~5 us, which halves the replay time of MD-sized systems."""
        self.model = model
        self.tune_gemms = tune_gemms
        # MD engines hand over a neighbour list every step that is, for small systems, the same list for many steps (every atom
        # of a 21-atom molecule sees every other one): comparing it with the list the last replayed graph holds (one kernel and
        # a round trip) is cheaper than rebuilding the sorted views (a dozen launches and their own round trip).  Off by
        # default: a stream of ever-new lists would only pay for the comparison.
        self.reuse_unchanged_topology = reuse_unchanged_topology
        self._last: Optional[_Captured] = None
        self.compute_forces = compute_forces
        self.compute_virial = compute_virial
        self.max_graphs = max_graphs
        self.warmup = warmup
        self._cache: "OrderedDict[tuple, _Captured]" = OrderedDict()
        self.captures = 0
        self._tracked = None
        self._param_state = self._parameter_state()

    # ------------------------------------------------------------------ helpers
    def _parameter_state(self) -> tuple:
        """Version counters of every parameter / buffer.  A captured graph reads the matrix-core kernels' PACKED weight copies
        (made when the graph was captured) next to the live tensors the library GEMMs read: after an in-place update
        (optimizer step, load_state_dict) a replay would mix old and new weights, so the graphs are dropped and re-captured.
        The tensor list is taken once (walking the module tree costs more than a small evaluation); replacing parameter
        OBJECTS of a captured model needs a new GraphedModel."""
        if self._tracked is None:
            m = self.model
            while not isinstance(m, torch.nn.Module) and hasattr(m, "model"):    # a plain callable around a module (md_model._Core)
                m = m.model
            self._tracked = (list(m.parameters()) + list(m.buffers())) if isinstance(m, torch.nn.Module) else []
        return tuple([t._version for t in self._tracked]) + (lib.pack_epoch(),)

    def _signature(self, data, eg: ops.EdgeGraph) -> tuple:
        pos = data[keys.POSITIONS]
        return (tuple(pos.shape), pos.dtype, pos.device.index, int(data[keys.EDGE_INDEX].shape[1]),
                int(data[keys.BATCH_PTR].numel()), keys.CELL in data, eg.c_perm is None, eg.mirror_walk, eg.periodic)

    @staticmethod
    def _edge_graph(data) -> ops.EdgeGraph:
        eg = data.get(keys.EDGE_GRAPH)
        ei = data[keys.EDGE_INDEX]
        if eg is None or eg.n_edges != ei.shape[1] or eg.edge_index.data_ptr() != ei.contiguous().data_ptr():
            eg = ops.EdgeGraph(ei, data[keys.POSITIONS].shape[0], ptr=data.get(keys.BATCH_PTR))
        return eg

    def _run(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        d = dict(data)
        d[keys.POSITIONS] = d[keys.POSITIONS].detach()
        with torch.enable_grad():
            out = self.model(d, compute_forces=self.compute_forces, compute_virial=self.compute_virial)
        return {k: v.detach() for k, v in out.items() if isinstance(v, torch.Tensor)}

    def _capture(self, data, eg: ops.EdgeGraph) -> _Captured:
        c = _Captured()
        c.inputs = {k: data[k].clone() for k in self._TENSOR_KEYS if k in data}
        c.derived_batch = keys.BATCH not in c.inputs
        if c.derived_batch:
            ptr = c.inputs[keys.BATCH_PTR]
            counts = ptr[1:] - ptr[:-1]
            c.inputs[keys.BATCH] = torch.repeat_interleave(torch.arange(counts.numel(), device=ptr.device), counts,
                                                           output_size=c.inputs[keys.POSITIONS].shape[0])
        # a private EdgeGraph over the captured edge_index; its CSR arrays are refreshed in place before every replay
        # The list kind comes from the EdgeGraph's construction (`periodic`), never from what it has built lazily since: a periodic
        # list whose sorted view exists (an fp64 evaluation, a second capture) must NOT be re-created as an open-boundary exact-mirror
        # list -- the offset-blind reverse-edge map pairs every image of (i, j) with the first (j, i) slot (round-5 advisor, high).
        periodic = eg.periodic or (eg.mirror_walk and keys.CELL_OFFSETS in c.inputs and keys.CELL in c.inputs)
        if periodic and eg.mirror_walk and keys.CELL_OFFSETS not in c.inputs:
            raise ValueError("GraphedModel: a periodic mirror-walk list needs data['cell_offsets'] to be captured")
        c.edge_graph = ops.EdgeGraph(c.inputs[keys.EDGE_INDEX], eg.n_nodes, center_sorted=eg.c_perm is None,
                                     ptr=c.inputs[keys.BATCH_PTR], symmetric=eg.mirror_walk,
                                     cell_offsets=c.inputs[keys.CELL_OFFSETS] if (periodic and eg.mirror_walk) else None)
        if periodic and not eg.mirror_walk:
            c.edge_graph.periodic = True
        static = dict(c.inputs)
        static[keys.EDGE_GRAPH] = c.edge_graph
        c.static = static
        # library GEMM selection is timed during the warm-up only: TunableOp is a process-wide switch, so the state the
        # host application had is restored afterwards (selections made here stay cached inside the libraries' wrapper)
        from .tuning import gemm_autotune_scope
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with gemm_autotune_scope(self.tune_gemms), torch.cuda.stream(side):   # warm-up off the capture: GEMM selection, lazy init
            for _ in range(self.warmup):
                self._run(static)
        torch.cuda.current_stream().wait_stream(side)
        c.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: HIP calls of other host threads (RCCL's watchdog in a multi-rank job) do not
        # invalidate the capture
        with torch.cuda.graph(c.graph, capture_error_mode="thread_local"):
            # the walk plans / stream tables of the message kernels depend on the neighbour list only; rebuilding them is part of
            # the replayed graph (fixed sizes, in place over the static CSR arrays), not a dozen host launches in front of it
            c.edge_graph.refresh_plans()
            c.outputs = self._run(static)
        self.captures += 1
        return c

    @staticmethod
    def _refresh(c: _Captured, data, eg: ops.EdgeGraph) -> None:
        """The new inputs and CSR arrays go into the captured buffers in one launch (ops.copy_many)."""
        pairs = [(t, data[k]) for k, t in c.inputs.items() if k in data]
        s = c.edge_graph
        pairs.append((s.c_rowptr, eg.c_rowptr))
        if s.mirror_map is not None:
            pairs.append((s.mirror_map, eg.mirror_map))
        if s._n_view is not None and s._n_view[1] is not s.mirror_map:   # the sorted view proper, where the captured step asked for it
            pairs += [(s.n_rowptr, eg.n_rowptr), (s.n_perm, eg.n_perm)]
        if s.c_perm is not None:
            pairs.append((s.c_perm, eg.c_perm))
        ops.copy_many(pairs)
        if c.derived_batch:   # the caller gave ptr only: the graph index per atom follows the CURRENT ptr
            ptr = c.inputs[keys.BATCH_PTR]
            counts = ptr[1:] - ptr[:-1]
            c.inputs[keys.BATCH].copy_(torch.repeat_interleave(torch.arange(counts.numel(), device=ptr.device), counts,
                                                               output_size=c.inputs[keys.POSITIONS].shape[0]))

    # --------------------------------------------------------------------- call
    def __call__(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        state = self._parameter_state()
        if state != self._param_state:      # weights changed since the graphs were captured
            self._cache.clear()
            self._last = None
            self._param_state = state
        c = self._last
        if (self.reuse_unchanged_topology and c is not None and keys.EDGE_GRAPH not in data
                and self._same_topology(c, data)):
            # same list as the last replay: the captured CSR arrays and plans are current, only the other inputs move -- and the
            # graph replayed is the one WITHOUT the plan rebuild (eight launches at the sizes that take the wq kernels)
            ops.copy_many([(t, data[k]) for k, t in c.inputs.items() if k in data and k != keys.EDGE_INDEX])
            if c.graph_same is None:
                c.graph_same = torch.cuda.CUDAGraph()
                with torch.cuda.graph(c.graph_same, pool=c.graph.pool(), capture_error_mode="thread_local"):
                    c.outputs_same = self._run(c.static)
                self.captures_same_list = getattr(self, "captures_same_list", 0) + 1
            c.graph_same.replay()
            return c.outputs_same
        eg = self._edge_graph(data)
        sig = self._signature(data, eg)
        c = self._cache.get(sig)
        if c is None:
            c = self._capture(data, eg)
            self._cache[sig] = c
            while len(self._cache) > self.max_graphs:
                self._cache.popitem(last=False)
        else:
            self._cache.move_to_end(sig)
        self._refresh(c, data, eg)
        c.graph.replay()
        self._last = c if self._cache.get(sig) is c else None
        return c.outputs

    @staticmethod
    def _same_topology(c: _Captured, data) -> bool:
        ei, ref = data[keys.EDGE_INDEX], c.inputs[keys.EDGE_INDEX]
        if ei.shape != ref.shape or ei.dtype != ref.dtype:
            return False
        for k in (keys.POSITIONS, keys.ATOMIC_NUMBERS, keys.BATCH_PTR, keys.CELL_OFFSETS, keys.CELL) + (() if c.derived_batch else (keys.BATCH,)):
            if (k in data) != (k in c.inputs) or (k in data and (data[k].shape != c.inputs[k].shape or data[k].dtype != c.inputs[k].dtype)):
                return False
        # one launch and one read-back for all of them (ops.any_differs): the list itself; ptr when the graph index was derived from it;
        # a periodic list's image offsets (its mirror map pairs edges by them as well: the same pairs under other offsets are another map)
        pairs = [(ei, ref)]
        if c.derived_batch and keys.BATCH_PTR in data:
            pairs.append((data[keys.BATCH_PTR], c.inputs[keys.BATCH_PTR]))
        if c.edge_graph.mirror_map is not None and keys.CELL_OFFSETS in data:
            pairs.append((data[keys.CELL_OFFSETS], c.inputs[keys.CELL_OFFSETS]))
        return not ops.any_differs(pairs)


# ----------------------------------------------------------------------------------------------- whole step as one graph
def _params_state(step) -> tuple:
    """Version counters of the model's parameters and buffers plus the pack epoch (include/xeq.h): a captured graph holds the packed
    weight copies of the moment it was captured (the pack kernels ran in the warm-up, outside the graph), so a step object
    re-captures when any of them moved (optimizer step, load_state_dict, a replayed captured training step)."""
    tracked = getattr(step, "_tracked_params", None)
    if tracked is None:
        m = step.model
        while not isinstance(m, torch.nn.Module) and hasattr(m, "model"):
            m = m.model
        tracked = step._tracked_params = (list(m.parameters()) + list(m.buffers())) if isinstance(m, torch.nn.Module) else []
    return tuple([t._version for t in tracked]) + (lib.pack_epoch(),)


def pair_capacity(ptr_host) -> int:
    """Upper bound of an open-boundary neighbour list: every ordered pair inside a graph, sum_g n_g (n_g - 1)."""
    import numpy as np

    n = np.diff(np.asarray(ptr_host, dtype=np.int64))
    return int((n * (n - 1)).sum())


def _offer_zero_start(step, data) -> None:
    """A static all-zero start buffer for the equivariant features (nn/xpainn.py, ZERO_EQUIVARIANT): the model's embedding takes it
    instead of filling 36 MB of zeros per evaluation.  Only for models whose embedding says how wide it is."""
    from .nn.xpainn import ZERO_EQUIVARIANT

    net = step.model if isinstance(step.model, torch.nn.Module) else getattr(step.model, "model", None)
    mods = getattr(net, "mods", None)
    emb = mods["embedding"] if (mods is not None and "embedding" in mods) else None
    if emb is None or not hasattr(emb, "node_irreps"):
        return
    zero = getattr(step, "_zero_x", None)
    if zero is None:
        zero = step._zero_x = torch.zeros((step.n_atoms, emb.node_irreps.dim), dtype=step.pos.dtype, device=step.pos.device)
    data[ZERO_EQUIVARIANT] = zero


class GraphedStep:
    """Neighbour list + model as ONE captured HIP graph that does not depend on the edge count (open boundaries).

    ``GraphedModel`` replays the model per (atoms, edges) signature and leaves the neighbour list outside the graph, because the
    list's edge count sizes the edge arrays and has to reach the host (as the reference's ``nonzero`` does,
    data/radius_graph.py:124-125; torch_cluster, data/transform.py:58-64): a stream of batches with ever-new edge counts
    re-captures every time.  Here every array is sized by a CAPACITY -- atoms, graphs, edges -- the count / scan / fill kernels,
    the reverse-edge map, the walk plans, the records and the model all run inside the graph, and the true edge count stays
    on the device (``rowptr[N]`` bounds every walk; the tail of the edge arrays is never read as an edge).  A batch smaller
    than the capacity is padded: trailing atoms with atomic number 0 in one trailing graph, 100 A apart (no edges; their
    energies / forces are cut off the result).  Results are bitwise those of the eager path on the padded batch.

    ``capacity = (n_atoms, n_graphs, n_edges)``; ``n_edges`` must bound the list: ``pair_capacity(ptr)`` always does.
    A batch beyond the capacity raises (make a larger GraphedStep)."""

    PAD_SPACING = 100.0

    def __init__(self, model: torch.nn.Module, capacity, cutoff: Optional[float] = None, compute_forces: bool = True, warmup: int = 2,
                 device=None, dtype=None) -> None:
        self.model = model
        self.n_atoms, self.n_graphs, self.n_edges = (int(c) for c in capacity)
        self.n_graphs += 1                                   # + the padding graph
        self.cutoff = float(model.cutoff_radius if cutoff is None else cutoff)
        self.compute_forces = compute_forces
        p = next(model.parameters())
        dev = p.device if device is None else torch.device(device)
        dt = p.dtype if dtype is None else dtype
        N, G, E = self.n_atoms, self.n_graphs, self.n_edges
        self.pos = torch.zeros((N, 3), dtype=dt, device=dev)
        self.z = torch.zeros(N, dtype=torch.int32, device=dev)
        self.ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev)
        self.batch = torch.zeros(N, dtype=torch.int64, device=dev)
        self.edge_index = torch.zeros((2, max(E, 1)), dtype=torch.int64, device=dev)   # zero = a valid node id in every unused slot
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.outputs: Dict[str, torch.Tensor] = {}
        self.rowptr: Optional[torch.Tensor] = None
        self.warmup = warmup
        self.captures = 0
        self._param_state = None
        # edges of every step run so far (true counts), accumulated on the device inside the step's own launches: what a benchmark
        # reads once behind its timed region (zero it in front)
        self.edge_total = torch.zeros(1, dtype=torch.int64, device=self.pos.device)

    def overflowed(self) -> bool:
        """Whether the last step's neighbour list outgrew the edge capacity (reads the device-side count: a synchronisation).  Such a
        list is replaced by an EMPTY one that reports its true count (xeq_rowptr_guard) -- nothing is written out of bounds -- and the
        step's results are those of the edge-less batch: re-run it on a larger capacity."""
        return bool(self.outputs) and int(self.outputs["n_edges"].item()) > self.n_edges

    # -- the step on the static buffers (what is captured)
    def _step(self) -> Dict[str, torch.Tensor]:
        rowptr, count = ops.radius_graph_capacity(self.pos, self.ptr, self.cutoff, self.edge_index, running_total=self.edge_total)
        eg = ops.EdgeGraph(self.edge_index, self.n_atoms, center_sorted=True, ptr=self.ptr, c_rowptr=rowptr, symmetric=True)
        data = {keys.POSITIONS: self.pos.detach(), keys.ATOMIC_NUMBERS: self.z, keys.EDGE_INDEX: self.edge_index, keys.BATCH: self.batch,
                keys.BATCH_PTR: self.ptr, keys.EDGE_GRAPH: eg}
        _offer_zero_start(self, data)
        with torch.enable_grad():
            out = self.model(data, compute_forces=self.compute_forces, compute_virial=False)
        res = {k: v.detach() for k, v in out.items() if isinstance(v, torch.Tensor)}
        res["n_edges"] = count                              # device-side TRUE edge count (the row pointer itself is cut at the capacity)
        return res

    def _load_shard(self, pos: torch.Tensor, atomic_numbers: torch.Tensor, ptr: torch.Tensor, batch: torch.Tensor, a: int, b: int, g0: int,
                    g1: int) -> None:
        """Molecules g0 .. g1 - 1 (atoms a .. b - 1) of a larger resident batch into the static buffers, renumbered from zero: one launch
        (xeq_load_padded_shard), no slicing or subtraction launches in front."""
        from .lib import call, dtype_code
        from ctypes import c_void_p

        n, g = b - a, g1 - g0
        if n > self.n_atoms or g > self.n_graphs - 1:
            raise ValueError(f"GraphedStep: shard of {n} atoms / {g} graphs exceeds the capacity {self.n_atoms} / {self.n_graphs - 1}")
        assert pos.dtype == self.pos.dtype and pos.is_contiguous() and ptr.dtype == torch.int64 and batch.dtype == torch.int64
        z64 = atomic_numbers.dtype == torch.int64
        assert z64 or atomic_numbers.dtype == torch.int32
        at = lambda t, off: c_void_p(t.data_ptr() + off * t.element_size() * (t.stride(0) if t.dim() > 0 else 1))
        call("xeq_load_padded_shard", dtype_code(pos), at(pos, a), at(atomic_numbers, a), int(z64), at(ptr, g0), at(batch, a), n, g, a, g0,
             self.n_atoms, self.n_graphs, 1.0e4, self.PAD_SPACING, c_void_p(self.pos.data_ptr()), c_void_p(self.z.data_ptr()),
             c_void_p(self.ptr.data_ptr()), c_void_p(self.batch.data_ptr()), c_void_p(torch.cuda.current_stream().cuda_stream))

    def _load(self, pos: torch.Tensor, atomic_numbers: torch.Tensor, ptr: torch.Tensor, batch: Optional[torch.Tensor]) -> None:
        """The batch into the static buffers, padded to the capacity: one launch (xeq_load_padded_batch), no read-back."""
        from .lib import call, dtype_code, ptr as p_, stream

        n, g = int(pos.shape[0]), int(ptr.numel() - 1)
        if n > self.n_atoms or g > self.n_graphs - 1:
            raise ValueError(f"GraphedStep: batch of {n} atoms / {g} graphs exceeds the capacity {self.n_atoms} / {self.n_graphs - 1}")
        pos_c = pos.detach().to(self.pos.dtype).contiguous()   # (batch None: the load kernel finds an atom's graph in ptr itself -- no launches here)
        z64 = atomic_numbers.dtype == torch.int64     # (a torch.long tensor is read as it is: no conversion launch per step)
        z_c = atomic_numbers.contiguous() if z64 else atomic_numbers.to(torch.int32).contiguous()
        ptr_c, batch_c = ptr.to(torch.int64).contiguous(), (None if batch is None else batch.to(torch.int64).contiguous())
        call("xeq_load_padded_batch_z64" if z64 else "xeq_load_padded_batch", dtype_code(pos_c), p_(pos_c), p_(z_c), p_(ptr_c), p_(batch_c), n, g, self.n_atoms, self.n_graphs,
             1.0e4, self.PAD_SPACING, p_(self.pos), p_(self.z), p_(self.ptr), p_(self.batch), stream())

    def _replay(self) -> None:
        """The step on whatever the static buffers hold: captured on first use (and again when the weights moved), replayed after."""
        state = _params_state(self)
        if self.graph is not None and state != self._param_state:   # weights moved since the capture: its packed copies are stale
            self.graph = None
        self._param_state = state
        if self.graph is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                counted = self.edge_total.clone()
                for _ in range(self.warmup):
                    self._step()
                self.edge_total.copy_(counted)                  # the warm-up runs are not steps: they leave the edge counter alone
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.outputs = self._step()
            self.captures += 1
        self.graph.replay()

    def __call__(self, pos: torch.Tensor, atomic_numbers: torch.Tensor, ptr: torch.Tensor, batch: Optional[torch.Tensor] = None,
                 ptr_host=None) -> Dict[str, torch.Tensor]:
        """-> {energy [G], atomic_energies [n], forces [n, 3], n_edges [1] (device)}: views of the graph's output buffers,
        overwritten by the next call.  ``ptr_host`` (optional): checks the edge capacity against the batch on the host."""
        if ptr_host is not None and pair_capacity(ptr_host) > self.n_edges:
            raise ValueError(f"GraphedStep: the batch may hold {pair_capacity(ptr_host)} edges, the capacity is {self.n_edges}")
        self._load(pos, atomic_numbers, ptr, batch)
        self._replay()
        n, g = int(pos.shape[0]), int(ptr.numel() - 1)
        out = {keys.TOTAL_ENERGY: self.outputs[keys.TOTAL_ENERGY][:g], "n_edges": self.outputs["n_edges"]}
        if keys.ATOMIC_ENERGIES in self.outputs:
            out[keys.ATOMIC_ENERGIES] = self.outputs[keys.ATOMIC_ENERGIES][:n]
        if keys.FORCES in self.outputs:
            out[keys.FORCES] = self.outputs[keys.FORCES][:n]
        return out


def auto_lanes(n_atoms: int) -> int:
    """Lanes of a whole-step graph for an open-boundary batch of this many atoms (GraphedLanes).  1: measured on QM9-1024, MD17 x 4096
    and QM9 x 8192 (profiles/r05_lanes.txt), two lanes gained 5 % while a step still carried 34 small launches (round 4: they vanished
    under the other lane's large kernels) and LOSE 3-9 % since those launches are gone -- a half-size node block or message kernel is
    a chain per wave that takes 60-70 % of the full-size launch, and two of them share the CUs' registers.  XEQ_LANES overrides."""
    import os

    env = os.environ.get("XEQ_LANES")
    return max(1, int(env)) if env else 1


class GraphedLanes:
    """One batch of independent molecules as ``lanes`` contiguous molecule ranges whose whole steps (``GraphedStep``: neighbour list +
    model + forces) run CONCURRENTLY as parallel branches of ONE captured HIP graph.

    Why: at 1 024 QM9-size molecules every big kernel of the step is a chain per wave with one or two waves per SIMD -- the node
    block runs 1.14 waves per SIMD, the message kernels wait on their gathers 40 % of the time -- so the chip's matrix pipes are busy
    ~15 % and HBM ~30 % over a step.  Molecules do not interact (data/transform.py:58-64, nn/output.py:124), so two halves of the
    batch are two independent steps; as two branches of one graph their kernels share the CUs (a node-block workgroup leaves half a
    CU's registers and LDS free), and the small launches of one half disappear under the large ones of the other.  Results are those of
    the unsplit step bit for bit (every kernel gives a row the same bits in any batch; both halves stay on one side of the
    node-block threshold or the split is refused).  Measured on QM9-1024: profiles/r05_lanes.txt.

    ``capacity`` as GraphedStep.  ``ptr_host`` is needed per call (the cut is balanced on the host by estimated edges,
    dist.shard_by_edges, cached per ``ptr_host`` object).  Outputs come back in the batch's own order through one copy launch."""

    def __init__(self, model: torch.nn.Module, capacity, lanes: int = 2, cutoff: Optional[float] = None, compute_forces: bool = True,
                 warmup: int = 2, slack: float = 0.15) -> None:
        from . import lib

        assert 2 <= lanes <= lib.COPY_MANY_MAX // 3
        n_atoms, n_graphs, n_edges = (int(c) for c in capacity)
        self.model, self.lanes, self.compute_forces, self.warmup = model, lanes, compute_forces, warmup
        self.n_atoms, self.n_graphs, self.n_edges = n_atoms, n_graphs, n_edges
        grow = lambda v: int(v / lanes * (1.0 + slack)) + 64
        self.steps = [GraphedStep(model, (grow(n_atoms), grow(n_graphs), grow(n_edges)), cutoff=cutoff, compute_forces=compute_forces, warmup=0)
                      for _ in range(lanes)]
        p = self.steps[0].pos
        self.energy = torch.zeros(n_graphs, dtype=p.dtype, device=p.device)
        self.atomic = torch.zeros(n_atoms, dtype=p.dtype, device=p.device)
        self.forces = torch.zeros((n_atoms, 3), dtype=p.dtype, device=p.device) if compute_forces else None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.lane_outputs = None
        self.captures = 0
        self._param_state = None
        self._cuts = (None, None)
        self._streams = [torch.cuda.Stream(device=p.device) for _ in range(lanes - 1)]

    @property
    def edge_total(self) -> torch.Tensor:
        """Edges of every step run so far, all lanes (a device scalar; a synchronisation-free sum of the lanes' counters)."""
        return torch.stack([st.edge_total for st in self.steps]).sum(0)

    def zero_edge_total(self) -> None:
        for st in self.steps:
            st.edge_total.zero_()

    def overflowed(self) -> bool:
        return any(st.outputs and int(st.outputs["n_edges"].item()) > st.n_edges for st in self.steps)

    def _plan(self, ptr_host):
        import numpy as np
        from .dist import shard_by_edges

        if self._cuts[0] is not ptr_host:
            ph = np.asarray(ptr_host, dtype=np.int64)
            cuts = shard_by_edges(ph, self.lanes)
            plan = [(int(ph[g0]), int(ph[g1]), int(g0), int(g1)) for g0, g1 in cuts]
            for (a, b, g0, g1), st in zip(plan, self.steps):
                if b - a > st.n_atoms or g1 - g0 > st.n_graphs - 1 or pair_capacity(ph[g0 : g1 + 1] - ph[g0]) > st.n_edges:
                    raise ValueError(f"GraphedLanes: lane of {b - a} atoms / {g1 - g0} graphs / {pair_capacity(ph[g0:g1 + 1] - ph[g0])} possible edges "
                                     f"exceeds the lane capacity {st.n_atoms} / {st.n_graphs - 1} / {st.n_edges}")
            self._cuts = (ptr_host, plan)
        return self._cuts[1]

    def _capture(self) -> None:
        main = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            counted = [st.edge_total.clone() for st in self.steps]
            for _ in range(self.warmup):
                for st in self.steps:
                    st._step()
            for st, c in zip(self.steps, counted):
                st.edge_total.copy_(c)                          # the warm-up runs are not steps
        main.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        outs = [None] * self.lanes
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            for i in range(1, self.lanes):                      # fork: every lane but the first on its own stream
                self._streams[i - 1].wait_stream(cur)
                with torch.cuda.stream(self._streams[i - 1]):
                    outs[i] = self.steps[i]._step()
            outs[0] = self.steps[0]._step()
            for i in range(1, self.lanes):                      # join
                cur.wait_stream(self._streams[i - 1])
        for st, o in zip(self.steps, outs):
            st.outputs = o
        self.lane_outputs = outs
        self.captures += 1

    def __call__(self, pos: torch.Tensor, atomic_numbers: torch.Tensor, ptr: torch.Tensor, batch: torch.Tensor, ptr_host) -> Dict[str, torch.Tensor]:
        """-> {energy [G], atomic_energies [n], forces [n, 3]}: this object's output buffers, overwritten by the next call."""
        n, g = int(pos.shape[0]), int(ptr.numel() - 1)
        if n > self.n_atoms or g > self.n_graphs:
            raise ValueError(f"GraphedLanes: batch of {n} atoms / {g} graphs exceeds the capacity {self.n_atoms} / {self.n_graphs}")
        plan = self._plan(ptr_host)
        sides = {bool(ops.lib.load().xeq_node_block_auto(b - a)) for a, b, _, _ in plan}
        if len(sides) > 1:
            raise ValueError("GraphedLanes: the lanes fall on both sides of the node-block threshold (their bits would differ from the unsplit step's)")
        pos_c = pos.detach().contiguous()
        ptr_c, batch_c = ptr.to(torch.int64).contiguous(), batch.to(torch.int64).contiguous()
        z_c = atomic_numbers.contiguous() if atomic_numbers.dtype in (torch.int32, torch.int64) else atomic_numbers.to(torch.int32).contiguous()
        for (a, b, g0, g1), st in zip(plan, self.steps):
            st._load_shard(pos_c, z_c, ptr_c, batch_c, a, b, g0, g1)
        state = _params_state(self)
        if self.graph is not None and state != self._param_state:
            self.graph = None
        self._param_state = state
        if self.graph is None:
            self._capture()
        self.graph.replay()
        pairs = []
        for (a, b, g0, g1), o in zip(plan, self.lane_outputs):
            pairs.append((self.energy[g0:g1], o[keys.TOTAL_ENERGY][: g1 - g0]))
            if keys.ATOMIC_ENERGIES in o:
                pairs.append((self.atomic[a:b], o[keys.ATOMIC_ENERGIES][: b - a]))
            if self.forces is not None and keys.FORCES in o:
                pairs.append((self.forces[a:b], o[keys.FORCES][: b - a]))
        ops.copy_many(pairs)                                     # ONE launch: the lanes' results into the batch's own order
        out = {keys.TOTAL_ENERGY: self.energy[:g], keys.ATOMIC_ENERGIES: self.atomic[:n]}
        if self.forces is not None:
            out[keys.FORCES] = self.forces[:n]
        return out


class GraphedStepsInFlight:
    """A STREAM of batches with ``depth`` of them in flight: ``depth`` independent ``GraphedStep`` contexts (their own static buffers
    and captured graphs), each replayed on its own HIP stream, taken in turn.

    Why: one whole step is a chain of 29 launches whose front (load, count / scan / fill of the neighbour list, reverse map, walk plan,
    records: ~110 us of launches that fill a fraction of the chip) and whose tails (the last workgroups of every big launch) leave
    CUs idle; nothing in step k+1 depends on step k (batches are independent: run/inference.py:39-75 walks a DataLoader), so the next
    batch's step is issued on another stream while this one runs and its kernels fill those holes.  Measured on QM9-1024
    (profiles/r05_in_flight.txt): 2.28 -> 2.03 ms per step with two in flight, nothing more from three.  A step's LATENCY does not
    shorten (it grows by what the other step's kernels take from it); the gain is throughput of a stream of batches.

    Every context's results are the bits of a lone ``GraphedStep`` (same kernels on the same static shapes; no kernel's result
    depends on what runs beside it).

        t = steps.submit(pos, z, ptr[, batch])     # enqueue; returns a ticket at once, nothing waits
        out = steps.result(t)                      # the CALLER's stream waits for that step; views valid until ``depth`` submits later
    """

    NODE_BLOCK_WAVES = 4

    def __init__(self, model: torch.nn.Module, capacity, depth: int = 2, **kw) -> None:
        assert depth >= 1
        self.depth = int(depth)
        self.steps = [GraphedStep(model, capacity, **kw) for _ in range(self.depth)]
        dev = self.steps[0].pos.device
        self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.depth)]
        self._done = [torch.cuda.Event() for _ in range(self.depth)]
        self._out = [None] * self.depth
        self._serial = [-1] * self.depth
        self.submitted = 0

    @property
    def edge_total(self) -> torch.Tensor:
        """Edges of every step submitted so far (a device scalar on the caller's stream; waits for the steps in flight)."""
        self.drain()
        return torch.stack([st.edge_total for st in self.steps]).sum(0)

    def zero_edge_total(self) -> None:
        self.drain()
        cur = torch.cuda.current_stream()
        for st, s in zip(self.steps, self._streams):
            st.edge_total.zero_()
            s.wait_stream(cur)

    def submit(self, pos: torch.Tensor, atomic_numbers: torch.Tensor, ptr: torch.Tensor, batch: Optional[torch.Tensor] = None,
               ptr_host=None) -> int:
        """Enqueue one step on the next context's stream (behind whatever the caller's stream has enqueued so far: the inputs may
        have been produced there).  The context's previous step -- ``depth`` submits ago -- is ahead of it on the same stream, so
        its static buffers are free by the time they are overwritten."""
        i = self.submitted % self.depth
        s = self._streams[i]
        s.wait_stream(torch.cuda.current_stream())
        L = ops.lib.load()
        # workgroups of four waves for the node-block launches captured here (the setting is read at launch, i.e. at capture): next to
        # another step's kernels a CU that finishes early is not idle (profiles/r05_in_flight.txt: 2.00 -> 1.89 ms)
        former = L.xeq_node_block_set_waves(self.NODE_BLOCK_WAVES) if self.depth > 1 else None
        try:
            with torch.cuda.stream(s):
                self._out[i] = self.steps[i](pos, atomic_numbers, ptr, batch, ptr_host)
                self._done[i].record(s)
        finally:
            if former is not None:
                L.xeq_node_block_set_waves(former)
        for t in (pos, atomic_numbers, ptr, batch):
            if t is not None and t.is_cuda:
                t.record_stream(s)                           # the caching allocator must not hand the inputs out while the side stream reads them
        self._serial[i] = self.submitted
        self.submitted += 1
        return self._serial[i]

    def result(self, ticket: int) -> Dict[str, torch.Tensor]:
        """The outputs of submit number ``ticket``, ordered into the caller's stream (no host synchronisation)."""
        i = ticket % self.depth
        if self._serial[i] != ticket:
            raise ValueError(f"GraphedStepsInFlight: the buffers of step {ticket} were reused by step {self._serial[i]} (depth {self.depth})")
        torch.cuda.current_stream().wait_event(self._done[i])
        return self._out[i]

    def drain(self) -> None:
        """The caller's stream waits for every step in flight."""
        cur = torch.cuda.current_stream()
        for i in range(self.depth):
            if self._serial[i] >= 0:
                cur.wait_event(self._done[i])

    def __call__(self, pos, atomic_numbers, ptr, batch=None, ptr_host=None) -> Dict[str, torch.Tensor]:
        """One step, waited for (the GraphedStep call form; nothing overlaps this way)."""
        return self.result(self.submit(pos, atomic_numbers, ptr, batch, ptr_host))

    def __del__(self):
        # the contexts' buffers were allocated on the creator's stream and are used on the side streams: nothing of theirs may still be
        # running when the allocator takes the memory back
        try:
            for s in self._streams:
                s.synchronize()
        except Exception:
            pass

    def overflowed(self) -> bool:
        self.drain()
        return any(st.overflowed() for st in self.steps)


class GraphedChunks:
    """ONE batch of independent open-boundary molecules of any size as contiguous molecule ranges ("chunks": at most ``max_edges``
    possible edges each, dist.plan_chunks -- the message kernels address with 32-bit offsets), every chunk a whole captured step
    (``GraphedStep``: neighbour list, walk plan, model, forces over capacity-sized arrays, edge count on the device) and ``depth`` of
    them in flight on their own streams.

    ``evaluate_in_chunks`` builds every chunk's list from the host, reads its edge count back to size the edge arrays and replays a
    model-only graph per (atoms, edges) signature; here nothing reaches the host inside an evaluation: a chunk is one shard-load launch
    (xeq_load_padded_shard), one graph replay and one copy launch that puts its results at the chunk's place in the batch's arrays, and
    the next chunk runs beside it (the same overlap as ``GraphedStepsInFlight``).  Molecules do not interact, so the results are those
    of one evaluation of the whole batch bit for bit (every kernel gives a row the same bits in any batch; chunks on both sides of the
    node-block threshold are refused).

    ``ptr_host``: the batch's graph pointer on the host (sizes the chunks once; a batch with another pointer needs another object)."""

    def __init__(self, model: torch.nn.Module, ptr_host, max_edges: int = None, depth: int = 2, compute_forces: bool = True) -> None:
        import numpy as np

        from .dist import plan_chunks

        ph = np.asarray(ptr_host, dtype=np.int64)
        self.ptr_host = ph
        cuts = plan_chunks(ph, int(WM_MAX_EDGES_PER_CHUNK if max_edges is None else max_edges))
        self.plan = [(int(ph[g0]), int(ph[g1]), int(g0), int(g1)) for g0, g1 in cuts]
        cap = (max(b - a for a, b, _, _ in self.plan) + 64, max(g1 - g0 for _, _, g0, g1 in self.plan),
               max(pair_capacity(ph[g0 : g1 + 1] - ph[g0]) for _, _, g0, g1 in self.plan))
        sides = {bool(ops.lib.load().xeq_node_block_auto(b - a)) for a, b, _, _ in self.plan}
        if len(sides) > 1:
            raise ValueError("GraphedChunks: the chunks fall on both sides of the node-block threshold (their bits would differ from one evaluation's)")
        self.depth = max(1, min(int(depth), len(self.plan)))
        self.steps = [GraphedStep(model, cap, compute_forces=compute_forces) for _ in range(self.depth)]
        p = self.steps[0].pos
        n, g = int(ph[-1]), len(ph) - 1
        self.energy = torch.zeros(g, dtype=p.dtype, device=p.device)
        self.atomic = torch.zeros(n, dtype=p.dtype, device=p.device)
        self.forces = torch.zeros((n, 3), dtype=p.dtype, device=p.device) if compute_forces else None
        self._streams = [torch.cuda.Stream(device=p.device) for _ in range(self.depth)]
        self._done = [torch.cuda.Event() for _ in range(self.depth)]

    @property
    def n_chunks(self) -> int:
        return len(self.plan)

    @property
    def edge_total(self) -> torch.Tensor:
        """Edges of every chunk evaluated so far (a device scalar)."""
        return torch.stack([st.edge_total for st in self.steps]).sum(0)

    def zero_edge_total(self) -> None:
        for st in self.steps:
            st.edge_total.zero_()

    def overflowed(self) -> bool:
        return any(st.overflowed() for st in self.steps)

    def __call__(self, pos: torch.Tensor, atomic_numbers: torch.Tensor, ptr: torch.Tensor, batch: torch.Tensor) -> Dict[str, torch.Tensor]:
        """-> {energy [G], atomic_energies [N], forces [N, 3]}: this object's result arrays (overwritten by the next call), ordered into the
        caller's stream.  ``batch``: the per-atom graph index (int64)."""
        assert int(pos.shape[0]) == int(self.ptr_host[-1]) and int(ptr.numel()) == len(self.ptr_host), "GraphedChunks: another batch than the planned one"
        pos_c = pos.detach().contiguous()
        ptr_c, batch_c = ptr.to(torch.int64).contiguous(), batch.to(torch.int64).contiguous()
        z_c = atomic_numbers.contiguous() if atomic_numbers.dtype in (torch.int32, torch.int64) else atomic_numbers.to(torch.int32).contiguous()
        cur = torch.cuda.current_stream()
        L = ops.lib.load()
        former = L.xeq_node_block_set_waves(GraphedStepsInFlight.NODE_BLOCK_WAVES) if self.depth > 1 else None
        try:
            for k, (a, b, g0, g1) in enumerate(self.plan):
                i = k % self.depth
                st, s = self.steps[i], self._streams[i]
                if k < self.depth:
                    s.wait_stream(cur)                               # the inputs (and the result arrays' last readers) are on the caller's stream
                with torch.cuda.stream(s):
                    st._load_shard(pos_c, z_c, ptr_c, batch_c, a, b, g0, g1)
                    st._replay()
                    o = st.outputs
                    pairs = [(self.energy[g0:g1], o[keys.TOTAL_ENERGY][: g1 - g0])]
                    if keys.ATOMIC_ENERGIES in o:
                        pairs.append((self.atomic[a:b], o[keys.ATOMIC_ENERGIES][: b - a]))
                    if self.forces is not None and keys.FORCES in o:
                        pairs.append((self.forces[a:b], o[keys.FORCES][: b - a]))
                    ops.copy_many(pairs)                             # the chunk's results to their place, behind its replay on its stream
                    self._done[i].record(s)
        finally:
            if former is not None:
                L.xeq_node_block_set_waves(former)
        for t in (pos_c, z_c, ptr_c, batch_c):
            for s in self._streams:
                t.record_stream(s)
        for i in range(self.depth):
            cur.wait_event(self._done[i])
        out = {keys.TOTAL_ENERGY: self.energy, keys.ATOMIC_ENERGIES: self.atomic}
        if self.forces is not None:
            out[keys.FORCES] = self.forces
        return out

    def __del__(self):
        try:    # (as GraphedStepsInFlight: the side streams are idle before the buffers go back to the allocator)
            for s in self._streams:
                s.synchronize()
        except Exception:
            pass


def evaluate_batches(model: torch.nn.Module, batches, capacity, depth: int = 2, compute_forces: bool = True):
    """The loop of run/inference.py:39-75 over an iterable of open-boundary batches ``(pos, atomic_numbers, ptr[, batch])`` already on the
    device, with ``depth`` steps in flight (``GraphedStepsInFlight``): yields one dict per batch, in order -- energy [G], atomic_energies [n],
    forces [n, 3] (own copies), n_edges (device scalar) -- each while the next ``depth - 1`` batches are already running.
    ``capacity = (atoms, graphs, edges)`` must bound every batch (``pair_capacity(ptr_host)`` bounds a batch's edges)."""
    from collections import deque

    fl = GraphedStepsInFlight(model, capacity, depth=depth, compute_forces=compute_forces)
    pending = deque()

    def fetch():
        out = fl.result(pending.popleft())
        return {k: v.clone() for k, v in out.items()}           # (on the caller's stream, behind the step's event)

    for b in batches:
        pending.append(fl.submit(*b))
        if len(pending) == fl.depth:
            yield fetch()
    while pending:
        yield fetch()


class GraphedStepPBC:
    """Neighbour search + model of ONE periodic (or open) system as one captured HIP graph: what an MD engine that hands over
    positions and a box per step runs (the GROMACS-style model, interface/jit_model.py:183-216: ``single_radius_graph`` inside
    ``forward``, then the evaluation).

    ``GraphedModel`` behind an eager search pays, per step, the search's launches from the host, the read-back of the edge count
    that sizes the edge arrays (the reference's ``nonzero``, data/radius_graph.py:124-125) and the sort of the neighbor view.
    Here the edge arrays have a CAPACITY; the search (``ops.radius_graph_pbc_capacity``: image-pruned pair sweep, positions not
    wrapped, as ``single_radius_graph`` does), the neighbor-sorted view (``xeq_csr_by_key_bounded``), walk plans, records and the
    model are ONE graph, the edge count stays on the device.  What does reach the host per step is the box (9 numbers, compared
    with the captured one: the cell-only tables are formed on the host, ``host_cell_tables_np``, and copied into static memory when
    the box moved; another image count per axis re-captures) and -- with ``check=True``, the default -- the edge count next to the
    results the caller reads anyway: a list that outgrew the capacity (``n_edges > capacity``: the kernels cut it, nothing is
    written out of bounds) re-captures with half as much room again and re-runs the step.

    Same kernels in the same order as the eager GROMACS-style model: bit-identical results (tests/test_gpu_interface.py)."""

    GROWTH = 1.5

    def __init__(self, model, n_atoms: int, edge_capacity: int, cutoff: Optional[float] = None,
                 compute_forces: bool = True, warmup: int = 2) -> None:
        """``model``: a BaseModel, or any callable ``(data, compute_forces, compute_virial) -> results`` that wraps one in ``.model``
        (the MD front ends' unit-free core)."""
        self.model = model
        net = model if isinstance(model, torch.nn.Module) else model.model
        self.n_atoms, self.n_edges = int(n_atoms), int(edge_capacity)
        self.cutoff = float(net.cutoff_radius if cutoff is None else cutoff)
        self.compute_forces = compute_forces
        p = next(net.parameters())
        self.device, self.dtype = p.device, p.dtype
        N = self.n_atoms
        self.pos = torch.zeros((N, 3), dtype=self.dtype, device=self.device)
        self.shift = torch.zeros((N, 3), dtype=self.dtype, device=self.device)          # positions are not wrapped (jit_model.py:189-195)
        self.z = torch.zeros(N, dtype=torch.int32, device=self.device)
        self.cell = torch.zeros((1, 3, 3), dtype=self.dtype, device=self.device)
        self.ptr = torch.tensor([0, N], dtype=torch.int64, device=self.device)
        self.batch = torch.zeros(N, dtype=torch.int64, device=self.device)
        self.warmup = warmup
        self.captures = 0
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.outputs: Dict[str, torch.Tensor] = {}
        self._cell_host = None
        self._pbc = None
        self._reps = None
        self._tab_flat: Optional[torch.Tensor] = None
        self._alloc_edges()

    def _alloc_edges(self) -> None:
        E = max(self.n_edges, 1)
        self.edge_index = torch.zeros((2, E), dtype=torch.int64, device=self.device)    # zero = a valid node id in every unused slot
        self.cell_offsets = torch.zeros((E, 3), dtype=self.dtype, device=self.device)

    def _step(self) -> Dict[str, torch.Tensor]:
        from .data.radius_graph import split_cell_tables

        n_cells = (2 * self._reps[0] + 1) * (2 * self._reps[1] + 1) * (2 * self._reps[2] + 1)
        tab = split_cell_tables(self._tab_flat, 1, n_cells, False)
        rowptr, count = ops.radius_graph_pbc_capacity(self.pos, self.ptr, tab["pbc_offsets"], tab["cell_offsets"], self.shift, self.cutoff,
                                                      (tab["recip"], tab["thr"], self._reps), self.edge_index, self.cell_offsets)
        eg = ops.EdgeGraph(self.edge_index, self.n_atoms, center_sorted=True, ptr=self.ptr, c_rowptr=rowptr, capacity_form=True, symmetric=True,
                           cell_offsets=self.cell_offsets)
        data = {keys.POSITIONS: self.pos.detach(), keys.ATOMIC_NUMBERS: self.z, keys.CELL: self.cell, keys.EDGE_INDEX: self.edge_index,
                keys.CELL_OFFSETS: self.cell_offsets, keys.BATCH: self.batch, keys.BATCH_PTR: self.ptr, keys.EDGE_GRAPH: eg}
        _offer_zero_start(self, data)
        with torch.enable_grad():
            out = self.model(data, compute_forces=self.compute_forces, compute_virial=False)
        res = {k: v.detach() for k, v in out.items() if isinstance(v, torch.Tensor)}
        res["n_edges"] = count                              # device-side edge count of the FULL list (above the capacity: see check)
        return res

    def _load_cell(self, cell: torch.Tensor, pbc) -> None:
        """Box and periodicity of this step: the tables follow the box; a new image count or periodicity drops the graph."""
        import numpy as np

        from .data.radius_graph import host_cell_tables_np

        # the usual step: the box the tables were built for (NVT, or NPT between barostat moves) and the very periodicity tensor of the
        # last call -- asked on the device with one launch and one read-back (ops.any_differs) instead of two read-backs of the values
        if (self._cell_host is not None and isinstance(pbc, torch.Tensor) and cell.is_cuda and cell.dtype == self.dtype
                and getattr(self, "_pbc_seen", None) is not None and self._pbc_seen[0] is pbc and self._pbc_seen[1] == pbc._version
                and not ops.any_differs([(cell.detach().reshape(self.cell.shape), self.cell)])):
            return
        c = np.ascontiguousarray(cell.detach().to(self.dtype).reshape(1, 3, 3).cpu().numpy())     # the step's one small round trip
        pbc_ = [bool(v) for v in (pbc.detach().cpu().tolist() if isinstance(pbc, torch.Tensor) else pbc)]
        self._pbc_seen = (pbc, pbc._version) if isinstance(pbc, torch.Tensor) else None
        if self._cell_host is not None and pbc_ == self._pbc and np.array_equal(c, self._cell_host):
            return
        reps, n_cells, flat_h = host_cell_tables_np(c, pbc_, self.cutoff, False, dtype_code_of=self.cell)
        flat = torch.from_numpy(flat_h)
        if self._tab_flat is None or reps != self._reps or flat.numel() != self._tab_flat.numel():
            self._tab_flat = flat.to(self.device)
            self.graph = None                                # another image table: another graph
        else:
            self._tab_flat.copy_(flat, non_blocking=False)
        self._reps, self._pbc, self._cell_host = reps, pbc_, c
        self.cell.copy_(torch.from_numpy(c))

    def _run(self) -> None:
        state = _params_state(self)
        if self.graph is not None and state != getattr(self, "_param_state", None):   # weights moved: the captured packed copies are stale
            self.graph = None
        self._param_state = state
        if self.graph is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self._step()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.outputs = self._step()
            self.captures += 1
        self.graph.replay()

    def __call__(self, pos: torch.Tensor, atomic_numbers: torch.Tensor, cell: torch.Tensor, pbc, check: bool = True) -> Dict[str, torch.Tensor]:
        """-> {energy [1], atomic_energies [n], forces [n, 3], n_edges [1] (device)}: the graph's output buffers, overwritten by the
        next call.  ``check=False`` leaves the capacity check to the caller (``overflowed()``), e.g. behind a timed loop."""
        if pos.shape[0] != self.n_atoms:
            raise ValueError(f"GraphedStepPBC: {pos.shape[0]} atoms, captured for {self.n_atoms}")
        self._load_cell(cell, pbc)
        ops.copy_many([(self.pos, pos.detach().to(self.dtype)), (self.z, atomic_numbers.to(torch.int32))])
        self._run()
        if check:
            while self.overflowed():
                self.n_edges = int(self.GROWTH * max(self.n_edges, int(self.outputs["n_edges"].item()))) + 64
                self._alloc_edges()
                self.graph = None
                self._run()
        return dict(self.outputs)

    def overflowed(self) -> bool:
        """The last step's list did not fit the edge capacity (reads the device-side count: a synchronisation)."""
        return int(self.outputs["n_edges"].item()) > self.n_edges


# ----------------------------------------------------------------------------------------------- chunked evaluation
WM_MAX_EDGES_PER_CHUNK = 8_000_000   # well inside the 14.9 M-edge bound of the matrix-core message kernels


def evaluate_in_chunks(model, pos: torch.Tensor, atomic_numbers: torch.Tensor, ptr: torch.Tensor, ptr_host=None,
                       max_edges: int = WM_MAX_EDGES_PER_CHUNK, compute_forces: bool = True,
                       runner=None) -> Dict[str, torch.Tensor]:
    """Energy (+ forces) of a batch of independent open-boundary molecules of ANY size: molecules are evaluated in
    contiguous ranges of at most ``max_edges`` edges (dist.plan_chunks) and the results concatenated.  Molecules do
    not interact (the neighbour list never crosses graphs, data/transform.py:58-64; the readout is per graph,
    nn/output.py:124), so the result is what one evaluation of the whole batch gives, bit for bit: every kernel's
    per-node / per-graph sums walk the same edges in the same order.  ``runner(data_dict) -> out`` defaults to the
    eager model call; pass a ``GraphedModel`` to replay recurring chunk shapes."""
    from .data import NeighborTransform, XequiBatch
    from .dist import plan_chunks

    if ptr_host is None:
        ptr_host = ptr.cpu().numpy()
    chunks = plan_chunks(ptr_host, max_edges)
    transform = NeighborTransform(model.cutoff_radius)
    energies, atomic, forces, n_edges = [], [], [], 0
    for g0, g1 in chunks:
        a, b = int(ptr_host[g0]), int(ptr_host[g1])
        batch = XequiBatch(pos[a:b].detach(), atomic_numbers[a:b], ptr[g0 : g1 + 1] - ptr[g0])
        batch = transform(batch)
        n_edges += int(batch.edge_index.shape[1])
        if runner is not None:
            out = runner(batch.to_dict())
            out = {k: v.clone() for k, v in out.items()}   # a replayed graph owns its outputs
        else:
            with torch.enable_grad():
                out = model(batch.to_dict(), compute_forces=compute_forces, compute_virial=False)
        energies.append(out[keys.TOTAL_ENERGY].detach())
        if keys.ATOMIC_ENERGIES in out:
            atomic.append(out[keys.ATOMIC_ENERGIES].detach())
        if compute_forces:
            forces.append(out[keys.FORCES].detach())
    res = {keys.TOTAL_ENERGY: torch.cat(energies), "n_edges": n_edges, "n_chunks": len(chunks)}
    if atomic:
        res[keys.ATOMIC_ENERGIES] = torch.cat(atomic)
    if compute_forces:
        res[keys.FORCES] = torch.cat(forces)
    return res

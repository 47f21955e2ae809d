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

from . import keys, ops


class _Captured:
    def __init__(self) -> None:
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.inputs: Dict[str, torch.Tensor] = {}
        self.edge_graph: Optional[ops.EdgeGraph] = None
        self.derived_batch = False
        self.outputs: Dict[str, torch.Tensor] = {}


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
        return tuple([t._version for t in self._tracked])

    def _signature(self, data, eg: ops.EdgeGraph) -> tuple:
        pos = data[keys.POSITIONS]
        return (tuple(pos.shape), pos.dtype, pos.device.index, int(data[keys.EDGE_INDEX].shape[1]),
                int(data[keys.BATCH_PTR].numel()), keys.CELL in data, eg.c_perm is None)

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
        c.edge_graph = ops.EdgeGraph(c.inputs[keys.EDGE_INDEX], eg.n_nodes, center_sorted=eg.c_perm is None,
                                     ptr=c.inputs[keys.BATCH_PTR])
        static = dict(c.inputs)
        static[keys.EDGE_GRAPH] = c.edge_graph
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
        pairs += [(s.c_rowptr, eg.c_rowptr), (s.n_rowptr, eg.n_rowptr), (s.n_perm, eg.n_perm)]
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
            # same list as the last replay: the captured CSR arrays and plans are current, only the other inputs move
            ops.copy_many([(t, data[k]) for k, t in c.inputs.items() if k in data and k != keys.EDGE_INDEX])
            c.graph.replay()
            return c.outputs
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
        if c.derived_batch and keys.BATCH_PTR in data and not torch.equal(data[keys.BATCH_PTR], c.inputs[keys.BATCH_PTR]):
            return False
        return bool(torch.equal(ei, ref))


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

"""One data-parallel optimisation step of the XPaiNN energy model: the reference's inner loop
(utils/trainer.py:286-308: forward in train mode, weighted loss, backward, clip, optimizer step) and its model wrapping
(run/train.py:185-190: ``DistributedDataParallel``, i.e. bucketed gradient all-reduce overlapped with the backward pass --
RCCL over xGMI with backend "nccl" on MI355X nodes, gloo in the CPU tests).  One process per GPU; every rank draws its own
molecules, the only exchange is the gradient all-reduce.

The forward / backward of a training pass is nn/training.py (differentiable to second order, so forces and virial may
enter the loss)."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F_

from . import keys, lib

_LOSSES = {"l1": F_.l1_loss, "mae": F_.l1_loss, "l2": F_.mse_loss, "mse": F_.mse_loss, "smoothl1": F_.smooth_l1_loss}


def weighted_loss(result: Dict[str, torch.Tensor], target: Dict[str, torch.Tensor], weights: Dict[str, float],
                  loss_fn: str = "l2") -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Sum over the named properties of weight x loss(result[prop], target[prop]) (utils/loss.py:47-110).  Properties:
    ``energy``, ``energy_per_atom`` (total energy / atoms of the molecule; the target holds ``ptr``), ``forces``, ``virial``."""
    fn = _LOSSES[loss_fn.lower()]
    if not weights:
        raise ValueError("At least one task should be present")
    terms = {}
    for prop, w in weights.items():
        if prop == keys.ENERGY_PER_ATOM:
            n_atoms = (target[keys.BATCH_PTR][1:] - target[keys.BATCH_PTR][:-1]).to(result[keys.TOTAL_ENERGY].dtype)
            got, want = result[keys.TOTAL_ENERGY] / n_atoms, target[keys.TOTAL_ENERGY] / n_atoms
        else:
            got, want = result[prop], target[prop]
        if got.shape != want.shape:
            raise ValueError(f"{prop}: result {tuple(got.shape)} vs target {tuple(want.shape)}")
        terms[prop] = fn(got, want)
    total = sum(float(weights[p]) * t for p, t in terms.items())
    return total, terms


def wrap_ddp(model: torch.nn.Module, local_rank: Optional[int] = None, find_unused_parameters: bool = False,
             bucket_cap_mb: int = 25) -> torch.nn.Module:
    """run/train.py:185-190.  Without an initialised process group (one GPU) the model is returned as it is.  The whole
    model is 3.5 MB of gradients: one bucket, one all-reduce per step, issued when the last gradient is ready."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP

    device = next(model.parameters()).device
    ids = None if device.type != "cuda" else [device.index if local_rank is None else local_rank]
    return DDP(model, device_ids=ids, find_unused_parameters=find_unused_parameters, bucket_cap_mb=bucket_cap_mb)


def train_step(model: torch.nn.Module, data: Dict[str, torch.Tensor], target: Dict[str, torch.Tensor],
               optimizer: torch.optim.Optimizer, weights: Dict[str, float], loss_fn: str = "l2",
               grad_clip: Optional[float] = None, ema_model=None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """utils/trainer.py:290-311 for one batch.  ``model`` is the (DDP-wrapped) BaseModel; forces / virial are evaluated
    iff they carry a loss weight (trainer.py:173-177).  ``ema_model``: a ``torch.optim.swa_utils.AveragedModel`` as the reference builds it
    (trainer.py:218-227), updated behind the optimizer step (:309-311).  Returns the detached loss and the model's result."""
    model.train()
    compute_forces = keys.FORCES in weights
    compute_virial = keys.VIRIAL in weights
    result = model(dict(data), compute_forces, compute_virial)
    loss, _ = weighted_loss(result, target, weights, loss_fn)
    optimizer.zero_grad()
    loss.backward()
    if grad_clip is not None:
        torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
    optimizer.step()
    if ema_model is not None:
        ema_model.update_parameters(model)
    return loss.detach(), {k: v.detach() for k, v in result.items()}


class GraphedTrainStep:
    """One optimisation step on an ENERGY loss -- neighbour list, forward pass, loss, the native reverse pass with its parameter
    gradients (nn/fused.py) and the optimizer's update -- as ONE captured HIP graph that does not depend on the batch's edge count.

    What ``train_step`` costs on the host for this model is ~400 kernel launches through Python and autograd per step (9.8 ms per
    QM9-1024 step where the GPU needs 8.5); a captured step is one launch.  The arrays have a capacity as in ``runtime.GraphedStep``
    (atoms, graphs, edges <= sum n_g (n_g - 1)); a batch is padded in one launch, the padding atoms sit in one trailing graph whose
    energy is masked out of the loss, so they receive no gradient.  The loss is the reference's weighted l2 / l1 loss on ``energy`` or
    ``energy_per_atom`` (utils/loss.py:47-110, utils/trainer.py:295-302), times ``energy_weight``; with ``forces_weight`` the same loss on
    the forces is added (``target_forces`` per call) and the capture holds the twice-differentiable pass (nn/training.py) instead of the
    native energy-loss pass.

    The optimizer must keep its state on the device (``torch.optim.Adam(..., capturable=True)``).  Python-float hyper-parameters
    (``lr``, betas, ``weight_decay``) are baked into the graph at capture: a scheduler or warm-up that rewrites
    ``param_groups[...]['lr']`` has no effect on replays unless ``lr`` is a device tensor (``Adam(lr=torch.tensor(...),
    capturable=True)``, updated in place) -- ``set_lr`` does that update and re-captures when ``lr`` is a plain float.

    ``grad_clip`` / ``ema_decay`` (round 5) put the rest of the reference's inner loop into the capture: ``clip_grad_norm_`` between the
    reverse pass and the optimizer (utils/trainer.py:303-307; its device-side form: the total norm never reaches the host) and the
    exponential moving average of the parameters behind it (trainer.py:218-227, 309-311: ``AveragedModel`` with avg = decay avg +
    (1 - decay) p, the first update a copy) -- kept in ``ema_parameters`` (one tensor per model parameter, in order; ``copy_ema_to``
    writes them into a model for validation, trainer.py:354-355), as multi-tensor launches, with the "first update copies" rule as a
    device-side weight that is 0 once and ``decay`` from then on.

    The warm-up iterations in front of the capture are real steps on the first batch; model, optimizer and average are put back to
    their state before them, so the first replayed step is the first update."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, capacity, prop: str = keys.TOTAL_ENERGY,
                 loss_fn: str = "l2", cutoff: Optional[float] = None, warmup: int = 3, energy_weight: float = 1.0,
                 forces_weight: Optional[float] = None, grad_clip: Optional[float] = None, ema_decay: Optional[float] = None) -> None:
        from . import runtime

        if prop not in (keys.TOTAL_ENERGY, keys.ENERGY_PER_ATOM):
            raise ValueError("GraphedTrainStep captures an energy loss (energy or energy_per_atom)")
        if loss_fn.lower() not in ("l2", "mse", "l1", "mae"):
            raise ValueError(f"GraphedTrainStep: loss {loss_fn!r}")
        for grp in optimizer.param_groups:
            if not grp.get("capturable", False):
                raise ValueError("GraphedTrainStep needs an optimizer whose state lives on the device (capturable=True)")
        self.model, self.optimizer, self.prop, self.loss_fn = model, optimizer, prop, loss_fn.lower()
        self._gs = runtime.GraphedStep(model, capacity, cutoff=cutoff, compute_forces=False, warmup=0)   # static buffers + padded loads
        g = self._gs
        dev, dt = g.pos.device, g.pos.dtype
        self.target = torch.zeros(g.n_graphs, dtype=dt, device=dev)
        self.mask = torch.zeros(g.n_graphs, dtype=dt, device=dev)         # 1 for the batch's graphs, 0 for the padding graph / unused slots
        # forces in the loss (round 4): the twice-differentiable pass (nn/training.py) inside the same capture.  Its kernels walk the row
        # pointer, so the capacity-sized edge list is safe there (ops._diff_bwd zeroes the per-edge products behind the true count);
        # the padding atoms have no neighbour, feel no force and carry weight 0
        self.energy_weight, self.forces_weight = float(energy_weight), None if forces_weight is None else float(forces_weight)
        self.target_forces = torch.zeros((g.n_atoms, 3), dtype=dt, device=dev) if forces_weight is not None else None
        self.atom_mask = torch.zeros((g.n_atoms, 1), dtype=dt, device=dev) if forces_weight is not None else None
        self.warmup = warmup
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.loss: Optional[torch.Tensor] = None
        self.captures = 0
        self._count = None
        self.grad_clip = None if grad_clip is None else float(grad_clip)
        self.ema_decay = None if ema_decay is None else float(ema_decay)
        self.ema_parameters = None
        if self.ema_decay is not None:
            self._params = [p for p in model.parameters()]
            self.ema_parameters = [p.detach().clone() for p in self._params]
            self._ema_w = torch.zeros((), dtype=dt, device=dev)            # weight of the old average: 0 at the first update (a copy), decay after

    def set_lr(self, lr: float) -> None:
        """A new learning rate for the next steps: written into a tensor-valued ``lr`` in place (the captured graph reads it), a
        float-valued one is replaced and the graph re-captured."""
        for grp in self.optimizer.param_groups:
            if torch.is_tensor(grp["lr"]):
                grp["lr"].fill_(float(lr))
            else:
                grp["lr"] = float(lr)
                self.graph = None

    def _body(self) -> torch.Tensor:
        from . import ops

        g = self._gs
        # a list that outgrows the capacity comes back EMPTY with its true count (xeq_rowptr_guard): no walk leaves a buffer, and the
        # loss of such a replay is multiplied by 0 below, so the wrong forward pass of the edge-less batch feeds no gradient
        rowptr, self._count = ops.radius_graph_capacity(g.pos, g.ptr, g.cutoff, g.edge_index)
        eg = ops.EdgeGraph(g.edge_index, g.n_atoms, center_sorted=True, ptr=g.ptr, c_rowptr=rowptr, symmetric=True)
        eg.edge_count_on_device = True
        data = {keys.POSITIONS: g.pos.detach(), keys.ATOMIC_NUMBERS: g.z, keys.EDGE_INDEX: g.edge_index, keys.BATCH: g.batch,
                keys.BATCH_PTR: g.ptr, keys.EDGE_GRAPH: eg}
        with_forces = self.forces_weight is not None
        result = self.model(data, with_forces, False)
        energy = result[keys.TOTAL_ENERGY]
        diff = energy - self.target
        if self.prop == keys.ENERGY_PER_ATOM:
            diff = diff / (g.ptr[1:] - g.ptr[:-1]).clamp(min=1).to(diff.dtype)
        per_graph = diff * diff if self.loss_fn in ("l2", "mse") else diff.abs()
        loss = (per_graph * self.mask).sum() / self.mask.sum()            # the mean over the batch's own graphs
        if with_forces:                                                   # the mean over the 3 n components of the batch's own atoms
            df = result[keys.FORCES] - self.target_forces
            per_atom = df * df if self.loss_fn in ("l2", "mse") else df.abs()
            loss = self.energy_weight * loss + self.forces_weight * (per_atom * self.atom_mask).sum() / (3.0 * self.atom_mask.sum())
        elif self.energy_weight != 1.0:
            loss = self.energy_weight * loss
        loss = loss * (self._count.reshape(-1)[0] <= g.n_edges).to(loss.dtype)   # device-side: an overflowed replay has zero gradients
        loss.backward()
        if self.grad_clip is not None:      # utils/trainer.py:303-307; no error_if_nonfinite: nothing of the norm is read on the host
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip)
        self.optimizer.step()
        if self.ema_decay is not None:      # trainer.py:309-311: avg <- w avg + (1 - w) p, w = 0 the first time, decay afterwards
            with torch.no_grad():
                torch._foreach_mul_(self.ema_parameters, self._ema_w)
                torch._foreach_add_(self.ema_parameters, torch._foreach_mul([p.detach() for p in self._params], 1.0 - self._ema_w))
                self._ema_w.fill_(self.ema_decay)
        return loss.detach()

    def copy_ema_to(self, model: torch.nn.Module) -> None:
        """The averaged parameters into ``model`` (same architecture): what the reference validates with (utils/trainer.py:354-355)."""
        if self.ema_parameters is None:
            raise ValueError("GraphedTrainStep was built without ema_decay")
        with torch.no_grad():
            for q, a in zip(model.parameters(), self.ema_parameters):
                q.copy_(a)
        lib.bump_pack_epoch()

    def _capture(self) -> None:
        import copy
        import gc

        self.model.train()
        model_state = copy.deepcopy(self.model.state_dict())
        opt_saved = {p: {k: v.clone() for k, v in st.items() if torch.is_tensor(v)} for p, st in self.optimizer.state.items()}
        ema_saved = None if self.ema_parameters is None else ([a.clone() for a in self.ema_parameters], self._ema_w.clone())
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # every iteration in front of the capture on THIS stream (an AccumulateGrad node kept alive
            for _ in range(self.warmup):   # from another stream breaks the capture)
                self.optimizer.zero_grad(set_to_none=True)
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # back to the state in front of the warm-up, IN PLACE: the optimizer's state tensors must exist before the capture (created
        # inside it, their zero-initialisation would be replayed with every step)
        self.model.load_state_dict(model_state)
        with torch.no_grad():
            for p, st in self.optimizer.state.items():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if p in opt_saved and k in opt_saved[p]:
                            v.copy_(opt_saved[p][k])
                        else:
                            v.zero_()
            if ema_saved is not None:
                for a, b in zip(self.ema_parameters, ema_saved[0]):
                    a.copy_(b)
                self._ema_w.copy_(ema_saved[1])
        gc.collect()
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._body()
        self.captures += 1
        # the capture itself ran no kernel; the replay below is the first update

    def __call__(self, pos: torch.Tensor, atomic_numbers: torch.Tensor, ptr: torch.Tensor, target_energy: torch.Tensor,
                 batch: Optional[torch.Tensor] = None, target_forces: Optional[torch.Tensor] = None, ptr_host=None) -> torch.Tensor:
        """One step on this batch; returns the loss (a device scalar owned by the graph, overwritten by the next call).
        ``ptr_host`` (optional, as ``runtime.GraphedStep.__call__``): checks the edge capacity against the batch on the host BEFORE the
        update.  Without it an overflowing batch is caught on the device: its loss and gradients are exactly zero (the optimizer's
        moments still decay), and ``overflowed()`` reports it afterwards."""
        from . import runtime

        g = self._gs
        if ptr_host is not None and runtime.pair_capacity(ptr_host) > g.n_edges:
            raise ValueError(f"GraphedTrainStep: the batch may hold {runtime.pair_capacity(ptr_host)} edges, the capacity is {g.n_edges}")
        n_graphs = int(ptr.numel() - 1)
        g._load(pos, atomic_numbers, ptr, batch)
        self.target.zero_()
        self.target[:n_graphs] = target_energy.to(self.target.dtype)
        self.mask.zero_()
        self.mask[:n_graphs] = 1.0
        if self.forces_weight is not None:
            if target_forces is None:
                raise ValueError("GraphedTrainStep(forces_weight=...) needs target_forces")
            n = int(pos.shape[0])
            self.target_forces.zero_()
            self.target_forces[:n] = target_forces.to(self.target_forces.dtype)
            self.atom_mask.zero_()
            self.atom_mask[:n] = 1.0
        if self.graph is None:
            self._capture()
        self.graph.replay()
        # the replay moved the weights in place without touching their version counters: every cached packed copy (nn/fused.py,
        # nn/nodeblock.py, csrc/xeq_torch.cpp) and every captured inference graph (runtime.Graphed*) is stale from here on
        lib.bump_pack_epoch()
        return self.loss

    def overflowed(self) -> bool:
        """Whether the last step's neighbour list outgrew the edge capacity (a synchronisation).  Such a list is replaced by an EMPTY
        one (xeq_rowptr_guard) and the step's loss multiplied by zero: nothing is overrun and no gradient of the wrong forward pass
        reaches the weights."""
        return self._count is not None and int(self._count.item()) > self._gs.n_edges

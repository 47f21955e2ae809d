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

from . import keys

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
               grad_clip: Optional[float] = None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """utils/trainer.py:290-308 for one batch.  ``model`` is the (DDP-wrapped) BaseModel; forces / virial are evaluated
    iff they carry a loss weight (trainer.py:173-177).  Returns the detached loss and the model's result."""
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
    return loss.detach(), {k: v.detach() for k, v in result.items()}

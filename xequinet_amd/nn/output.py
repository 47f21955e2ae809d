"""Energy head -- mirror of ``EnergyOut`` in ``xequinet/nn/output.py:79-128``
(the only output module on the energy+force path; the other heads are out of
scope)."""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from .. import keys
from ..scatter import scatter_sum
from . import training
from .basic import resolve_activation


class OutputModule(nn.Module):
    extra_properties: List[str]

    def __init__(self) -> None:
        super().__init__()

    def forward(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        raise NotImplementedError


class EnergyOut(OutputModule):
    def __init__(
        self,
        node_dim: int = 128,
        hidden_dim: int = 64,
        activation: str = "silu",
        node_shift: float = 0.0,
        node_scale: float = 1.0,
        **kwargs,
    ) -> None:
        super().__init__()
        self.node_dim = node_dim
        self.hidden_dim = hidden_dim
        final_linear = nn.Linear(self.hidden_dim, 1)
        final_linear.weight.data *= node_scale
        nn.init.constant_(final_linear.bias, node_shift)
        self.out_mlp = nn.Sequential(
            nn.Linear(self.node_dim, self.hidden_dim),
            resolve_activation(activation),
            final_linear,
        )
        self.extra_properties = [keys.TOTAL_ENERGY, keys.ATOMIC_ENERGIES]

    def forward(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        if training.active(self, data):
            return training.energy_out(self, data)
        batch = data[keys.BATCH]
        node_scalar = data[keys.NODE_INVARIANT]
        from .fused import EnergyHead, EnergyReadout

        ptr_ = data.get(keys.BATCH_PTR)
        if (ptr_ is not None and keys.ATOMIC_ENERGIES not in data and not data.get(training.PARAM_GRADS, False)
                and EnergyReadout.supported(self.out_mlp, node_scalar)):
            # the head, the per-graph sum and (saved as one row per node) the head's whole reverse pass: two launches (nn/fused.py)
            atomic_energies, total_energy = EnergyReadout.apply(node_scalar, self.out_mlp, batch.to(torch.int64).contiguous(), ptr_)
            data[keys.ATOMIC_ENERGIES] = atomic_energies
            data[keys.TOTAL_ENERGY] = total_energy
            return data
        if EnergyHead.supported(self.out_mlp, node_scalar):    # matrix-core kernels, explicit reverse pass (nn/fused.py)
            params = EnergyHead.params(self.out_mlp) if data.get(training.PARAM_GRADS, False) else ()
            atom_eng_out = EnergyHead.apply(node_scalar, self.out_mlp, *params)
        else:                                                  # f64, other activations / widths: library GEMMs
            atom_eng_out = self.out_mlp(node_scalar).reshape(-1)
        if keys.ATOMIC_ENERGIES in data:
            atomic_energies = data[keys.ATOMIC_ENERGIES] + atom_eng_out
        else:
            atomic_energies = atom_eng_out
        total_energy = scatter_sum(atomic_energies, batch, dim=0, ptr=data.get(keys.BATCH_PTR))
        data[keys.ATOMIC_ENERGIES] = atomic_energies
        data[keys.TOTAL_ENERGY] = total_energy
        return data


def resolve_output(mode: str, **kwargs) -> OutputModule:
    """nn/output.py output factory, energy mode only."""
    if mode == "energy":
        return EnergyOut(**{k: v for k, v in kwargs.items() if k in ("node_dim", "hidden_dim", "activation", "node_shift", "node_scale")})
    raise NotImplementedError(f"output mode {mode!r} is outside the energy+force hot path")

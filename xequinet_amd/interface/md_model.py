"""MD front ends of the energy+force path: the models LAMMPS and GROMACS drive (single graph, no batch).

Mirror of ``xequinet/interface/jit_model.py``: ``XPaiNNLMP`` (:12-89, ``forward(data, compute_forces, compute_virial)``
in LAMMPS units), ``XPaiNNGMX`` (:148-216, ``forward(positions, atomic_numbers, box, pbc) -> energy`` in GROMACS units
with the neighbour search inside the model) and ``resolve_jit_model`` (:219-236).  Unit factors follow the same rules
(``unit_conversion`` between the model's default units and the engine's).

The reference exports these through ``torch.jit.script`` because its MD plug-ins are libtorch programs; here the
evaluation is the HIP path of this package, called from Python or through ``include/xeq.h``, and MD-sized systems
are latency-bound on kernel launches, so ``replay=True`` runs the evaluation as a captured HIP graph
(``runtime.GraphedModel``): same kernels, same results, one graph launch per step.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import keys, ops
from ..data.radius_graph import single_radius_graph
from ..nn.basic import compute_edge_data, compute_properties
from ..nn.model import BaseModel, XPaiNN
from ..utils import get_default_units, unit_conversion


def _default_unit(prop: str) -> str:
    units = get_default_units()
    if prop not in units:
        raise KeyError(f"default unit of '{prop}' is not set: call utils.set_default_units(ckpt['config']['default_units']) first")
    return units[prop]


class XPaiNNLMP(XPaiNN):
    """XPaiNN for LAMMPS (jit_model.py:12-89).  ``data`` carries ``atomic_numbers``, ``pos`` and the neighbour list
    LAMMPS built (``edge_index`` [+ ``cell_offsets``, ``cell``, ``pbc``]) in LAMMPS units; results come back in
    LAMMPS units.  ``cutoff_radius`` is converted to LAMMPS units for the caller that builds the list."""

    def __init__(self, unit_style: str = "metal", net_charge: Optional[int] = None, replay: bool = False,
                 tune_gemms: bool = True, native: bool = False, **kwargs) -> None:
        """``replay``: HIP-graph replay per (atoms, edges) signature.  ``native``: an evaluation is one registered operator
        (``xeq::xpainn_eval``, every kernel enqueued from C++: the numbers of the Python modules at a third of their host
        time; fp32, a copy of the parameters is taken at first use) -- what ``interface.scripted.XPaiNNLMPScript`` runs."""
        super().__init__(**kwargs)
        lammps_units = keys.LAMMPS_UNIT_STYLE[unit_style]
        self.pos_unit_factor = unit_conversion(lammps_units[keys.POSITIONS], _default_unit(keys.POSITIONS))        # LAMMPS -> model
        self.energy_unit_factor = unit_conversion(_default_unit(keys.TOTAL_ENERGY), lammps_units[keys.TOTAL_ENERGY])  # model -> LAMMPS
        self.forces_unit_factor = unit_conversion(_default_unit(keys.FORCES), lammps_units[keys.FORCES])
        self.net_charge = net_charge
        self.cutoff_radius /= self.pos_unit_factor
        self._replay = None
        self._use_replay = replay
        self._tune_gemms = tune_gemms
        self._use_native, self._native = native, None

    def _evaluate(self, data, compute_forces: bool, compute_virial: bool) -> Dict[str, torch.Tensor]:
        if self._use_native and not self._use_replay and data[keys.POSITIONS].dtype == torch.float32:
            if self._native is None:
                from .scripted import XPaiNNNative
                object.__setattr__(self, "_native", XPaiNNNative(self))     # (not a sub-module: it holds a copy of the parameters)
            pos = data[keys.POSITIONS]
            ptr = data.get(keys.BATCH_PTR)
            if ptr is None:
                ptr = torch.tensor([0, pos.shape[0]], dtype=torch.long, device=pos.device)
            # a list of this package's own search comes with its EdgeGraph and that one's promises (center-sorted; every edge with its
            # mirror); the engine's list (LAMMPS) promises nothing
            graph = data.get(keys.EDGE_GRAPH)
            out = self._native(pos, data[keys.ATOMIC_NUMBERS], data[keys.EDGE_INDEX], ptr, data.get(keys.CELL), data.get(keys.CELL_OFFSETS),
                               graph is not None and graph.c_perm is None, graph is not None and graph.mirror_walk, compute_forces, compute_virial)
            result = {keys.TOTAL_ENERGY: out[0], keys.ATOMIC_ENERGIES: out[1]}
            if compute_forces:
                result[keys.FORCES] = out[2]
            if compute_virial:
                result[keys.VIRIAL] = out[3]
            return result
        if self._use_replay:
            from ..runtime import GraphedModel
            if (self._replay is None or self._replay.compute_forces != compute_forces
                    or self._replay.compute_virial != compute_virial):
                self._replay = GraphedModel(_Core(self), compute_forces=compute_forces, compute_virial=compute_virial,
                                            tune_gemms=self._tune_gemms, reuse_unchanged_topology=True)
            if keys.BATCH_PTR not in data:   # one graph: [0, n], kept per (n, device) -- no host-to-device copy per step
                n, dev = data[keys.POSITIONS].shape[0], data[keys.POSITIONS].device
                if getattr(self, "_one_graph_ptr", None) is None or self._one_graph_ptr[0] != (n, dev):
                    object.__setattr__(self, "_one_graph_ptr", ((n, dev), torch.tensor([0, n], dtype=torch.long, device=dev)))
                data[keys.BATCH_PTR] = self._one_graph_ptr[1]
            # the replayed graph's outputs are its static buffers: whatever forward() does not rescale into a new tensor is copied here
            rescaled = self._rescaled_keys(compute_forces, compute_virial)
            out = self._replay(data)
            fresh = {k: torch.empty_like(v) for k, v in out.items() if k not in rescaled}
            ops.copy_many([(fresh[k], out[k]) for k in fresh])      # one launch for all of them
            return {k: fresh.get(k, v) for k, v in out.items()}
        return _Core(self)(data, compute_forces, compute_virial)

    def _rescaled_keys(self, compute_forces: bool, compute_virial: bool) -> set:
        """Results that forward() multiplies by a unit factor other than 1 (the product is a new tensor)."""
        out = set()
        if self.energy_unit_factor != 1.0:
            out.add(keys.TOTAL_ENERGY)
            if compute_virial:
                out.add(keys.VIRIAL)
        if compute_forces and self.forces_unit_factor != 1.0:
            out.add(keys.FORCES)
        return out

    def forward(self, data: Dict[str, torch.Tensor], compute_forces: bool = True,
                compute_virial: bool = False) -> Dict[str, torch.Tensor]:
        data = dict(data)
        # (a unit factor of exactly 1 -- LAMMPS "metal" against eV / Angstrom -- is not multiplied in: x * 1.0 is x bit for bit, and an
        # MD-sized step pays ~5 us per launch)
        data[keys.POSITIONS] = (data[keys.POSITIONS] * self.pos_unit_factor if self.pos_unit_factor != 1.0
                                else data[keys.POSITIONS].detach())   # (an alias, not the caller's tensor: the evaluation marks it requires_grad)
        # like the reference (jit_model.py:62) only the positions are rescaled: a cell, when present, is taken as given
        if self.net_charge is not None:
            data[keys.TOTAL_CHARGE] = torch.tensor([self.net_charge], device=data[keys.POSITIONS].device)
        result = self._evaluate(data, compute_forces, compute_virial)
        if self.energy_unit_factor != 1.0:
            result[keys.TOTAL_ENERGY] = result[keys.TOTAL_ENERGY] * self.energy_unit_factor
            if compute_virial:
                result[keys.VIRIAL] = result[keys.VIRIAL] * self.energy_unit_factor
        if compute_forces and self.forces_unit_factor != 1.0:
            result[keys.FORCES] = result[keys.FORCES] * self.forces_unit_factor
        return result


class _Core:
    """``BaseModel.forward`` over an MD model's blocks without its unit handling (the callable GraphedModel captures)."""

    def __init__(self, model: BaseModel) -> None:
        self.model = model

    def __call__(self, data, compute_forces: bool = True, compute_virial: bool = False):
        m = self.model
        data = compute_edge_data(data=data, compute_forces=compute_forces, compute_virial=compute_virial)
        # one dL/dvec for all message blocks of this evaluation (ops.EdgeGradDeferral), as BaseModel.forward does (nn/model.py)
        g = data.get(keys.EDGE_GRAPH)
        if g is not None:
            from .. import ops
            g.edge_grad_deferral = ops.EdgeGradDeferral() if (compute_forces or compute_virial) else None
        for mod in m.mods.values():
            data = mod(data)
        return compute_properties(data=data, compute_forces=compute_forces, compute_virial=compute_virial,
                                  training=m.training, extra_properties=m.extra_properties)


class _EnergyOfPositions(torch.autograd.Function):
    """energy(positions) whose reverse pass hands back the forces the replayed evaluation already produced"""

    @staticmethod
    def forward(ctx, positions, energy, forces):
        ctx.save_for_backward(forces)
        return energy.clone()

    @staticmethod
    def backward(ctx, grad_energy):
        (forces,) = ctx.saved_tensors
        return -forces * grad_energy.reshape(-1)[0], None, None


class XPaiNNGMX(XPaiNN):
    """XPaiNN for GROMACS' NNPot interface (jit_model.py:148-216): positions / box in nm, energy in kJ/mol; the
    caller differentiates the returned energy with respect to ``positions`` for the forces.  With ``replay=True`` the
    evaluation (energy and forces) runs as one HIP-graph launch after the neighbour search, and the caller's
    ``backward`` receives those forces."""

    def __init__(self, net_charge: Optional[int] = None, replay: bool = False, tune_gemms: bool = True, whole_step: bool = False,
                 **kwargs) -> None:
        """``replay``: the evaluation behind the search as one HIP-graph launch per (atoms, edges) signature.  ``whole_step`` (with
        ``replay``): search AND evaluation as one graph over capacity-sized edge arrays (runtime.GraphedStepPBC): no edge count
        read back in front of the model, no re-capture when the count moves -- what a trajectory wants."""
        kwargs.pop("unit_style", None)
        super().__init__(**kwargs)
        self._whole_step, self._step_graph = bool(whole_step), None
        self.pos_unit_factor = unit_conversion("nm", _default_unit(keys.POSITIONS))
        self.energy_unit_factor = unit_conversion(_default_unit(keys.TOTAL_ENERGY), "kJ/mol")
        self.forces_unit_factor = unit_conversion(_default_unit(keys.FORCES), "kJ/(mol*nm)")
        self.net_charge = net_charge
        self._use_replay, self._tune_gemms, self._replay = replay, tune_gemms, None

    def forward(self, positions: torch.Tensor, atomic_numbers: torch.Tensor, box: Optional[torch.Tensor] = None,
                pbc: Optional[torch.Tensor] = None) -> torch.Tensor:
        positions = positions * self.pos_unit_factor
        if box is None or pbc is None:   # the stand-ins of an open system, kept per (dtype, device): no fill launches per step
            key = (positions.dtype, positions.device)
            if getattr(self, "_open_system", None) is None or self._open_system[0] != key:
                object.__setattr__(self, "_open_system", (key, torch.eye(3, dtype=positions.dtype, device=positions.device),
                                                          torch.zeros(3, dtype=torch.bool, device=positions.device)))
        cell = self._open_system[1] if box is None else box * self.pos_unit_factor
        if pbc is None:
            pbc = self._open_system[2]
        if self._use_replay and self._whole_step and self.net_charge is None:
            from ..runtime import GraphedStepPBC
            if self._step_graph is None or self._step_graph.n_atoms != positions.shape[0]:
                with torch.no_grad():     # one sized search for the capacity: a quarter more room than the first list needs
                    ei0, _ = single_radius_graph(pos=positions, cell=cell, pbc=pbc, cutoff=self.cutoff_radius)
                self._step_graph = GraphedStepPBC(_Core(self), positions.shape[0], int(1.25 * ei0.shape[1]) + 1024,
                                                  cutoff=self.cutoff_radius, compute_forces=True)
            # the capacity check reads the edge count on the host (a synchronisation): everything that follows the replay is enqueued
            # FIRST, so that the wait covers it instead of standing in front of it; a list that outgrew the capacity (rare: the step
            # graph then grows and re-captures) repeats the step
            sg = self._step_graph
            def result(out):   # the graph's output buffers copied out with ONE launch
                e, f = torch.empty_like(out[keys.TOTAL_ENERGY]), torch.empty_like(out[keys.FORCES])
                ops.copy_many([(e, out[keys.TOTAL_ENERGY]), (f, out[keys.FORCES])])
                return _EnergyOfPositions.apply(positions, e, f) * self.energy_unit_factor

            energy = result(sg(positions.detach(), atomic_numbers, cell, pbc, check=False))
            if sg.overflowed():
                energy = result(sg(positions.detach(), atomic_numbers, cell, pbc))
            return energy
        with torch.no_grad():
            edge_index, cell_offsets, rowptr = single_radius_graph(pos=positions, cell=cell, pbc=pbc, cutoff=self.cutoff_radius,
                                                                   return_rowptr=True)
        data = {
            keys.POSITIONS: positions,
            keys.ATOMIC_NUMBERS: atomic_numbers,
            keys.CELL: cell.unsqueeze(0),
            keys.PBC: pbc.unsqueeze(0),
            keys.EDGE_INDEX: edge_index,
            keys.CELL_OFFSETS: cell_offsets,
            # the search's list is center-sorted and comes with its row pointer: no sortedness check, no second pass over it
            keys.EDGE_GRAPH: ops.EdgeGraph(edge_index, positions.shape[0], center_sorted=True, c_rowptr=rowptr, symmetric=True,
                                           cell_offsets=cell_offsets),
        }
        if self.net_charge is not None:
            data[keys.TOTAL_CHARGE] = torch.tensor([self.net_charge], device=positions.device)
        if self._use_replay:
            from ..runtime import GraphedModel
            if self._replay is None:
                self._replay = GraphedModel(_Core(self), compute_forces=True, compute_virial=False, tune_gemms=self._tune_gemms)
            data[keys.POSITIONS] = positions.detach()
            data[keys.BATCH_PTR] = torch.tensor([0, positions.shape[0]], dtype=torch.long, device=positions.device)
            out = self._replay(data)
            energy = _EnergyOfPositions.apply(positions, out[keys.TOTAL_ENERGY], out[keys.FORCES].clone())
            return energy * self.energy_unit_factor
        data = compute_edge_data(data=data, compute_forces=True, compute_virial=False)
        for mod in self.mods.values():
            data = mod(data)
        return data[keys.TOTAL_ENERGY] * self.energy_unit_factor


def resolve_jit_model(mode: str = "lmp", unit_style: str = "metal", net_charge: Optional[int] = None, **kwargs) -> BaseModel:
    """jit_model.py:219-236; the dipole head is outside the energy+force path."""
    factory = {"lmp": XPaiNNLMP, "gmx": XPaiNNGMX}
    if mode not in factory:
        raise NotImplementedError(f"Unsupported mode {mode}")
    return factory[mode](unit_style=unit_style, net_charge=net_charge, **kwargs)

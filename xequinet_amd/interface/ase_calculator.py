"""ASE calculator over the HIP energy+force(+stress) path -- mirror of ``xequinet/interface/ase_calculator.py:20-118``.

``XequiCalculator(ckpt_file=..., dtype=..., device=...)`` implements ``energy / energies / forces / stress`` in ASE's
units (eV, eV/Angstrom, eV/Angstrom^3 Voigt).  ASE is imported lazily: with ASE installed the class derives from
``ase.calculators.calculator.Calculator`` exactly like the reference's; without it (this build image has none) the
same ``calculate(atoms, properties, system_changes)`` logic runs on a minimal stand-in base, which is what the tests
exercise with a duck-typed ``Atoms``.  A ready model can be passed as ``model=`` instead of a checkpoint file.
``replay=True`` evaluates through a captured HIP graph (``runtime.GraphedModel``), the fast mode for MD.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from .. import keys
from ..data import NeighborTransform, XequiBatch
from ..utils import get_default_units, set_default_units, unit_conversion

try:  # pragma: no cover - ASE is optional
    from ase.calculators.calculator import Calculator as _AseCalculator, all_changes
    _HAVE_ASE = True
except Exception:  # noqa: BLE001
    _HAVE_ASE = False
    all_changes = ["positions", "numbers", "cell", "pbc", "initial_charges", "initial_magmoms"]

    class _AseCalculator:  # the slice of ase's Calculator that calculate() relies on
        default_parameters: Dict = {}

        def __init__(self, **kwargs) -> None:
            self.parameters = dict(self.default_parameters)
            self.results: Dict = {}
            self.atoms = None
            self.set(**kwargs)

        def set(self, **kwargs):
            changed = {k: v for k, v in kwargs.items() if k not in self.parameters or self.parameters[k] != v}
            self.parameters.update(changed)
            return changed

        def reset(self) -> None:
            self.atoms = None
            self.results = {}

        def calculate(self, atoms=None, properties=None, system_changes=None) -> None:
            if atoms is not None:
                self.atoms = atoms.copy()


def full_3x3_to_voigt_6_stress(stress: np.ndarray) -> np.ndarray:
    """xx, yy, zz, yz, xz, xy with the off-diagonals symmetrised (ase.stress convention)."""
    s = np.asarray(stress)
    return np.array([s[0, 0], s[1, 1], s[2, 2], 0.5 * (s[1, 2] + s[2, 1]), 0.5 * (s[0, 2] + s[2, 0]), 0.5 * (s[0, 1] + s[1, 0])])


def datapoint_from_ase(atoms, dtype: Optional[torch.dtype] = None) -> XequiBatch:
    """data/fmt_conversion.py:14-44 for the fields of this path: wrapped positions in the model's length unit,
    atomic numbers, ``pbc`` and ``cell`` only when some direction is periodic."""
    dtype = dtype if dtype is not None else torch.get_default_dtype()
    factor = unit_conversion("Angstrom", get_default_units()[keys.POSITIONS])
    pbc = np.asarray(atoms.get_pbc(), dtype=bool)
    pos = np.asarray(atoms.get_positions(wrap=True), dtype=np.float64) * factor
    cell = np.asarray(getattr(atoms.get_cell(), "array", atoms.get_cell()), dtype=np.float64) * factor if pbc.any() else None
    return XequiBatch(
        pos=torch.from_numpy(pos).to(dtype),
        atomic_numbers=torch.from_numpy(np.asarray(atoms.get_atomic_numbers())).to(torch.int),
        pbc=torch.from_numpy(pbc).view(1, 3) if cell is not None else None,
        cell=torch.from_numpy(cell).view(1, 3, 3).to(dtype) if cell is not None else None,
    )


class XequiCalculator(_AseCalculator):
    implemented_properties = ["energy", "energies", "forces", "stress"]
    default_parameters = {"ckpt_file": "model.pt", "dtype": "float32", "device": None}

    def __init__(self, model: Optional[torch.nn.Module] = None, replay: bool = False, tune_gemms: bool = True, native: bool = False,
                 **kwargs) -> None:
        """``native=True``: an evaluation is ONE registered operator (``xeq::xpainn_eval``: every kernel enqueued from C++; the same
        kernels, the same numbers) instead of ~100 launches through the Python modules -- 0.8 instead of 2 ms for a small
        molecule.  fp32 models with the default blocks; the operator keeps a copy of the parameters taken when it is first used
        (a new checkpoint through ``set(ckpt_file=...)`` takes a new copy).  ``replay=True`` captures HIP graphs per system size
        instead (fastest when atom and edge counts recur)."""
        self.dtype = torch.float32
        self.device = torch.device("cuda")
        self.model = model
        self.transform = None
        self._replay_on = replay
        self._tune_gemms = tune_gemms
        self._replay = {}
        self._native_on = native
        self._native = None
        if model is not None:
            self.device = next(model.parameters()).device
            self.transform = NeighborTransform(model.cutoff_radius)
        _AseCalculator.__init__(self, **kwargs)

    def set(self, **kwargs):
        changed = _AseCalculator.set(self, **kwargs)
        if changed:
            self.reset()
        if "dtype" in changed:
            self.dtype = {"float32": torch.float32, "float64": torch.float64}[self.parameters["dtype"]]
        if "device" in changed and self.parameters["device"] is not None:
            self.device = torch.device(self.parameters["device"])
        if self.model is None or ("ckpt_file" in changed and "ckpt_file" in kwargs):
            from ..nn import resolve_model
            ckpt = torch.load(self.parameters["ckpt_file"], map_location=self.device)
            config = ckpt["config"]
            set_default_units(config["default_units"])
            self.model = resolve_model(config["model_name"], **config["model_kwargs"]).to(self.device).eval()
            self.model.load_reference_state_dict(ckpt["model"])
            self.transform = NeighborTransform(self.model.cutoff_radius)
            self._replay = {}
            self._native = None
        if self.model is not None:
            self.model = self.model.to(self.dtype)
        return changed

    def _evaluate(self, data, compute_forces: bool, compute_virial: bool):
        if self._native_on and not self._replay_on and data[keys.POSITIONS].dtype == torch.float32:
            if self._native is None:
                from .scripted import XPaiNNNative
                self._native = XPaiNNNative(self.model)
            graph = data.get(keys.EDGE_GRAPH)
            symmetric = graph is not None and graph.mirror_walk          # the lists of NeighborTransform (open and periodic): every edge with its mirror
            out = self._native(data[keys.POSITIONS], data[keys.ATOMIC_NUMBERS], data[keys.EDGE_INDEX], data[keys.BATCH_PTR],
                               data.get(keys.CELL), data.get(keys.CELL_OFFSETS), True, symmetric, compute_forces, compute_virial)
            result = {keys.TOTAL_ENERGY: out[0], keys.ATOMIC_ENERGIES: out[1]}
            if compute_forces:
                result[keys.FORCES] = out[2]
            if compute_virial:
                result[keys.VIRIAL] = out[3]
            return result
        if not self._replay_on:
            return self.model(data, compute_forces, compute_virial)
        from ..runtime import GraphedModel
        key = (compute_forces, compute_virial)
        if key not in self._replay:
            self._replay[key] = GraphedModel(self.model, compute_forces=compute_forces, compute_virial=compute_virial,
                                             tune_gemms=self._tune_gemms)
        return self._replay[key](data)

    def calculate(self, atoms=None, properties: Optional[List[str]] = None, system_changes: List[str] = all_changes) -> None:
        if properties is None:
            properties = self.implemented_properties
        _AseCalculator.calculate(self, atoms, properties, system_changes)
        atoms.wrap()
        data = datapoint_from_ase(self.atoms, self.dtype).to(self.device)
        data = self.transform(data).to_dict()
        compute_forces = "forces" in properties
        compute_virial = "stress" in properties
        result = self._evaluate(data, compute_forces, compute_virial)
        units = get_default_units()
        e_fac = unit_conversion(units[keys.TOTAL_ENERGY], "eV")
        self.results["energy"] = result[keys.TOTAL_ENERGY].item() * e_fac
        self.results["energies"] = result[keys.ATOMIC_ENERGIES].detach().cpu().numpy() * e_fac
        if compute_forces:
            self.results["forces"] = result[keys.FORCES].detach().cpu().numpy() * unit_conversion(units[keys.FORCES], "eV/Angstrom")
        if compute_virial:
            cell = np.asarray(getattr(self.atoms.get_cell(), "array", self.atoms.get_cell()))
            assert np.linalg.matrix_rank(cell) == 3
            virial = result[keys.VIRIAL].detach().cpu().numpy().reshape(3, 3) * e_fac
            self.results["stress"] = full_3x3_to_voigt_6_stress(virial) / self.atoms.get_volume()

"""TorchScript front ends: the models a libtorch MD plug-in loads.

Mirror of ``xequinet/run/jit_script.py:28-86`` (``compile_model``: script the MD model, save it with ``_extra_files``) and
of the scripted classes of ``xequinet/interface/jit_model.py`` (``XPaiNNLMP`` :12-89, ``XPaiNNGMX`` :148-216).  The
reference's models script because every op under them is a registered torch operator (ATen, torch_scatter,
torch_cluster); here the operators are ``xeq::xpainn_eval`` / ``xeq::radius_graph`` of ``libxeq_torch.so``
(``csrc/xeq_torch.cpp``), which enqueue the hand-written HIP kernels of ``libxeq_hip.so`` from C++.  A saved file is
loaded by ``torch.jit.load`` after ``load_torch_library()`` (Python) or after ``dlopen("libxeq_torch.so")`` (a libtorch
host program such as the LAMMPS / GROMACS plug-ins).

``XPaiNNNative`` is also the fast path for batches whose topology never repeats: one operator call per evaluation, no
Python between the ~150 kernel launches, no graph capture.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import keys, lib
from ..utils import get_default_units, unit_conversion

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TORCH_LIB_PATH = os.environ.get("XEQ_TORCH_LIB_PATH") or os.path.join(_HERE, "libxeq_torch.so")
_loaded = False

# H .. Rn, as the reference's periodic table (utils/qc.py) spells them in the `_extra_files` of a compiled model
ELEMENTS_LIST = ("X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y "
                 "Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os "
                 "Ir Pt Au Hg Tl Pb Bi Po At Rn").split()


def load_torch_library() -> None:
    """Register the ``xeq::`` operators (fails loudly when the extension has not been built: there is no fallback)."""
    global _loaded
    if _loaded:
        return
    if not os.path.exists(TORCH_LIB_PATH):
        raise ImportError(f"{TORCH_LIB_PATH} not found: build it with `python -m xequinet_amd.csrc.build_torch`")
    lib.load()                                  # libxeq_hip.so first: the extension resolves the C ABI against it
    torch.ops.load_library(TORCH_LIB_PATH)
    _loaded = True


def _flatten_params(model) -> List[torch.Tensor]:
    """The flat parameter list of ``xeq::xpainn_eval`` (layout: csrc/xeq_torch.cpp), from an ``nn.XPaiNN``."""
    from ..nn.fused import _packed_uv

    emb = model.mods["embedding"]
    dev = next(model.parameters()).device
    dt = next(p for p in model.parameters() if p.is_floating_point()).dtype
    empty = torch.empty(0, dtype=dt, device=dev)
    out: List[torch.Tensor] = []
    if isinstance(emb.embedding, nn.Embedding):
        out += [emb.embedding.weight, empty, empty]
    else:
        out += [emb.embedding[0].embed_ten, emb.embedding[1].weight, emb.embedding[1].bias]
    p0, p1 = emb.rbf.params()
    out += [p0.reshape(-1), empty if p1 is None else p1.reshape(-1)]
    n_blocks = sum(1 for k in model.mods if k.startswith("message_"))
    for i in range(n_blocks):
        m, u = model.mods[f"message_{i}"], model.mods[f"update_{i}"]
        ln = isinstance(m.norm, nn.LayerNorm)
        out += [m.scalar_mlp[0].weight, m.scalar_mlp[0].bias, m.scalar_mlp[2].weight, m.scalar_mlp[2].bias, m.rbf_lin.weight,
                m.rbf_lin.bias]
        out += [m.norm.weight, m.norm.bias, m.o3norm.affine_weight, m.o3norm.affine_bias] if ln else [empty] * 4
        packs, bias = _packed_uv(u)
        by_l = {l: W for (mul, l, _, _), W in zip(u.node_irreps.blocks(), packs)}
        out += [by_l.get(l, empty) for l in range(3)] + [empty if bias is None else bias]
        out += [u.dot_lin.weight, u.update_mlp[0].weight, u.update_mlp[0].bias, u.update_mlp[2].weight, u.update_mlp[2].bias]
        out += [u.norm.weight, u.norm.bias, u.o3norm.affine_weight, u.o3norm.affine_bias] if ln else [empty] * 4
        out += [empty] * 4
    head = model.mods["output_energy"].out_mlp
    out += [head[0].weight, head[0].bias, head[2].weight, head[2].bias]
    return [t.detach().to(dt).contiguous() for t in out]


class XPaiNNNative(nn.Module):
    """A whole XPaiNN energy (+ forces, + virial) evaluation as ONE registered operator.  Scriptable: the module holds one
    flat parameter buffer and the hyper-parameters; ``forward`` slices the buffer and calls ``torch.ops.xeq.xpainn_eval``."""

    def __init__(self, model) -> None:
        super().__init__()
        load_torch_library()
        from ..nn.model import XPaiNN

        if not isinstance(model, XPaiNN):
            raise TypeError("XPaiNNNative wraps an nn.XPaiNN")
        emb, msg0, upd0 = model.mods["embedding"], model.mods["message_0"], model.mods["update_0"]
        for k, m in model.mods.items():
            if hasattr(m, "scalar_mlp") and not isinstance(m.scalar_mlp[1], nn.SiLU):
                raise NotImplementedError("xeq::xpainn_eval implements the SiLU activation (the reference's default)")
        params = _flatten_params(model)
        self.shapes: List[List[int]] = [list(p.shape) for p in params]
        self.offsets: List[int] = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4            # every tensor starts on a 16-byte boundary
        flat = torch.zeros(max(off, 1), dtype=params[0].dtype, device=params[0].device)
        for p, o in zip(params, self.offsets):
            flat[o:o + p.numel()] = p.reshape(-1)
        self.register_buffer("flat", flat)
        # the embedding table is float64 in the reference's loader only until cast to the default dtype: one dtype here
        mul = msg0._mul
        n_blocks = sum(1 for k in model.mods if k.startswith("message_"))
        self.iparams: List[int] = [int(msg0.node_dim), int(mul[0]), int(mul[1]), int(mul[2]), int(msg0.num_basis), int(n_blocks),
                                   int(lib.RBF_KINDS[emb.rbf.kind]), int(lib.CUTOFF_KINDS[emb.cutoff_fn.kind]),
                                   int(isinstance(msg0.norm, nn.LayerNorm)), int(isinstance(emb.embedding, nn.Embedding))]
        self.fparams: List[float] = [float(emb.cutoff_fn.cutoff), float(upd0.invariant.eps)]
        self.cutoff_radius: float = float(model.cutoff_radius)

    def forward(self, pos: torch.Tensor, atomic_numbers: torch.Tensor, edge_index: torch.Tensor, ptr: torch.Tensor,
                cell: Optional[torch.Tensor] = None, cell_offsets: Optional[torch.Tensor] = None, center_sorted: bool = False,
                symmetric: bool = False, compute_forces: bool = True, compute_virial: bool = False) -> List[torch.Tensor]:
        """-> [energy [G], atomic_energies [N], forces [N, 3] (empty unless asked), virial [G, 3, 3] (empty unless asked)]"""
        params: List[torch.Tensor] = []
        for i in range(len(self.offsets)):
            shape = self.shapes[i]
            n = 1
            for d in shape:
                n *= d
            params.append(self.flat[self.offsets[i]:self.offsets[i] + n].view(shape))
        return torch.ops.xeq.xpainn_eval(pos, atomic_numbers, edge_index, ptr, cell, cell_offsets, params, self.iparams,
                                         self.fparams, center_sorted, symmetric, compute_forces, compute_virial)


class XPaiNNLMPScript(nn.Module):
    """Scriptable XPaiNN for LAMMPS: the reference's call signature (interface/jit_model.py:40-89),
    ``forward(data, compute_forces, compute_virial) -> {energy, atomic_energies[, forces][, virial]}`` in LAMMPS units; ``data``
    holds ``atomic_numbers``, ``pos`` and the neighbour list LAMMPS built (``edge_index`` [+ ``cell``, ``cell_offsets``])."""

    def __init__(self, model, unit_style: str = "metal") -> None:
        super().__init__()
        self.core = XPaiNNNative(model)
        units = get_default_units()
        lmp = keys.LAMMPS_UNIT_STYLE[unit_style]
        self.pos_unit_factor: float = float(unit_conversion(lmp[keys.POSITIONS], units[keys.POSITIONS]))
        self.energy_unit_factor: float = float(unit_conversion(units[keys.TOTAL_ENERGY], lmp[keys.TOTAL_ENERGY]))
        self.forces_unit_factor: float = float(unit_conversion(units[keys.FORCES], lmp[keys.FORCES]))
        self.cutoff_radius: float = float(model.cutoff_radius) / self.pos_unit_factor

    def forward(self, data: Dict[str, torch.Tensor], compute_forces: bool = True,
                compute_virial: bool = False) -> Dict[str, torch.Tensor]:
        pos = data["pos"] * self.pos_unit_factor
        n = pos.shape[0]
        if "ptr" in data:
            ptr = data["ptr"]
        else:
            ptr = torch.tensor([0, n], dtype=torch.long, device=pos.device)
        cell: Optional[torch.Tensor] = None
        cell_offsets: Optional[torch.Tensor] = None
        if "cell" in data:
            cell = data["cell"]
            cell_offsets = data["cell_offsets"]
        out = self.core(pos, data["atomic_numbers"], data["edge_index"], ptr, cell, cell_offsets, False, False, compute_forces,
                        compute_virial)
        result: Dict[str, torch.Tensor] = {"energy": out[0] * self.energy_unit_factor, "atomic_energies": out[1]}
        if compute_forces:
            result["forces"] = out[2] * self.forces_unit_factor
        if compute_virial:
            result["virial"] = out[3] * self.energy_unit_factor
        return result


class XPaiNNGMXScript(nn.Module):
    """Scriptable XPaiNN for GROMACS' NNPot interface (interface/jit_model.py:148-216): positions in nm in, energy in kJ/mol
    out, the caller differentiates the energy (``xeq::xpainn_eval`` hands its forces to autograd).  The neighbour search runs
    inside the model under ``no_grad`` (jit_model.py:183-195): ``xeq::radius_graph_pbc`` for a periodic box -- the reference's
    ``single_radius_graph`` order and cell offsets, the same list the Python ``XPaiNNGMX`` searches, bit for bit --
    ``xeq::radius_graph`` for open boundaries."""

    def __init__(self, model) -> None:
        super().__init__()
        self.core = XPaiNNNative(model)
        units = get_default_units()
        self.pos_unit_factor: float = float(unit_conversion("nm", units[keys.POSITIONS]))
        self.energy_unit_factor: float = float(unit_conversion(units[keys.TOTAL_ENERGY], "kJ/mol"))
        self.cutoff_radius: float = float(model.cutoff_radius)

    def forward(self, positions: torch.Tensor, atomic_numbers: torch.Tensor, box: Optional[torch.Tensor] = None,
                pbc: Optional[torch.Tensor] = None) -> torch.Tensor:
        pos = positions * self.pos_unit_factor
        ptr = torch.tensor([0, pos.shape[0]], dtype=torch.long, device=pos.device)
        periodic = False
        if pbc is not None and box is not None:
            periodic = bool(pbc.any())
        if periodic:
            assert box is not None and pbc is not None
            cell = box * self.pos_unit_factor
            with torch.no_grad():
                edge_index, cell_offsets, _ = torch.ops.xeq.radius_graph_pbc(pos, cell, pbc, self.cutoff_radius)
            # (this package's own periodic search: center-sorted, every edge with its mirror image -- the operator's mirror map)
            out = self.core(pos, atomic_numbers, edge_index, ptr, cell.unsqueeze(0), cell_offsets, True, True, False, False)
        else:
            with torch.no_grad():
                edge_index, _ = torch.ops.xeq.radius_graph(pos, ptr, self.cutoff_radius)
            out = self.core(pos, atomic_numbers, edge_index, ptr, None, None, True, True, False, False)
        return out[0] * self.energy_unit_factor


def compile_model(model, mode: str = "lmp", unit_style: str = "metal", output_file: Optional[str] = None,
                  fusion_strategy: str = "DYNAMICS,3"):
    """run/jit_script.py:28-86 for a built model: script the MD front end and (with ``output_file``) save it with the
    reference's ``_extra_files`` (cutoff radius in engine units, fusion strategy, number of species, periodic table)."""
    if mode == "lmp":
        front = XPaiNNLMPScript(model, unit_style=unit_style)
    elif mode == "gmx":
        front = XPaiNNGMXScript(model)
    else:
        raise NotImplementedError(f"Unsupported mode {mode}")
    scripted = torch.jit.script(front.eval())
    if output_file is not None:
        n_species = ELEMENTS_LIST.index("Rn") + 1
        extra = {keys.CUTOFF_RADIUS: front.cutoff_radius, "fusion_strategy": fusion_strategy, keys.N_SPECIES: n_species,
                 keys.PERIODIC_TABLE: " ".join(ELEMENTS_LIST[:n_species])}
        scripted.save(output_file, _extra_files={k: str(v).encode("ascii") for k, v in extra.items()})
    return scripted

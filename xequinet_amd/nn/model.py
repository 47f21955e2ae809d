"""Model assembly -- mirror of ``xequinet/nn/model.py`` for the XPaiNN energy path:
``BaseModel.forward(data, compute_forces, compute_virial)`` (nn/model.py:26-46),
``XPaiNN`` (:49-122), ``resolve_model`` (:310-318), ``load_model`` (:321-351)."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Union

import torch
import torch.nn as nn

from . import training
from .basic import compute_edge_data, compute_properties
from .output import resolve_output
from .xpainn import XEmbedding, XPainnMessage, XPainnUpdate

# e3nn bookkeeping entries of reference checkpoints that have no counterpart here
# (weight-less TensorProducts register an empty `weight` and an `output_mask` buffer)
_E3NN_ONLY = (".rsh_conv.", ".tp.", ".scalar_mul.", ".sph_harm.", ".output_mask", "._", ".invariant.tp", ".equidot.tp")


class BaseModel(nn.Module):
    cutoff_radius: float

    def __init__(self) -> None:
        super().__init__()
        self.mods = nn.ModuleDict()
        self.extra_properties = []
        self.native_training = True   # False: an energy-only training pass takes the differentiable form too (nn/training.py)

    def forward(
        self,
        data: Dict[str, torch.Tensor],
        compute_forces: bool = True,
        compute_virial: bool = False,
    ) -> Dict[str, torch.Tensor]:
        # a training pass (train mode, parameters asking for gradients) whose result carries forces or a virial runs every block
        # in its differentiable form so that the force evaluation can itself be differentiated (create_graph=training,
        # nn/basic.py:143-159); with energies only (no second order) the blocks stay on the fused kernels and return their
        # parameter gradients themselves (nn/fused.py); everything else is the fused inference path
        train_pass = training.wants_training_pass(self)
        native = train_pass and not compute_forces and not compute_virial and self.native_training and training.native_pass_supported(self)
        data[training.PARAM_GRADS] = native
        train_pass = train_pass and not native
        data[training.TRAIN_PASS] = train_pass
        if train_pass:
            data = training.edge_data(data, compute_forces=compute_forces, compute_virial=compute_virial)
        else:
            data = compute_edge_data(data=data, compute_forces=compute_forces, compute_virial=compute_virial)
            # one edge-gradient launch for all message blocks of this evaluation (ops.EdgeGradDeferral): a fresh collector per call
            from .. import keys, ops

            g = data.get(keys.EDGE_GRAPH)
            if g is not None:
                g.edge_grad_deferral = ops.EdgeGradDeferral() if (compute_forces or compute_virial) and not native else None
        for mod in self.mods.values():
            data = mod(data)
        result = compute_properties(
            data=data,
            compute_forces=compute_forces,
            compute_virial=compute_virial,
            training=train_pass,
            extra_properties=self.extra_properties,
        )
        return result

    def load_reference_state_dict(self, state_dict: Dict[str, torch.Tensor]) -> None:
        """Load a reference checkpoint's ``ckpt['model']``: e3nn-only bookkeeping
        entries are dropped, everything else must match strictly."""
        own = self.state_dict()
        kept = {}
        for k, v in state_dict.items():
            if k in own:
                kept[k] = v.reshape(own[k].shape) if v.numel() == own[k].numel() else v
            elif not any(tag in k for tag in _E3NN_ONLY):
                raise KeyError(f"unexpected key in reference state dict: {k}")
        missing = [k for k in own if k not in kept]
        if missing:
            raise KeyError(f"missing keys in reference state dict: {missing}")
        self.load_state_dict(kept, strict=True)


class XPaiNN(BaseModel):
    """eXtended PaiNN (nn/model.py:49-122); charge/spin embeddings are off-path."""

    def __init__(self, **kwargs) -> None:
        super().__init__()
        node_dim: int = kwargs.get("node_dim", 128)
        node_irreps: str = kwargs.get("node_irreps", "128x0e + 64x1o + 32x2e")
        embed_basis: str = kwargs.get("embed_basis", "gfn2-xtb")
        aux_basis: str = kwargs.get("aux_basis", "aux56")
        num_basis: int = kwargs.get("num_basis", 20)
        rbf_kernel: str = kwargs.get("rbf_kernel", "bessel")
        cutoff: float = kwargs.get("cutoff", 5.0)
        cutoff_fn: str = kwargs.get("cutoff_fn", "cosine")
        action_blocks: int = kwargs.get("action_blocks", 3)
        activation: str = kwargs.get("activation", "silu")
        layer_norm: bool = kwargs.get("layer_norm", True)
        charge_embed: bool = kwargs.get("charge_embed", False)
        spin_embed: bool = kwargs.get("spin_embed", False)
        output_modes: Union[str, List[str]] = kwargs.get("output_modes", ["energy"])
        if charge_embed or spin_embed:
            raise NotImplementedError("charge/spin embeddings are outside the energy+force hot path (SURVEY 2)")

        self.cutoff_radius = cutoff
        self.mods["embedding"] = XEmbedding(
            node_dim=node_dim, node_irreps=node_irreps, embed_basis=embed_basis, aux_basis=aux_basis,
            num_basis=num_basis, rbf_kernel=rbf_kernel, cutoff=cutoff, cutoff_fn=cutoff_fn,
        )
        for i in range(action_blocks):
            self.mods[f"message_{i}"] = XPainnMessage(
                node_dim=node_dim, node_irreps=node_irreps, num_basis=num_basis, activation=activation, layer_norm=layer_norm,
            )
            self.mods[f"update_{i}"] = XPainnUpdate(
                node_dim=node_dim, node_irreps=node_irreps, activation=activation, layer_norm=layer_norm,
            )
        if action_blocks > 0:                # the embedding gathers the first message block's front half with the node scalars
            self.mods["embedding"]._next_message = [self.mods["message_0"]]
        for i in range(action_blocks - 1):   # an update block launches the front half of the message block behind it (nn/fused.py::NodeBlock)
            self.mods[f"update_{i}"]._next_message = [self.mods[f"message_{i + 1}"]]
        if output_modes is None:
            output_modes = ["energy"]
        elif isinstance(output_modes, str) or not isinstance(output_modes, Iterable):
            output_modes = [output_modes]
        for mode in output_modes:
            output = resolve_output(mode, **kwargs)
            self.mods[f"output_{mode}"] = output
            self.extra_properties.extend(output.extra_properties)
        # the heads built here read the node scalars only (nn/output.py:114-128), so the last update block's equivariant output has
        # no consumer: it is not computed (data[NODE_EQUIVARIANT] is None behind that block) and its gradient is not formed
        if action_blocks > 0 and all(not getattr(self.mods[f"output_{m}"], "reads_equivariant", False) for m in output_modes):
            self.mods[f"update_{action_blocks - 1}"].equivariant_output_unused = True


def resolve_model(model_name: str, **kwargs) -> BaseModel:
    models_factory = {"xpainn": XPaiNN}
    if model_name.lower() not in models_factory:
        raise NotImplementedError(f"Unsupported model {model_name}")
    return models_factory[model_name.lower()](**kwargs)


def load_model(ckpt_file: str, device: Optional[torch.device] = None):
    """nn/model.py:321-351: rebuild the model from ``ckpt['config']`` and wrap it
    with the neighbour transform."""
    from ..data import NeighborTransform

    class ModelWithTransform:
        def __init__(self, model, transform, device):
            self.model = model
            self.transform = transform
            self.device = device

        def __call__(self, data, **kwargs):
            data = data.to(self.device)
            data = self.transform(data)
            return self.model(data.to_dict(), **kwargs)

    if device is None:
        device = torch.device("cuda")
    ckpt = torch.load(ckpt_file, map_location=device)
    model_config = ckpt["config"]
    model = resolve_model(model_config["model_name"], **model_config["model_kwargs"]).to(device)
    model.load_reference_state_dict(ckpt["model"])
    model.eval()
    return ModelWithTransform(model, NeighborTransform(model.cutoff_radius), device)

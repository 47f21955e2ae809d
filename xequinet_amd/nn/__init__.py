from .basic import resolve_activation
from .model import load_model, resolve_model
from .output import resolve_output
from .rbf import resolve_cutoff, resolve_rbf

__all__ = ["resolve_model", "load_model", "resolve_output", "resolve_rbf", "resolve_cutoff", "resolve_activation"]

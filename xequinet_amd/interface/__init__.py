from .ase_calculator import XequiCalculator
from .md_model import XPaiNNGMX, XPaiNNLMP, resolve_jit_model

__all__ = ["XPaiNNLMP", "XPaiNNGMX", "resolve_jit_model", "XequiCalculator"]

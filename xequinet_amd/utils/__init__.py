from . import units
from .units import check_unit, eval_unit, get_default_units, set_default_units, unit_conversion

__all__ = ["units", "check_unit", "eval_unit", "get_default_units", "set_default_units", "unit_conversion"]

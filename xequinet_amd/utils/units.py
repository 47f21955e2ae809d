"""Unit algebra for the callers of the hot path (LAMMPS / GROMACS / ASE front ends).

Mirror of the reference's ``unit_conversion / set_default_units / get_default_units`` (utils/qc.py:106-148): a unit
string is an arithmetic expression over unit names (``"kcal/mol/Angstrom"``, ``"eV/Angstrom^3"``); every name has a
value in Hartree atomic units and ``unit_conversion(a, b) = value(a) / value(b)``.  The reference evaluates the
string with Python's ``eval`` after a token check (qc.py:80-103); here a small recursive-descent parser does the
same arithmetic (``^`` binds like ``**``) and nothing else.  Constants are CODATA 2018 (the set qc.py:17-31 names).
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional

from .. import keys

# ---- CODATA 2018, SI -------------------------------------------------------------------------------------------
_C = 299792458.0                 # speed of light (exact)
_H = 6.62607015e-34              # Planck constant (exact)
_QE = 1.602176634e-19            # elementary charge (exact)
_ME = 9.1093837015e-31           # electron mass
_NA = 6.02214076e23              # Avogadro number (exact)
_AMU = 1.66053906660e-27         # atomic mass unit
_MU0 = 4.0e-7 * math.pi          # vacuum permeability as the reference takes it (qc.py:19)

_EPS0 = 1.0 / (_MU0 * _C * _C)
_HBAR = _H / (2.0 * math.pi)
_K = 4.0 * math.pi * _EPS0       # 4 pi eps0
_BOHR_M = _K * _HBAR**2 / (_ME * _QE**2)            # Bohr radius in metres
_HARTREE_J = _ME * _QE**4 / (_K * _HBAR) ** 2       # Hartree in joules
_AUT_S = _HBAR / _HARTREE_J                         # atomic unit of time in seconds


def _table() -> Dict[str, float]:
    """value of one <unit> in atomic units"""
    t: Dict[str, float] = {}

    def put(value: float, *names: str) -> None:
        for n in names:
            t[n] = value

    put(1.0, "AU", "au", "e", "Bohr", "a0", "Hartree", "Ha", "Eh")
    put(_NA, "mol")
    put(1.0 / _QE, "Coulomb", "C")
    metre = 1.0 / _BOHR_M
    put(metre, "meter", "m")
    put(metre * 1e-10, "Angstrom", "Ang")
    put(metre * 1e-2, "cm")
    put(metre * 1e-10 * 10, "nm")
    put(1.0 / _AMU, "kg")
    put(1e-3 / _AMU, "g")
    joule = 1.0 / _HARTREE_J
    put(joule, "Joule", "J")
    put(joule * 1000, "kJoule", "kJ")
    put(joule * _QE, "eV")
    put(joule * _QE / 1000, "meV")
    put(joule * 4.184, "cal")
    put(joule * 4.184 * 1000, "kcal")
    # 1 Debye = 1e-21 / c  C m
    put(1e-21 / _C / _QE * metre, "Debye", "D")
    second = 1.0 / _AUT_S
    put(second, "second", "s")
    put(second * 1e-15, "fs")
    put(second * 1e-15 * 1000, "ps")
    pascal = joule / metre**3
    put(pascal, "Pascal", "Pa")
    put(pascal * 1e9, "GPa")
    put(pascal * 1e5, "bar")
    put(pascal * 1e5 * 1e3, "kbar")
    put(0.5, "Bohr_magneton", "muB")
    return t


units: Dict[str, float] = _table()

_TOKEN = re.compile(r"\s*(?:(\d+)|([A-Za-z_][A-Za-z_0-9]*)|(.))")


def _tokens(text: str) -> List[tuple]:
    out = []
    for num, name, op in _TOKEN.findall(text):
        if num:
            out.append(("num", float(int(num))))
        elif name:
            if name not in units:
                raise ValueError(f"Invalid unit {text}")
            out.append(("num", units[name]))
        elif op.strip():
            if op not in "+-*/^()":
                raise ValueError(f"Invalid unit {text}")
            out.append((op, None))
    return out


class _Parser:
    """expr := term (('+'|'-') term)* ; term := unary (('*'|'/') unary)* ; unary := ('+'|'-') unary | power ;
    power := atom ('^' unary)? ; atom := number | unit | '(' expr ')'   -- Python's precedence with ^ read as **"""

    def __init__(self, text: str) -> None:
        self.text = text
        self.tok = _tokens(text)
        self.i = 0

    def _peek(self) -> Optional[str]:
        return self.tok[self.i][0] if self.i < len(self.tok) else None

    def _fail(self):
        raise ValueError(f"Invalid unit {self.text}")

    def parse(self) -> float:
        if not self.tok:
            self._fail()
        v = self.expr()
        if self.i != len(self.tok):
            self._fail()
        return v

    def expr(self) -> float:
        v = self.term()
        while self._peek() in ("+", "-"):
            op = self.tok[self.i][0]
            self.i += 1
            r = self.term()
            v = v + r if op == "+" else v - r
        return v

    def term(self) -> float:
        v = self.unary()
        while self._peek() in ("*", "/"):
            op = self.tok[self.i][0]
            self.i += 1
            r = self.unary()
            v = v * r if op == "*" else v / r
        return v

    def unary(self) -> float:
        if self._peek() in ("+", "-"):
            op = self.tok[self.i][0]
            self.i += 1
            v = self.unary()
            return v if op == "+" else -v
        return self.power()

    def power(self) -> float:
        base = self.atom()
        if self._peek() == "^":
            self.i += 1
            return base ** self.unary()
        return base

    def atom(self) -> float:
        kind = self._peek()
        if kind == "num":
            v = self.tok[self.i][1]
            self.i += 1
            return v
        if kind == "(":
            self.i += 1
            v = self.expr()
            if self._peek() != ")":
                self._fail()
            self.i += 1
            return v
        self._fail()


def check_unit(unit: str) -> bool:
    try:
        _Parser(unit).parse()
    except (ValueError, ZeroDivisionError, OverflowError):
        return False
    return True


def eval_unit(unit: str) -> float:
    return _Parser(unit).parse()


def unit_conversion(unit_in: Optional[str], unit_out: Optional[str]) -> float:
    """Factor that takes a number in ``unit_in`` to ``unit_out`` (qc.py:106-114); 1 when either is None."""
    if unit_in is None or unit_out is None or unit_in == unit_out:
        return 1.0
    return eval_unit(unit_in) / eval_unit(unit_out)


DEFAULT_UNITS_MAP: Dict[str, str] = {keys.POSITIONS: "Angstrom"}


def set_default_units(unit_dict: Dict[str, str]) -> None:
    """qc.py:117-144 for the properties of this path: units are set for energy / positions; the force and virial
    units follow from them and cannot be set directly."""
    for prop, unit in unit_dict.items():
        if prop in keys.GRAD_PROPERTIES:
            raise ValueError("Please do not set units for gradient properties directly. "
                             "Set the units for the corresponding properties instead.")
        if not check_unit(unit):
            raise ValueError(f"Invalid unit {unit} for property {prop}")
    DEFAULT_UNITS_MAP.update(unit_dict)
    if keys.TOTAL_ENERGY in DEFAULT_UNITS_MAP:
        e, p = DEFAULT_UNITS_MAP[keys.TOTAL_ENERGY], DEFAULT_UNITS_MAP[keys.POSITIONS]
        DEFAULT_UNITS_MAP[keys.FORCES] = f"{e}/{p}"
        DEFAULT_UNITS_MAP[keys.VIRIAL] = f"{e}/{p}^3"


def get_default_units() -> Dict[str, str]:
    return DEFAULT_UNITS_MAP

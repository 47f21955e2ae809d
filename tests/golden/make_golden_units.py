"""Golden unit-conversion factors from the reference's own unit table (utils/qc.py:13-114).

Run once, here (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden_units.py

``utils/qc.py`` as a whole imports pyscf (absent here), so only its unit functions are executed: the module is
parsed, the definitions of ``gen_units_dict / check_unit / eval_unit / unit_conversion`` and the two statements
that publish the table are run in a scratch namespace.  The output is data: (unit_in, unit_out, factor) triples.
"""
import ast
import json
import os
import re
from math import pi
from typing import Dict, Optional

REF_QC = "/root/reference/xequinet/utils/qc.py"
HERE = os.path.dirname(os.path.abspath(__file__))

PAIRS = [
    ("Hartree", "eV"), ("Hartree", "kcal/mol"), ("Hartree", "kJ/mol"), ("eV", "kcal/mol"), ("eV", "meV"),
    ("Bohr", "Angstrom"), ("Angstrom", "Bohr"), ("nm", "Angstrom"), ("Angstrom", "nm"), ("cm", "Bohr"),
    ("Hartree/Bohr", "eV/Angstrom"), ("eV/Angstrom", "kcal/mol/Angstrom"), ("eV/Angstrom", "kJ/(mol*nm)"),
    ("kcal/mol/Angstrom", "Hartree/Bohr"), ("eV/Angstrom^3", "GPa"), ("Hartree/Bohr^3", "kbar"), ("eV/Angstrom^3", "bar"),
    ("e*Bohr", "Debye"), ("Debye", "e*Angstrom"), ("ps", "fs"), ("fs", "AU"), ("g/mol", "AU"), ("kg", "g"),
    ("muB", "AU"), ("2*eV", "eV"), ("eV/Angstrom^2", "Hartree/Bohr^2"), ("(kcal/mol)/Angstrom", "eV/Angstrom"),
    ("Coulomb", "e"), ("J", "cal"), ("Pa", "AU"),
]
INVALID = ["furlong", "eV/__import__", "eV;1", "os.system", "eV/parsec"]


def main():
    tree = ast.parse(open(REF_QC).read())
    wanted = {"gen_units_dict", "check_unit", "eval_unit", "unit_conversion"}
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in wanted:
            body.append(node)
        elif isinstance(node, ast.Assign) and getattr(node.targets[0], "id", None) == "units":
            body.append(node)
        elif isinstance(node, ast.Expr) and "globals().update(units)" in ast.unparse(node):
            body.append(node)
    ns = {"re": re, "pi": pi, "Optional": Optional, "Dict": Dict}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF_QC, "exec"), ns)
    out = {"pairs": [[a, b, ns["unit_conversion"](a, b)] for a, b in PAIRS],
           "table": {k: v for k, v in ns["units"].items()},
           "invalid": [u for u in INVALID if not ns["check_unit"](u)]}
    assert out["invalid"] == INVALID
    with open(os.path.join(HERE, "units.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(len(out["pairs"]), "pairs,", len(out["table"]), "table entries")


if __name__ == "__main__":
    main()

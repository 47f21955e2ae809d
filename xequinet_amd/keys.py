"""String / int keys of the data dictionary -- same names and values as the
reference's ``xequinet/keys.py:4-50`` so that data dicts are interchangeable."""
from typing import Dict, Final, Set

# basic keys in datapoints
POSITIONS: Final[str] = "pos"
ATOMIC_NUMBERS: Final[str] = "atomic_numbers"
EDGE_INDEX: Final[str] = "edge_index"
CELL_OFFSETS: Final[str] = "cell_offsets"
CELL: Final[str] = "cell"
PBC: Final[str] = "pbc"
# keys for collated batches
BATCH: Final[str] = "batch"
BATCH_PTR: Final[str] = "ptr"
NUM_GRAPHS: Final[str] = "num_graphs"

# intermediate variables
CENTER_IDX: Final[int] = 0
NEIGHBOR_IDX: Final[int] = 1
EDGE_LENGTH: Final[str] = "edge_length"
EDGE_VECTOR: Final[str] = "edge_vector"
STRAIN: Final[str] = "strain"

RADIAL_BASIS_FUNCTION: Final[str] = "radial_basis_function"
ENVELOPE_FUNCTION: Final[str] = "envelope_function"
SPHERICAL_HARMONICS: Final[str] = "spherical_harmonics"
NODE_INVARIANT: Final[str] = "node_invariant"
NODE_EQUIVARIANT: Final[str] = "node_equivariant"

# properties
ATOMIC_ENERGIES: Final[str] = "atomic_energies"
TOTAL_ENERGY: Final[str] = "energy"
FORCES: Final[str] = "forces"
VIRIAL: Final[str] = "virial"
ENERGY_PER_ATOM: Final[str] = "energy/atom"     # loss-only property (utils/loss.py:59-67)

TOTAL_CHARGE: Final[str] = "charge"

GRAD_PROPERTIES: Final[Set[str]] = {FORCES, VIRIAL}

# MD front ends (keys.py:96-120 of the reference): metadata names and the LAMMPS unit styles
CUTOFF_RADIUS: Final[str] = "cutoff_radius"
N_SPECIES: Final[str] = "n_species"
PERIODIC_TABLE: Final[str] = "periodic_table"
LAMMPS_UNIT_STYLE: Final[Dict[str, Dict[str, str]]] = {
    "metal": {TOTAL_ENERGY: "eV", POSITIONS: "Angstrom", FORCES: "eV/Angstrom", TOTAL_CHARGE: "e"},
    "real": {TOTAL_ENERGY: "kcal/mol", POSITIONS: "Angstrom", FORCES: "kcal/mol/Angstrom", TOTAL_CHARGE: "e"},
    "electron": {TOTAL_ENERGY: "Hartree", POSITIONS: "Bohr", FORCES: "Hartree/Bohr", TOTAL_CHARGE: "e"},
}

# private: destination-sorted CSR views of edge_index built once per batch by
# xequinet_amd.ops.EdgeGraph and shared by all message blocks
EDGE_GRAPH: Final[str] = "_xeq_edge_graph"

from .datapoint import XequiBatch
from .radius_graph import radius_graph_pbc, single_radius_graph, wrap_positions
from .transform import NeighborTransform

__all__ = ["XequiBatch", "NeighborTransform", "radius_graph_pbc", "single_radius_graph", "wrap_positions"]

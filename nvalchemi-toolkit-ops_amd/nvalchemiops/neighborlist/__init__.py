"""Neighbour-list API (reference: nvalchemiops/neighborlist/__init__.py:15-74) for the MI355X hot path."""
from nvalchemiops.neighborlist.batch_cell_list import (batch_build_cell_list, batch_cell_list, batch_query_cell_list,
                                                       estimate_batch_cell_list_sizes)
from nvalchemiops.neighborlist.buffers import tuned_neighbor_buffers
from nvalchemiops.neighborlist._engine import D3SearchContext, attach_dftd3_context, invalidate
from nvalchemiops.neighborlist.batch_naive import batch_naive_neighbor_list
from nvalchemiops.neighborlist.batch_naive_dual_cutoff import batch_naive_neighbor_list_dual_cutoff
from nvalchemiops.neighborlist.cell_list import build_cell_list, cell_list, estimate_cell_list_sizes, query_cell_list
from nvalchemiops.neighborlist.naive import naive_neighbor_list
from nvalchemiops.neighborlist.naive_dual_cutoff import naive_neighbor_list_dual_cutoff
from nvalchemiops.neighborlist.neighbor_utils import (NeighborOverflowError, allocate_cell_list, compute_naive_num_shifts,
                                                      estimate_max_neighbors, get_neighbor_list_from_neighbor_matrix)
from nvalchemiops.neighborlist.neighborlist import neighbor_list
from nvalchemiops.neighborlist.rebuild_detection import (cell_list_needs_rebuild, check_cell_list_rebuild_needed,
                                                         check_neighbor_list_rebuild_needed, neighbor_list_needs_rebuild)

__all__ = [
    "neighbor_list", "cell_list", "batch_cell_list", "naive_neighbor_list", "build_cell_list", "query_cell_list",
    "batch_build_cell_list", "batch_query_cell_list", "estimate_cell_list_sizes", "estimate_batch_cell_list_sizes",
    "estimate_max_neighbors", "allocate_cell_list", "get_neighbor_list_from_neighbor_matrix", "compute_naive_num_shifts",
    "NeighborOverflowError", "cell_list_needs_rebuild", "neighbor_list_needs_rebuild", "check_cell_list_rebuild_needed",
    "check_neighbor_list_rebuild_needed", "batch_naive_neighbor_list", "naive_neighbor_list_dual_cutoff",
    "batch_naive_neighbor_list_dual_cutoff",
    "tuned_neighbor_buffers",  # MI355X addition: output buffers chosen by a measured trial search (neighborlist/buffers.py)
    # MI355X additions around the packed companion of a padded matrix (neighborlist/_engine.py): a search that also sums the DFT-D3
    # coordination numbers, and the call that drops a companion after a raw-pointer edit of the matrix
    "D3SearchContext", "attach_dftd3_context", "invalidate",
]

"""Batched cell-list neighbour search -- drop-in for neighborlist/batch_cell_list.py of the reference
(`estimate_batch_cell_list_sizes` :659, `batch_build_cell_list` :1070, `batch_query_cell_list` :1139,
`batch_cell_list` :1229).  Same fused HIP pipeline as the single-system path; systems only enter through
`batch_idx[atom]`, `cell[system]`, `pbc[system]` (one launch for the whole batch).
"""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import _engine as E
from nvalchemiops.neighborlist.cell_list import _build_cache, _empty_result, _search
from nvalchemiops.neighborlist.neighbor_utils import estimate_max_neighbors


@C.eager
def estimate_batch_cell_list_sizes(cell: torch.Tensor, pbc: torch.Tensor, cutoff: float, max_nbins: int = 1000):
    """(sum over systems of the per-system cell count, neighbor_search_radius[B,3]) (batch_cell_list.py:36-99, 659-736)."""
    n_sys = cell.shape[0]
    dev = cell.device
    if n_sys == 0 or cutoff <= 0:
        return 1, torch.zeros((n_sys, 3), device=dev, dtype=torch.int32)
    C.require_device(cell, pbc)
    c = cell.contiguous()
    p = pbc.reshape(-1, 3).to(torch.bool).contiguous()
    ncells = torch.zeros(n_sys, dtype=torch.int32, device=dev)
    radius = torch.zeros((n_sys, 3), dtype=torch.int32, device=dev)
    rc = C.lib().mi_nl_estimate_sizes(C.ptr(c), C.ptr(p), n_sys, C.cdouble(cutoff), int(max_nbins), C.dtype_code(cell.dtype),
                                      C.ptr(ncells), C.ptr(radius), C.stream_of(c))
    C.check(rc, "mi_nl_estimate_sizes")
    return int(ncells.sum().item()), radius


@C.hybrid
def batch_build_cell_list(positions, cutoff, cell, pbc, batch_idx, cells_per_dimension, neighbor_search_radius, atom_periodic_shifts,
                          atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices, cell_atom_list) -> None:
    """Fill the caller's batch cache tensors in place (batch_cell_list.py:739-912, 1070-1136); traced as `nvalchemiops::batch_build_cell_list`."""
    if positions.shape[0] == 0 or cutoff <= 0:
        return
    if C.tracing():
        torch.ops.nvalchemiops.batch_build_cell_list(positions, cutoff, cell, pbc, batch_idx, cells_per_dimension, neighbor_search_radius,
                                                     atom_periodic_shifts, atom_to_cell_mapping, atoms_per_cell_count,
                                                     cell_atom_start_indices, cell_atom_list)
        return
    C.require_device(positions, cell, pbc, batch_idx)
    pos, c, p = E.canon_geometry(positions, cell, pbc)
    _build_cache(pos, c, p, C.i32(batch_idx), cutoff, cells_per_dimension, atom_periodic_shifts, atom_to_cell_mapping,
                 atoms_per_cell_count, cell_atom_start_indices, cell_atom_list)


@C.hybrid
def batch_query_cell_list(positions, cell, pbc, cutoff, batch_idx, cells_per_dimension, neighbor_search_radius, atom_periodic_shifts,
                          atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices, cell_atom_list, neighbor_matrix,
                          neighbor_matrix_shifts, num_neighbors, half_fill: bool = False) -> None:
    """Batch query into pre-filled outputs; note the reference's argument order (cell, pbc BEFORE cutoff):
    batch_cell_list.py:1139-1156.  Traced as `nvalchemiops::batch_query_cell_list`."""
    if positions.shape[0] == 0:
        return
    if C.tracing():
        torch.ops.nvalchemiops.batch_query_cell_list(positions, cell, pbc, cutoff, batch_idx, cells_per_dimension, neighbor_search_radius,
                                                     atom_periodic_shifts, atom_to_cell_mapping, atoms_per_cell_count,
                                                     cell_atom_start_indices, cell_atom_list, neighbor_matrix, neighbor_matrix_shifts,
                                                     num_neighbors, half_fill)
        return
    C.require_device(positions, cell, pbc, batch_idx, neighbor_matrix, neighbor_matrix_shifts, num_neighbors)
    pos, c, p = E.canon_geometry(positions, cell, pbc)
    E.neighbor_matrix(pos, c, p, C.i32(batch_idx), cutoff, neighbor_matrix.shape[1], 0, half_fill, neighbor_matrix,
                      neighbor_matrix_shifts, num_neighbors, pad=False)


@C.hybrid
def batch_cell_list(positions: torch.Tensor, cutoff: float, cell: torch.Tensor, pbc: torch.Tensor, batch_idx: torch.Tensor,
                    max_neighbors: int | None = None, half_fill: bool = False, fill_value: int | None = None,
                    return_neighbor_list: bool = False, neighbor_matrix: torch.Tensor | None = None,
                    neighbor_matrix_shifts: torch.Tensor | None = None, num_neighbors: torch.Tensor | None = None,
                    cells_per_dimension: torch.Tensor | None = None, neighbor_search_radius: torch.Tensor | None = None,
                    atom_periodic_shifts: torch.Tensor | None = None, atom_to_cell_mapping: torch.Tensor | None = None,
                    atoms_per_cell_count: torch.Tensor | None = None, cell_atom_start_indices: torch.Tensor | None = None,
                    cell_atom_list: torch.Tensor | None = None):
    """Neighbour matrix (or COO list) of a batch of independent systems (batch_cell_list.py:1229-1468)."""
    total_atoms = positions.shape[0]
    if total_atoms <= 0 or cutoff <= 0:
        return _empty_result(total_atoms, -1, return_neighbor_list, positions.device)  # -1: batch_cell_list.py:1369
    if not C.tracing():
        C.require_device(positions, cell, pbc, batch_idx)
    if max_neighbors is None and neighbor_matrix is None:
        max_neighbors = estimate_max_neighbors(cutoff)
    if fill_value is None:
        fill_value = total_atoms
    pos, c, p = E.canon_geometry(positions, cell, pbc)
    bi = C.i32(batch_idx)
    cache = (cells_per_dimension, neighbor_search_radius, atom_periodic_shifts, atom_to_cell_mapping, atoms_per_cell_count,
             cell_atom_start_indices, cell_atom_list)
    if all(t is not None for t in cache):
        batch_build_cell_list(pos, cutoff, c, p, bi, *cache)
    return _search(pos, c, p, bi, cutoff, max_neighbors, half_fill, fill_value, return_neighbor_list, neighbor_matrix,
                   neighbor_matrix_shifts, num_neighbors)


__all__ = ["estimate_batch_cell_list_sizes", "batch_build_cell_list", "batch_query_cell_list", "batch_cell_list"]

"""Single-system cell-list neighbour search -- drop-in for neighborlist/cell_list.py of the reference
(`estimate_cell_list_sizes` :639, `build_cell_list` :1037, `query_cell_list` :1108, `cell_list` :1195).

The search itself is the fused HIP pipeline of csrc/nlist.hip (own binning; wave64-per-atom compaction).  The
reference-format cache tensors (`cells_per_dimension`, `atom_periodic_shifts`, ...) are still produced by
`build_cell_list` with the reference's binning rule for callers that read them; `query_cell_list` accepts them for
signature compatibility and re-bins internally (result sets do not depend on the binning).
"""
from __future__ import annotations

import ctypes

import torch

from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import _engine as E
from nvalchemiops.neighborlist.neighbor_utils import (allocate_cell_list, estimate_max_neighbors,
                                                      get_neighbor_list_from_neighbor_matrix)

_CACHE_NAMES = ("cells_per_dimension", "neighbor_search_radius", "atom_periodic_shifts", "atom_to_cell_mapping",
                "atoms_per_cell_count", "cell_atom_start_indices", "cell_atom_list")


@C.eager
def estimate_cell_list_sizes(cell: torch.Tensor, pbc: torch.Tensor, cutoff: float, max_nbins: int = 1000):
    """(max_total_cells, neighbor_search_radius[3]) with the reference's rule: cells/dim = max(int(face/cutoff),1),
    radius = ceil(cutoff*cells/face), halve all dims until the product fits max_nbins (cell_list.py:35-99, 639-722)."""
    dev = cell.device
    if (cell.ndim == 3 and cell.shape[0] == 0) or cutoff <= 0:
        return 1, torch.zeros((3,), dtype=torch.int32, device=dev)
    C.require_device(cell, pbc)
    code = C.dtype_code(cell.dtype)
    c = (cell if cell.ndim == 3 else cell.unsqueeze(0)).contiguous()
    p = pbc.reshape(-1, 3).to(torch.bool).contiguous()
    ncells = torch.zeros(1, dtype=torch.int32, device=dev)
    radius = torch.zeros((3,), dtype=torch.int32, device=dev)
    rc = C.lib().mi_nl_estimate_sizes(C.ptr(c), C.ptr(p), 1, C.cdouble(cutoff), int(max_nbins), code, C.ptr(ncells), C.ptr(radius),
                                      C.stream_of(c))
    C.check(rc, "mi_nl_estimate_sizes")
    return int(ncells.item()), radius


def _build_cache(pos, cell, pbc, batch_idx, cutoff, cpd, shifts, mapping, counts, starts, atoms):
    n = pos.shape[0]
    ws = E.workspace(n, cell.shape[0], pos.dtype, pos.device)
    rc = C.lib().mi_nl_build_cell_cache(C.ptr(pos), n, C.ptr(cell), C.ptr(pbc), C.ptr(batch_idx), cell.shape[0], C.cdouble(cutoff),
                                        C.dtype_code(pos.dtype), counts.shape[0], C.ptr(cpd), C.ptr(shifts), C.ptr(mapping), C.ptr(counts),
                                        C.ptr(starts), C.ptr(atoms), C.ptr(ws), ctypes.c_size_t(ws.numel()), C.stream_of(pos))
    C.check(rc, "mi_nl_build_cell_cache")


@C.hybrid
def build_cell_list(positions, cutoff, cell, pbc, cells_per_dimension, neighbor_search_radius, atom_periodic_shifts,
                    atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices, cell_atom_list) -> None:
    """Fill the caller's cache tensors in place (cell_list.py:725-889, 1037-1105).  Capacity = atoms_per_cell_count.shape[0].
    Traced by `torch.compile` as ONE call of the mutating op `nvalchemiops::build_cell_list`."""
    if positions.shape[0] == 0 or cutoff <= 0:
        return
    if C.tracing():
        torch.ops.nvalchemiops.build_cell_list(positions, cutoff, cell, pbc, cells_per_dimension, neighbor_search_radius, atom_periodic_shifts,
                                               atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices, cell_atom_list)
        return
    C.require_device(positions, cell, pbc)
    pos, c, p = E.canon_geometry(positions, cell, pbc)
    _build_cache(pos, c, p, None, cutoff, cells_per_dimension, atom_periodic_shifts, atom_to_cell_mapping, atoms_per_cell_count,
                 cell_atom_start_indices, cell_atom_list)


@C.hybrid
def query_cell_list(positions, cutoff, cell, pbc, cells_per_dimension, neighbor_search_radius, atom_periodic_shifts,
                    atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices, cell_atom_list, neighbor_matrix,
                    neighbor_matrix_shifts, num_neighbors, half_fill: bool = False) -> None:
    """Write neighbours of the current positions into the caller's pre-filled outputs (cell_list.py:892-1034, 1108-1192).
    Traced by `torch.compile` as ONE call of the mutating op `nvalchemiops::query_cell_list`."""
    if positions.shape[0] == 0:
        return
    if C.tracing():
        torch.ops.nvalchemiops.query_cell_list(positions, cutoff, cell, pbc, cells_per_dimension, neighbor_search_radius, atom_periodic_shifts,
                                               atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices, cell_atom_list,
                                               neighbor_matrix, neighbor_matrix_shifts, num_neighbors, half_fill)
        return
    C.require_device(positions, cell, pbc, neighbor_matrix, neighbor_matrix_shifts, num_neighbors)
    pos, c, p = E.canon_geometry(positions, cell, pbc)
    E.neighbor_matrix(pos, c, p, None, cutoff, neighbor_matrix.shape[1], 0, half_fill, neighbor_matrix, neighbor_matrix_shifts,
                      num_neighbors, pad=False)


def _empty_result(total_atoms, fill_value, return_neighbor_list, device):
    i32 = dict(dtype=torch.int32, device=device)
    if return_neighbor_list:
        return torch.zeros((2, 0), **i32), torch.zeros((total_atoms + 1,), **i32), torch.zeros((0, 3), **i32)
    return (torch.full((total_atoms, 0), fill_value, **i32), torch.zeros((total_atoms,), **i32), torch.zeros((total_atoms, 0, 3), **i32))


def _search(pos, c, p, batch_idx, cutoff, max_neighbors, half_fill, fill_value, return_neighbor_list, neighbor_matrix,
            neighbor_matrix_shifts, num_neighbors):
    """Common tail of cell_list / batch_cell_list: padded matrix into (possibly caller-owned) buffers, or direct CSR."""
    n, dev = pos.shape[0], pos.device
    own_buffers = neighbor_matrix is None and neighbor_matrix_shifts is None and num_neighbors is None
    if return_neighbor_list and own_buffers:
        lst, nptr, sh, _ = E.neighbor_csr(pos, c, p, batch_idx, cutoff, half_fill, max_neighbors=max_neighbors)
        return lst, nptr, sh
    if neighbor_matrix is None:
        neighbor_matrix = torch.empty((n, max_neighbors), dtype=torch.int32, device=dev)
    m = neighbor_matrix.shape[1]
    if neighbor_matrix_shifts is None:
        neighbor_matrix_shifts = torch.empty((n, m, 3), dtype=torch.int32, device=dev)
    if num_neighbors is None:
        num_neighbors = torch.empty((n,), dtype=torch.int32, device=dev)
    # the kernel writes every slot of every row (hits, then padding): no separate fill_/zero_ pass
    E.neighbor_matrix(pos, c, p, batch_idx, cutoff, m, fill_value, half_fill, neighbor_matrix, neighbor_matrix_shifts, num_neighbors)
    if return_neighbor_list:
        return get_neighbor_list_from_neighbor_matrix(neighbor_matrix, num_neighbors=num_neighbors,
                                                      neighbor_shift_matrix=neighbor_matrix_shifts, fill_value=fill_value)
    return neighbor_matrix, num_neighbors, neighbor_matrix_shifts


@C.hybrid
def cell_list(positions: torch.Tensor, cutoff: float, cell: torch.Tensor, pbc: torch.Tensor, max_neighbors: int | None = None,
              half_fill: bool = False, fill_value: int | None = None, return_neighbor_list: bool = False,
              neighbor_matrix: torch.Tensor | None = None, neighbor_matrix_shifts: torch.Tensor | None = None,
              num_neighbors: torch.Tensor | None = None, cells_per_dimension: torch.Tensor | None = None,
              neighbor_search_radius: torch.Tensor | None = None, atom_periodic_shifts: torch.Tensor | None = None,
              atom_to_cell_mapping: torch.Tensor | None = None, atoms_per_cell_count: torch.Tensor | None = None,
              cell_atom_start_indices: torch.Tensor | None = None, cell_atom_list: torch.Tensor | None = None):
    """Neighbour matrix (or COO list) of one periodic/non-periodic system (cell_list.py:1195-1443).

    Returns ``(neighbor_matrix[N,M], num_neighbors[N], neighbor_matrix_shifts[N,M,3])`` or, with
    ``return_neighbor_list=True``, ``(neighbor_list[2,P], neighbor_ptr[N+1], neighbor_list_shifts[P,3])``."""
    total_atoms = positions.shape[0]
    if fill_value is None:
        fill_value = total_atoms
    if total_atoms <= 0 or cutoff <= 0:
        return _empty_result(total_atoms, fill_value, return_neighbor_list, positions.device)
    if not C.tracing():
        C.require_device(positions, cell, pbc)
    pos, c, p = E.canon_geometry(positions, cell, pbc)
    if max_neighbors is None and (neighbor_matrix is None or neighbor_matrix_shifts is None or num_neighbors is None):
        max_neighbors = estimate_max_neighbors(cutoff)
    cache = (cells_per_dimension, neighbor_search_radius, atom_periodic_shifts, atom_to_cell_mapping, atoms_per_cell_count,
             cell_atom_start_indices, cell_atom_list)
    if all(t is not None for t in cache):
        # caller-owned cache: refresh it in the reference's format (cell_list.py:1394-1417)
        build_cell_list(pos, cutoff, c, p, *cache)
    return _search(pos, c, p, None, cutoff, max_neighbors, half_fill, fill_value, return_neighbor_list, neighbor_matrix,
                   neighbor_matrix_shifts, num_neighbors)


__all__ = ["estimate_cell_list_sizes", "build_cell_list", "query_cell_list", "cell_list", "allocate_cell_list"]

"""Batched O(N^2)-semantics search -- drop-in for `batch_naive_neighbor_list` (neighborlist/batch_naive.py:480-763).

Same result semantics as `naive_neighbor_list` (naive distance expression, |S_d| <= ceil(cutoff/face_d) images,
no wrapping), evaluated per system by the batched wave64 HIP pipeline (flag MI_NL_NAIVE_EXPR).
"""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import _engine as E
from nvalchemiops.neighborlist.naive import _bounding_cell
from nvalchemiops.neighborlist.neighbor_utils import (_prepare_batch_idx_ptr, estimate_max_neighbors,
                                                      get_neighbor_list_from_neighbor_matrix)


@C.hybrid
def batch_naive_neighbor_list(positions: torch.Tensor, cutoff: float, batch_idx: torch.Tensor | None = None,
                              batch_ptr: torch.Tensor | None = None, pbc: torch.Tensor | None = None, cell: torch.Tensor | None = None,
                              max_neighbors: int | None = None, half_fill: bool = False, fill_value: int | None = None,
                              return_neighbor_list: bool = False, neighbor_matrix: torch.Tensor | None = None,
                              neighbor_matrix_shifts: torch.Tensor | None = None, num_neighbors: torch.Tensor | None = None,
                              shift_range_per_dimension: torch.Tensor | None = None, shift_offset: torch.Tensor | None = None,
                              total_shifts: int | None = None, max_atoms_per_system: int | None = None):
    """Return tuples as the reference: ``(matrix, num[, shifts])`` or ``(list, ptr[, shifts])``; shifts only with pbc
    (batch_naive.py:730-763).  The shift-table arguments are accepted for signature parity; the search derives the
    image range on the device."""
    n, dev = positions.shape[0], positions.device
    if pbc is None and cell is not None:
        raise ValueError("If cell is provided, pbc must also be provided")
    if pbc is not None and cell is None:
        raise ValueError("If pbc is provided, cell must also be provided")
    batch_idx, batch_ptr = _prepare_batch_idx_ptr(batch_idx, batch_ptr, n, dev)
    periodic = pbc is not None
    if fill_value is None:
        fill_value = n
    if max_neighbors is None and (neighbor_matrix is None or (neighbor_matrix_shifts is None and periodic) or num_neighbors is None):
        max_neighbors = estimate_max_neighbors(cutoff)
    i32 = dict(dtype=torch.int32, device=dev)
    if neighbor_matrix is None:
        neighbor_matrix = torch.empty((n, max_neighbors), **i32)
    m = neighbor_matrix.shape[1]
    if num_neighbors is None:
        num_neighbors = torch.empty((n,), **i32)
    if periodic and neighbor_matrix_shifts is None:
        neighbor_matrix_shifts = torch.empty((n, m, 3), **i32)
    if n > 0 and cutoff > 0:
        if not C.tracing():
            C.require_device(positions, cell, pbc, batch_idx)
        bi = C.i32(batch_idx)
        origin = None
        if periodic:
            pos, c, p = E.canon_geometry(positions, cell, pbc)
        else:
            pos = E.canon_positions(positions)
            n_sys = batch_ptr.shape[0] - 1
            c, origin = _bounding_cell(pos, bi, n_sys)
            p = torch.zeros((n_sys, 3), dtype=torch.bool, device=dev)
        E.neighbor_matrix(pos, c, p, bi, cutoff, m, fill_value, half_fill, neighbor_matrix, neighbor_matrix_shifts if periodic else None,
                          num_neighbors, naive=True, want_shifts=periodic, origin=origin)
    else:
        neighbor_matrix.fill_(fill_value)
        num_neighbors.zero_()
        if periodic:
            neighbor_matrix_shifts.zero_()
    if return_neighbor_list:
        return get_neighbor_list_from_neighbor_matrix(neighbor_matrix, num_neighbors=num_neighbors,
                                                      neighbor_shift_matrix=neighbor_matrix_shifts if periodic else None,
                                                      fill_value=fill_value)
    return (neighbor_matrix, num_neighbors, neighbor_matrix_shifts) if periodic else (neighbor_matrix, num_neighbors)


__all__ = ["batch_naive_neighbor_list"]

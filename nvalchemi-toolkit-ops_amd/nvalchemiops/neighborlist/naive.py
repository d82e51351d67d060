"""O(N^2)-semantics neighbour search -- drop-in for `naive_neighbor_list` (neighborlist/naive.py:400-706).

Result semantics of the reference's naive kernels (naive.py:37-182): cutoff^2 is squared in double and then cast,
the distance is evaluated as (S.cell + r_i) - r_j for the upper half of the image shifts, images are limited to
|S_d| <= ceil(cutoff/face_d), positions are NOT wrapped.  The work itself runs through the same wave64 HIP
pipeline as the cell list (flag MI_NL_NAIVE_EXPR) -- an O(N) search that reproduces the O(N^2) answer.
"""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import _engine as E
from nvalchemiops.neighborlist.neighbor_utils import estimate_max_neighbors, get_neighbor_list_from_neighbor_matrix


def _bounding_cell(pos: torch.Tensor, batch_idx=None, n_sys: int = 1):
    """Non-periodic input has no cell: bin inside the per-system axis-aligned bounding box (pbc = F,F,F; shifts stay zero).
    One small HIP kernel pair (`mi_nl_bounding_cells`), no host sync.  Returns (cell[B,3,3], origin[B,3]) in the positions dtype."""
    if C.tracing():
        return torch.ops.nvalchemiops.bounding_cells(pos, batch_idx, n_sys)
    dev = pos.device
    cell = torch.empty((n_sys, 3, 3), dtype=pos.dtype, device=dev)
    origin = torch.empty((n_sys, 3), dtype=pos.dtype, device=dev)
    scratch = torch.empty(6 * n_sys, dtype=torch.int64, device=dev)
    rc = C.lib().mi_nl_bounding_cells(C.ptr(pos), C.ptr(batch_idx), pos.shape[0], n_sys, C.dtype_code(pos.dtype), C.ptr(cell), C.ptr(origin),
                                      C.ptr(scratch), C.stream_of(pos))
    C.check(rc, "mi_nl_bounding_cells")
    return cell, origin


@C.hybrid
def naive_neighbor_list(positions: torch.Tensor, cutoff: float, cell: torch.Tensor | None = None, pbc: torch.Tensor | None = None,
                        max_neighbors: int | None = None, half_fill: bool = False, fill_value: int | None = None,
                        return_neighbor_list: bool = False, neighbor_matrix: torch.Tensor | None = None,
                        neighbor_matrix_shifts: torch.Tensor | None = None, num_neighbors: torch.Tensor | None = None,
                        shift_range_per_dimension: torch.Tensor | None = None, shift_offset: torch.Tensor | None = None,
                        total_shifts: int | None = None):
    """Same arguments and return tuples as the reference: without pbc the shifts element is omitted
    (naive.py:664-680); with pbc it is present (:681-706)."""
    if pbc is None and cell is not None:
        raise ValueError("If cell is provided, pbc must also be provided")
    if pbc is not None and cell is None:
        raise ValueError("If pbc is provided, cell must also be provided")
    n, dev = positions.shape[0], positions.device
    periodic = pbc is not None
    origin = None
    if max_neighbors is None and (neighbor_matrix is None or (neighbor_matrix_shifts is None and periodic) or num_neighbors is None):
        max_neighbors = estimate_max_neighbors(cutoff)
    if fill_value is None:
        fill_value = n
    i32 = dict(dtype=torch.int32, device=dev)
    if cutoff <= 0 or n == 0:
        if neighbor_matrix is None:
            neighbor_matrix = torch.full((n, max_neighbors), fill_value, **i32)
        else:
            neighbor_matrix.fill_(fill_value)
        num_neighbors = torch.zeros(n, **i32) if num_neighbors is None else num_neighbors.zero_()
        if periodic:
            neighbor_matrix_shifts = (torch.zeros((n, neighbor_matrix.shape[1], 3), **i32) if neighbor_matrix_shifts is None
                                      else neighbor_matrix_shifts.zero_())
        if return_neighbor_list and cutoff <= 0:  # the reference's own shapes for this case (naive.py:627-657)
            out = (torch.zeros((2, 0), **i32), torch.zeros((n,), **i32), torch.zeros((n + 1,), **i32))
            return out + (torch.zeros((0, 3), **i32),) if periodic else out
        if return_neighbor_list:  # no atoms: the regular COO conversion of an empty matrix (test_neighborlist.py:905-921)
            out = (torch.zeros((2, 0), **i32), torch.zeros((n + 1,), **i32))
            return out + (torch.zeros((0, 3), **i32),) if periodic else out
        return (neighbor_matrix, num_neighbors, neighbor_matrix_shifts) if periodic else (neighbor_matrix, num_neighbors)
    if not C.tracing():
        C.require_device(positions, cell, pbc)
    if periodic:
        pos, c, p = E.canon_geometry(positions, cell, pbc)
    else:
        pos = E.canon_positions(positions)
        c, origin = _bounding_cell(pos)  # origin shifts the BINNING only; distances use the caller's coordinates
        p = torch.zeros((1, 3), dtype=torch.bool, device=dev)
    if neighbor_matrix is None:
        neighbor_matrix = torch.empty((n, max_neighbors), **i32)
    m = neighbor_matrix.shape[1]
    if num_neighbors is None:
        num_neighbors = torch.empty((n,), **i32)
    if periodic and neighbor_matrix_shifts is None:
        neighbor_matrix_shifts = torch.empty((n, m, 3), **i32)
    E.neighbor_matrix(pos, c, p, None, cutoff, m, fill_value, half_fill, neighbor_matrix, neighbor_matrix_shifts if periodic else None,
                      num_neighbors, naive=True, want_shifts=periodic, origin=origin)
    if not return_neighbor_list:
        return (neighbor_matrix, num_neighbors, neighbor_matrix_shifts) if periodic else (neighbor_matrix, num_neighbors)
    return get_neighbor_list_from_neighbor_matrix(neighbor_matrix, num_neighbors=num_neighbors,
                                                  neighbor_shift_matrix=neighbor_matrix_shifts if periodic else None,
                                                  fill_value=fill_value)


__all__ = ["naive_neighbor_list"]

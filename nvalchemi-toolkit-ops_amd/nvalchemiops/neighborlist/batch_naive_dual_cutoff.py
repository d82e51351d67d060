"""Batched two-cutoff search -- drop-in for `batch_naive_neighbor_list_dual_cutoff`
(neighborlist/batch_naive_dual_cutoff.py:592-900).  See naive_dual_cutoff.py for the evaluation strategy."""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C
from nvalchemiops.neighborlist.neighbor_utils import estimate_max_neighbors
from nvalchemiops.neighborlist.batch_naive import batch_naive_neighbor_list


@C.eager
def batch_naive_neighbor_list_dual_cutoff(positions: torch.Tensor, cutoff1: float, cutoff2: float, batch_idx: torch.Tensor | None = None,
                                          batch_ptr: torch.Tensor | None = None, pbc: torch.Tensor | None = None,
                                          cell: torch.Tensor | None = None, max_neighbors1: int | None = None,
                                          max_neighbors2: int | None = None, half_fill: bool = False, fill_value: int | None = None,
                                          return_neighbor_list: bool = False, neighbor_matrix1: torch.Tensor | None = None,
                                          neighbor_matrix2: torch.Tensor | None = None,
                                          neighbor_matrix_shifts1: torch.Tensor | None = None,
                                          neighbor_matrix_shifts2: torch.Tensor | None = None,
                                          num_neighbors1: torch.Tensor | None = None, num_neighbors2: torch.Tensor | None = None,
                                          shift_range_per_dimension: torch.Tensor | None = None,
                                          shift_offset: torch.Tensor | None = None, total_shifts: int | None = None,
                                          max_atoms_per_system: int | None = None):
    periodic = pbc is not None
    if max_neighbors1 is None and (neighbor_matrix1 is None or neighbor_matrix2 is None or num_neighbors1 is None or num_neighbors2 is None
                                   or (periodic and (neighbor_matrix_shifts1 is None or neighbor_matrix_shifts2 is None))):
        max_neighbors1 = max_neighbors2 = estimate_max_neighbors(cutoff2)  # reference default: both sized for the long cutoff
    if max_neighbors2 is None:
        max_neighbors2 = max_neighbors1
    common = dict(batch_idx=batch_idx, batch_ptr=batch_ptr, cell=cell, pbc=pbc, half_fill=half_fill, fill_value=fill_value,
                  return_neighbor_list=return_neighbor_list)
    r1 = batch_naive_neighbor_list(positions, cutoff1, max_neighbors=max_neighbors1, neighbor_matrix=neighbor_matrix1,
                                   neighbor_matrix_shifts=neighbor_matrix_shifts1, num_neighbors=num_neighbors1, **common)
    r2 = batch_naive_neighbor_list(positions, cutoff2, max_neighbors=max_neighbors2, neighbor_matrix=neighbor_matrix2,
                                   neighbor_matrix_shifts=neighbor_matrix_shifts2, num_neighbors=num_neighbors2, **common)
    return tuple(r1) + tuple(r2)


__all__ = ["batch_naive_neighbor_list_dual_cutoff"]

"""Batched two-cutoff search -- drop-in for `batch_naive_neighbor_list_dual_cutoff`
(neighborlist/batch_naive_dual_cutoff.py:592-900): one sweep per call fills both matrices (see naive_dual_cutoff.py)."""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C
from nvalchemiops.neighborlist.naive_dual_cutoff import _dual_cutoff
from nvalchemiops.neighborlist.neighbor_utils import _prepare_batch_idx_ptr


@C.hybrid
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
    batch_idx, batch_ptr = _prepare_batch_idx_ptr(batch_idx, batch_ptr, positions.shape[0], positions.device)
    return _dual_cutoff(positions, cutoff1, cutoff2, batch_idx, batch_ptr.shape[0] - 1, pbc, cell, max_neighbors1, max_neighbors2, half_fill,
                        fill_value, return_neighbor_list, neighbor_matrix1, neighbor_matrix2, neighbor_matrix_shifts1, neighbor_matrix_shifts2,
                        num_neighbors1, num_neighbors2)


__all__ = ["batch_naive_neighbor_list_dual_cutoff"]

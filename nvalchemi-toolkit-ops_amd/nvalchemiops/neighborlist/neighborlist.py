"""`neighbor_list()` dispatcher -- drop-in for neighborlist/neighborlist.py:41-310 of the reference.

Method selection follows the reference (:213-234): two cutoffs -> dual-cutoff naive; >= 5000 atoms -> cell list; else
naive; a `batch_` prefix when batch_idx / batch_ptr is given.  On this hot path `cell_list`, `batch_cell_list` and
`naive` are the HIP search; `batch_naive` is served by the batched HIP pipeline with the naive result semantics; the
dual-cutoff variants run that search once per cutoff.

One conscious deviation: when no cell is given for a >= 5000-atom input the reference fabricates a
UNIT cell with pbc=False, which degenerates to one bin (and indexes out of range for batches).  Here a per-system
bounding box is used for binning instead; the returned neighbour sets are the same.
"""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import _engine as E
from nvalchemiops.neighborlist.batch_cell_list import batch_cell_list
from nvalchemiops.neighborlist.batch_naive import batch_naive_neighbor_list
from nvalchemiops.neighborlist.batch_naive_dual_cutoff import batch_naive_neighbor_list_dual_cutoff
from nvalchemiops.neighborlist.cell_list import _empty_result, cell_list
from nvalchemiops.neighborlist.naive import _bounding_cell, naive_neighbor_list
from nvalchemiops.neighborlist.naive_dual_cutoff import naive_neighbor_list_dual_cutoff
from nvalchemiops.neighborlist.neighbor_utils import (_prepare_batch_idx_ptr, estimate_max_neighbors,
                                                      get_neighbor_list_from_neighbor_matrix)


def _free_space_cell_list(positions, cutoff, batch_idx, half_fill, fill_value, return_neighbor_list, kwargs, n_systems=None):
    """Cell-list search of non-periodic input without a cell: bounding-box binning, shifts are identically zero."""
    n = positions.shape[0]
    if fill_value is None:
        fill_value = n
    if n <= 0 or cutoff <= 0:
        return _empty_result(n, fill_value, return_neighbor_list, positions.device)
    if not C.tracing():
        C.require_device(positions, batch_idx)
    pos = E.canon_positions(positions)
    # the dispatcher already derived batch_ptr: its length gives the system count without another host sync
    n_sys = 1 if batch_idx is None else (n_systems if n_systems is not None else int(batch_idx.max().item()) + 1)
    bi = None if batch_idx is None else C.i32(batch_idx)
    cellb, origin = _bounding_cell(pos, bi, n_sys)
    pbc = torch.zeros((n_sys, 3), dtype=torch.bool, device=pos.device)
    max_neighbors = kwargs.get("max_neighbors")
    nm, nsh, num = kwargs.get("neighbor_matrix"), kwargs.get("neighbor_matrix_shifts"), kwargs.get("num_neighbors")
    if max_neighbors is None and nm is None:
        max_neighbors = estimate_max_neighbors(cutoff)
    if return_neighbor_list and nm is None and num is None:
        lst, nptr, sh, _ = E.neighbor_csr(pos, cellb, pbc, bi, cutoff, half_fill, max_neighbors=max_neighbors, origin=origin)
        return lst, nptr, sh
    dev = pos.device
    if nm is None:
        nm = torch.empty((n, max_neighbors), dtype=torch.int32, device=dev)
    if nsh is None:
        nsh = torch.empty((n, nm.shape[1], 3), dtype=torch.int32, device=dev)
    if num is None:
        num = torch.empty((n,), dtype=torch.int32, device=dev)
    E.neighbor_matrix(pos, cellb, pbc, bi, cutoff, nm.shape[1], fill_value, half_fill, nm, nsh, num, origin=origin)
    if return_neighbor_list:
        return get_neighbor_list_from_neighbor_matrix(nm, num_neighbors=num, neighbor_shift_matrix=nsh, fill_value=fill_value)
    return nm, num, nsh


@C.hybrid
def neighbor_list(positions: torch.Tensor, cutoff: float, cell: torch.Tensor | None = None, pbc: torch.Tensor | None = None,
                  batch_idx: torch.Tensor | None = None, batch_ptr: torch.Tensor | None = None, cutoff2: float | None = None,
                  half_fill: bool = False, fill_value: int | None = None, return_neighbor_list: bool = False,
                  method: str | None = None, **kwargs):
    """Neighbour matrix ``(neighbor_matrix, num_neighbors[, neighbor_matrix_shifts])`` or, with
    ``return_neighbor_list=True``, COO/CSR ``(neighbor_list[2,P], neighbor_ptr[N+1][, shifts[P,3]])``.
    The shifts element is present whenever ``pbc`` is given or a cell-list method runs (reference: neighborlist.py:146-190)."""
    free_space = False
    if method is None:
        total_atoms = positions.shape[0]
        if cutoff2 is not None:
            method = "naive_dual_cutoff"
        elif total_atoms >= 5000:
            method = "cell_list"
            free_space = cell is None or pbc is None
        else:
            method = "naive"
        if batch_idx is not None or batch_ptr is not None:
            method = "batch_" + method
            batch_idx, batch_ptr = _prepare_batch_idx_ptr(batch_idx, batch_ptr, total_atoms, positions.device)
    if method in ("cell_list", "batch_cell_list") and free_space:
        batched = method.startswith("batch_")
        return _free_space_cell_list(positions, cutoff, batch_idx if batched else None, half_fill, fill_value, return_neighbor_list, kwargs,
                                     n_systems=(batch_ptr.shape[0] - 1) if batched and batch_ptr is not None else None)
    if method == "naive":
        return naive_neighbor_list(positions, cutoff, pbc=pbc, cell=cell, half_fill=half_fill, fill_value=fill_value,
                                   return_neighbor_list=return_neighbor_list, **kwargs)
    if method == "cell_list":
        return cell_list(positions, cutoff, cell, pbc, half_fill=half_fill, fill_value=fill_value,
                         return_neighbor_list=return_neighbor_list, **kwargs)
    if method == "batch_cell_list":
        return batch_cell_list(positions, cutoff, cell, pbc, batch_idx, half_fill=half_fill, fill_value=fill_value,
                               return_neighbor_list=return_neighbor_list, **kwargs)
    if method == "batch_naive":
        if batch_idx is None or batch_ptr is None:
            batch_idx, batch_ptr = _prepare_batch_idx_ptr(batch_idx, batch_ptr, positions.shape[0], positions.device)
        return batch_naive_neighbor_list(positions, cutoff, pbc=pbc, cell=cell, batch_idx=batch_idx, batch_ptr=batch_ptr,
                                         half_fill=half_fill, fill_value=fill_value, return_neighbor_list=return_neighbor_list, **kwargs)
    if method == "naive_dual_cutoff":
        return naive_neighbor_list_dual_cutoff(positions, cutoff, cutoff2, pbc=pbc, cell=cell, half_fill=half_fill, fill_value=fill_value,
                                               return_neighbor_list=return_neighbor_list, **kwargs)
    if method == "batch_naive_dual_cutoff":
        return batch_naive_neighbor_list_dual_cutoff(positions, cutoff, cutoff2, pbc=pbc, cell=cell, batch_idx=batch_idx,
                                                     batch_ptr=batch_ptr, half_fill=half_fill, fill_value=fill_value,
                                                     return_neighbor_list=return_neighbor_list, **kwargs)
    raise ValueError(f"Invalid method: {method}")

"""Two cutoffs in one call -- drop-in for `naive_neighbor_list_dual_cutoff` (neighborlist/naive_dual_cutoff.py:544-860) and, through
`_dual_cutoff`, for `batch_naive_neighbor_list_dual_cutoff` (batch_naive_dual_cutoff.py:592-900).

As in the reference (naive_dual_cutoff.py:115-290) ONE sweep over the pair set fills both matrices: a pair inside `cutoff2` goes to
list 2 and, nested in that test, to list 1 if it is also inside `cutoff1`; both lists share the image range of `cutoff2`
(`compute_naive_num_shifts(cell, cutoff2, pbc)`, :835).  The sweep itself is the O(N) wave64 search of `naive_neighbor_list`
(`mi_nl_neighbors_dual`: one binning, one candidate walk, two compactions per 64 candidates).  With cutoff1 > cutoff2 the reference's
nesting leaves list 1 equal to list 2's pair set; that is reproduced by clamping the short cutoff.
Results are interleaved exactly as the reference returns them: (data1, num1[, shifts1], data2, num2[, shifts2]).
"""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import _engine as E
from nvalchemiops.neighborlist.naive import _bounding_cell
from nvalchemiops.neighborlist.neighbor_utils import estimate_max_neighbors, get_neighbor_list_from_neighbor_matrix


def _check_outputs(n, dev, nm, nsh, num, which):
    """Caller-supplied output buffers go to the kernel as raw pointers (rows written at i * M + slot, whole rows padded): anything but
    int32 / contiguous / the positions' device / N rows / matching widths would be an out-of-bounds device write, so it is an error here."""
    for name, t, shape in ((f"neighbor_matrix{which}", nm, (n, nm.shape[1] if nm.dim() == 2 else -1)), (f"num_neighbors{which}", num, (n,)),
                           (f"neighbor_matrix_shifts{which}", nsh, (n, nm.shape[1] if nm.dim() == 2 else -1, 3))):
        if t is None:
            continue
        if t.dtype != torch.int32 or not t.is_contiguous() or t.device != dev or tuple(t.shape) != shape:
            raise ValueError(f"{name} must be a contiguous int32 tensor of shape {shape} on {dev}, got {t.dtype} {tuple(t.shape)} on {t.device}"
                             f"{'' if t.is_contiguous() else ' (non-contiguous)'}")


def _dual_cutoff(positions, cutoff1, cutoff2, batch_idx, n_sys, pbc, cell, max_neighbors1, max_neighbors2, half_fill, fill_value,
                 return_neighbor_list, nm1, nm2, nsh1, nsh2, num1, num2):
    """Shared body of the single-system and the batched entry point (`batch_idx` None = one system)."""
    if pbc is None and cell is not None:
        raise ValueError("If cell is provided, pbc must also be provided")
    if pbc is not None and cell is None:
        raise ValueError("If pbc is provided, cell must also be provided")
    n, dev = positions.shape[0], positions.device
    periodic = pbc is not None
    if max_neighbors1 is None and (nm1 is None or nm2 is None or num1 is None or num2 is None or (periodic and (nsh1 is None or nsh2 is None))):
        max_neighbors1 = max_neighbors2 = estimate_max_neighbors(cutoff2)  # reference default: both sized for the long cutoff
    if max_neighbors2 is None:
        max_neighbors2 = max_neighbors1
    if fill_value is None:
        fill_value = n
    i32 = dict(dtype=torch.int32, device=dev)
    nm1 = torch.empty((n, max_neighbors1), **i32) if nm1 is None else nm1
    nm2 = torch.empty((n, max_neighbors2), **i32) if nm2 is None else nm2
    num1 = torch.empty((n,), **i32) if num1 is None else num1
    num2 = torch.empty((n,), **i32) if num2 is None else num2
    if periodic:
        nsh1 = torch.empty((n, nm1.shape[1], 3), **i32) if nsh1 is None else nsh1
        nsh2 = torch.empty((n, nm2.shape[1], 3), **i32) if nsh2 is None else nsh2
    _check_outputs(n, dev, nm1, nsh1 if periodic else None, num1, 1)
    _check_outputs(n, dev, nm2, nsh2 if periodic else None, num2, 2)
    if n > 0 and cutoff2 > 0 and cutoff1 > 0:
        if not C.tracing():
            C.require_device(positions, cell, pbc, batch_idx)
        bi = None if batch_idx is None else C.i32(batch_idx)
        origin = None
        if periodic:
            pos, c, p = E.canon_geometry(positions, cell, pbc)
        else:
            pos = E.canon_positions(positions)
            c, origin = _bounding_cell(pos, bi, n_sys)
            p = torch.zeros((n_sys, 3), dtype=torch.bool, device=dev)
        E.neighbor_matrix_dual(pos, c, p, bi, min(cutoff1, cutoff2), cutoff2, fill_value, half_fill, (nm1, nsh1 if periodic else None, num1),
                               (nm2, nsh2 if periodic else None, num2), naive=True, want_shifts=periodic, origin=origin)
    else:  # a non-positive cutoff selects nothing (naive.py:627-657); list 1 is nested in list 2
        for nm, num, nsh in ((nm1, num1, nsh1), (nm2, num2, nsh2)):
            nm.fill_(fill_value)
            num.zero_()
            if periodic:
                nsh.zero_()
        if n > 0 and cutoff2 > 0:  # cutoff1 <= 0 < cutoff2: only the long list has entries
            from nvalchemiops.neighborlist.batch_naive import batch_naive_neighbor_list
            from nvalchemiops.neighborlist.naive import naive_neighbor_list

            kw = dict(cell=cell, pbc=pbc, half_fill=half_fill, fill_value=fill_value, neighbor_matrix=nm2, neighbor_matrix_shifts=nsh2,
                      num_neighbors=num2)
            if batch_idx is None:
                naive_neighbor_list(positions, cutoff2, **kw)
            else:
                batch_naive_neighbor_list(positions, cutoff2, batch_idx=batch_idx, **kw)
    out = []
    for nm, num, nsh in ((nm1, num1, nsh1), (nm2, num2, nsh2)):
        if return_neighbor_list:
            out += list(get_neighbor_list_from_neighbor_matrix(nm, num_neighbors=num, neighbor_shift_matrix=nsh if periodic else None,
                                                               fill_value=fill_value))
        else:
            out += [nm, num] + ([nsh] if periodic else [])
    return tuple(out)


@C.hybrid
def naive_neighbor_list_dual_cutoff(positions: torch.Tensor, cutoff1: float, cutoff2: float, pbc: torch.Tensor | None = None,
                                    cell: torch.Tensor | None = None, max_neighbors1: int | None = None,
                                    max_neighbors2: int | None = None, half_fill: bool = False, fill_value: int | None = None,
                                    return_neighbor_list: bool = False, neighbor_matrix1: torch.Tensor | None = None,
                                    neighbor_matrix2: torch.Tensor | None = None, neighbor_matrix_shifts1: torch.Tensor | None = None,
                                    neighbor_matrix_shifts2: torch.Tensor | None = None, num_neighbors1: torch.Tensor | None = None,
                                    num_neighbors2: torch.Tensor | None = None, shift_range_per_dimension: torch.Tensor | None = None,
                                    shift_offset: torch.Tensor | None = None, total_shifts: int | None = None):
    """The shift-table arguments are accepted for signature parity; the search derives the image range (of `cutoff2`) on the device."""
    return _dual_cutoff(positions, cutoff1, cutoff2, None, 1, pbc, cell, max_neighbors1, max_neighbors2, half_fill, fill_value,
                        return_neighbor_list, neighbor_matrix1, neighbor_matrix2, neighbor_matrix_shifts1, neighbor_matrix_shifts2,
                        num_neighbors1, num_neighbors2)


__all__ = ["naive_neighbor_list_dual_cutoff"]

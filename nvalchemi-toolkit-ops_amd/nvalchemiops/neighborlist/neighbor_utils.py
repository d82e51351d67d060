"""Helpers shared by the neighbour-list entry points.

Reference counterparts: neighborlist/neighbor_utils.py:233-539 (`compute_naive_num_shifts`, `estimate_max_neighbors`,
`NeighborOverflowError`, `assert_max_neighbors`, `get_neighbor_list_from_neighbor_matrix`, `_prepare_batch_idx_ptr`,
`allocate_cell_list`).  Same names, arguments, return values and error behaviour.
"""
from __future__ import annotations

import ctypes
import math

import torch

from nvalchemiops import _capi as C


def estimate_max_neighbors(cutoff: float, atomic_density: float = 0.35, safety_factor: float = 5.0) -> int:
    """safety_factor x density x cutoff-sphere volume, rounded up to a multiple of 16 (neighbor_utils.py:296-340)."""
    if cutoff <= 0:
        return 0
    sphere = (4.0 / 3.0) * math.pi * cutoff**3
    expected = max(1, safety_factor * (atomic_density * sphere))
    return 16 * int(math.ceil(expected / 16))


class NeighborOverflowError(Exception):
    """More neighbours than the matrix has columns (neighbor_utils.py:343-349)."""

    def __init__(self, max_neighbors: int, num_neighbors: int):
        super().__init__(f"The number of neighbors is larger than the maximum allowed: {num_neighbors} > {max_neighbors}.")


def assert_max_neighbors(neighbor_matrix: torch.Tensor, num_neighbors: torch.Tensor) -> None:
    worst = 0 if num_neighbors.numel() == 0 else int(num_neighbors.max().item())
    if worst > neighbor_matrix.shape[1]:
        raise NeighborOverflowError(neighbor_matrix.shape[1], worst)


@C.eager
def get_neighbor_list_from_neighbor_matrix(neighbor_matrix: torch.Tensor, num_neighbors: torch.Tensor,
                                           neighbor_shift_matrix: torch.Tensor | None = None, fill_value: int = -1):
    """Padded matrix -> (neighbor_list[2,P], neighbor_ptr[N+1][, shifts[P,3]]) (neighbor_utils.py:362-441).

    One HIP kernel (wave per row, ballot compaction of entries != fill_value) instead of mask/where/index passes."""
    dev = neighbor_matrix.device
    n = num_neighbors.shape[0]
    if n == 0:
        lst = torch.zeros(2, 0, dtype=neighbor_matrix.dtype, device=dev)
        nptr = torch.zeros(1, dtype=torch.int32, device=dev)
        if neighbor_shift_matrix is None:
            return lst, nptr
        return lst, nptr, torch.empty(0, 2, 3, dtype=neighbor_shift_matrix.dtype, device=dev)
    C.require_device(neighbor_matrix, num_neighbors, neighbor_shift_matrix)
    assert_max_neighbors(neighbor_matrix, num_neighbors)
    nm = C.i32(neighbor_matrix)
    nptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    torch.cumsum(num_neighbors, dim=0, out=nptr[1:])
    # The list holds the entries != fill_value (the reference's boolean mask, :428-438) while neighbor_ptr is the cumsum of
    # `num_neighbors`; the two differ when the matrix was padded with another value than `fill_value` (e.g. the default -1 on a
    # matrix padded with N).  Rows are therefore placed by the mask's own counts, and both outputs are what the reference returns.
    place = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    torch.cumsum((nm != int(fill_value)).sum(dim=1, dtype=torch.int32), dim=0, out=place[1:])
    total = int(place[-1].item())
    lst = torch.empty((2, total), dtype=torch.int32, device=dev)
    sh_in = None if neighbor_shift_matrix is None else C.i32(neighbor_shift_matrix)
    sh = None if sh_in is None else torch.empty((total, 3), dtype=torch.int32, device=dev)
    rc = C.lib().mi_nl_matrix_to_coo(C.ptr(nm), C.ptr(sh_in), C.ptr(place), n, nm.shape[1], int(fill_value), C.ptr(lst), C.ptr(sh),
                                     ctypes.c_longlong(total), C.stream_of(nm))
    C.check(rc, "mi_nl_matrix_to_coo")
    lst = lst.to(neighbor_matrix.dtype)
    return (lst, nptr) if sh is None else (lst, nptr, sh)


def _prepare_batch_idx_ptr(batch_idx: torch.Tensor | None, batch_ptr: torch.Tensor | None, num_atoms: int, device: torch.device):
    """Derive whichever of batch_idx / batch_ptr is missing (neighbor_utils.py:444-491)."""
    if batch_idx is None and batch_ptr is None:
        raise ValueError("Either batch_idx or batch_ptr must be provided.")
    if batch_idx is None:
        per_system = batch_ptr[1:] - batch_ptr[:-1]
        batch_idx = torch.repeat_interleave(torch.arange(per_system.shape[0], dtype=torch.int32, device=device), per_system)
    elif batch_ptr is None:
        n_sys = int(batch_idx.max().item()) + 1 if batch_idx.numel() else 0
        per_system = torch.bincount(batch_idx, minlength=n_sys)
        batch_ptr = torch.zeros(n_sys + 1, dtype=torch.int32, device=device)
        torch.cumsum(per_system, dim=0, out=batch_ptr[1:])
    return batch_idx, batch_ptr


def allocate_cell_list(total_atoms: int, max_total_cells: int, neighbor_search_radius: torch.Tensor, device: torch.device):
    """Zeroed cache tensors in the reference's layout (neighbor_utils.py:494-539)."""
    lead = (3,) if neighbor_search_radius.ndim == 1 else (neighbor_search_radius.shape[0], 3)
    z = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=device)  # noqa: E731
    return (z(*lead), neighbor_search_radius, z(total_atoms, 3), z(total_atoms, 3), z(max_total_cells), z(max_total_cells), z(total_atoms))


def compute_naive_num_shifts(cell: torch.Tensor, cutoff: float, pbc: torch.Tensor):
    """Image range per dimension and shift-count prefix for the naive method (neighbor_utils.py:150-293).

    s_d = ceil(|column d of cell^-1| * cutoff) for periodic d; #shifts = s0*(2 s1+1)*(2 s2+1) + s1*(2 s2+1) + s2 + 1.
    Tiny (B x 3) host-side torch arithmetic; the kernels recompute the same ranges on the device."""
    inv = torch.linalg.inv(cell)
    dinv = torch.linalg.norm(inv, dim=-2) * pbc.to(cell.dtype)
    s = torch.ceil(dinv * torch.tensor(cutoff, dtype=cell.dtype, device=cell.device)).to(torch.int32)
    k1, k2 = 2 * s[:, 1] + 1, 2 * s[:, 2] + 1
    num_shifts = (s[:, 0] * k1 * k2 + s[:, 1] * k2 + s[:, 2] + 1).to(torch.int32)
    offset = torch.zeros(cell.shape[0] + 1, dtype=torch.int32, device=cell.device)
    torch.cumsum(num_shifts, dim=0, out=offset[1:])
    return s, offset, int(offset[-1].item())

"""Rebuild detection for MD loops -- drop-in for neighborlist/rebuild_detection.py (`cell_list_needs_rebuild` :336,
`neighbor_list_needs_rebuild` :457, `check_cell_list_rebuild_needed` :505, `check_neighbor_list_rebuild_needed` :579).

Both checks return a one-element bool tensor ON THE DEVICE (no host sync); the `check_*` conveniences call `.item()`.
"""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import _engine as E


@C.hybrid
def cell_list_needs_rebuild(current_positions: torch.Tensor, atom_to_cell_mapping: torch.Tensor, cells_per_dimension: torch.Tensor,
                            cell: torch.Tensor, pbc: torch.Tensor) -> torch.Tensor:
    """True when any atom now bins into a different cell than `atom_to_cell_mapping` (from `build_cell_list`) says.
    Traced by `torch.compile` as the op `nvalchemiops::_cell_list_needs_rebuild` (rebuild_detection.py:258)."""
    if C.tracing():
        return torch.ops.nvalchemiops._cell_list_needs_rebuild(current_positions, atom_to_cell_mapping, cells_per_dimension, cell, pbc)
    dev = current_positions.device
    flag = torch.zeros(1, dtype=torch.bool, device=dev)
    n = current_positions.shape[0]
    if n == 0:
        return flag
    C.require_device(current_positions, atom_to_cell_mapping, cells_per_dimension, cell, pbc)
    pos = E.canon_positions(current_positions)
    c = cell.detach().to(pos.dtype).reshape(-1, 3, 3)[0].contiguous()
    # converted tensors are bound to names so they outlive the enqueue (a freed temporary's block could be handed out again)
    amap, cpd, pb = C.i32(atom_to_cell_mapping), C.i32(cells_per_dimension.reshape(-1)), pbc.reshape(-1).to(torch.bool).contiguous()
    rc = C.lib().mi_nl_cells_changed(C.ptr(pos), C.ptr(c), C.ptr(amap), C.ptr(cpd), C.ptr(pb), n, C.dtype_code(pos.dtype), C.ptr(flag),
                                     C.stream_of(pos))
    C.check(rc, "mi_nl_cells_changed")
    return flag


@C.hybrid
def neighbor_list_needs_rebuild(reference_positions: torch.Tensor, current_positions: torch.Tensor,
                                skin_distance_threshold: float) -> torch.Tensor:
    """True when any atom moved farther than `skin_distance_threshold` from its position at list-build time
    (or when the two position arrays have different shapes).  Traced as `nvalchemiops::_neighbor_list_needs_rebuild` (:386)."""
    if C.tracing():
        return torch.ops.nvalchemiops._neighbor_list_needs_rebuild(reference_positions, current_positions, skin_distance_threshold)
    dev = current_positions.device
    if reference_positions.shape != current_positions.shape:
        return torch.tensor([True], device=dev, dtype=torch.bool)
    flag = torch.zeros(1, dtype=torch.bool, device=dev)
    n = reference_positions.shape[0]
    if n == 0:
        return flag
    C.require_device(reference_positions, current_positions)
    ref = E.canon_positions(reference_positions)
    cur = current_positions.detach().to(ref.dtype).contiguous()
    rc = C.lib().mi_nl_moved_beyond_skin(C.ptr(ref), C.ptr(cur), C.cdouble(skin_distance_threshold), n, C.dtype_code(ref.dtype), C.ptr(flag),
                                         C.stream_of(ref))
    C.check(rc, "mi_nl_moved_beyond_skin")
    return flag


@C.eager
def check_cell_list_rebuild_needed(cells_per_dimension, neighbor_search_radius, atom_periodic_shifts, atom_to_cell_mapping,
                                   atoms_per_cell_count, cell_atom_start_indices, cell_atom_list, current_positions, current_cell,
                                   current_pbc, cutoff: float) -> bool:
    """Host bool for a cached cell list (argument list of rebuild_detection.py:505-576)."""
    return bool(cell_list_needs_rebuild(current_positions, atom_to_cell_mapping, cells_per_dimension, current_cell, current_pbc).item())


@C.eager
def check_neighbor_list_rebuild_needed(reference_positions, current_positions, skin_distance_threshold: float) -> bool:
    return bool(neighbor_list_needs_rebuild(reference_positions, current_positions, skin_distance_threshold).item())


__all__ = ["cell_list_needs_rebuild", "neighbor_list_needs_rebuild", "check_cell_list_rebuild_needed", "check_neighbor_list_rebuild_needed"]

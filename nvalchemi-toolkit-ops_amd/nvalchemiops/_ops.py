"""`torch.library.custom_op` registration of the hot-path ops under the reference's names and argument lists, so that code which
reaches the seam directly -- `torch.ops.nvalchemiops.query_cell_list(...)`, or `torch.compile(fullgraph=True)` graphs that
need an opaque, mutation-annotated op -- finds the same ops as in the reference:

    nvalchemiops::build_cell_list / ::query_cell_list               (cell_list.py:725-736, 892-895)
    nvalchemiops::batch_build_cell_list / ::batch_query_cell_list   (batch_cell_list.py:739-749, 915-918)
    nvalchemiops::dftd3_nm / ::dftd3_nl                             (dftd3.py:1792-1795, 2125-2128)

    nvalchemiops::_cell_list_needs_rebuild / ::_neighbor_list_needs_rebuild   (rebuild_detection.py:258, :386)

The first six mutate caller-owned tensors and return None, as in the reference, so no fake implementation is needed; the two
rebuild checks return a fresh one-element bool tensor (fake registered).  Three ops of this build's own serve the high-level
entry points, whose search is ONE fused pipeline here instead of the reference's build + query pair:

    nvalchemiops::neighbor_search        padded matrix of any method (cell list / naive semantics, single / batch, +- shifts, padding)
    nvalchemiops::neighbor_search_dual   both matrices of the dual-cutoff methods in one sweep
    nvalchemiops::bounding_cells         binning frame of free-space input (returns cell[B,3,3], origin[B,3])

Outside a trace the Python functional API does NOT route through these (a custom-op dispatch costs tens of microseconds per
call, which matters on a path whose kernels are that short).  While TorchDynamo traces the caller it does: the public functions
are `_capi.hybrid` and branch on `torch.compiler.is_compiling()` into these ops, so `torch.compile(fullgraph=True)` under
Inductor captures `build_cell_list` / `query_cell_list` / `cell_list` / `neighbor_list` / `dftd3` as the reference's example does
(examples/neighborlist/04_neighbors_list_torch_compile_performance.py:323-346).  The op bodies call the same public functions
back: at run time nothing is being traced, so they take the ctypes path.  Registration happens on import of `nvalchemiops`.
"""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C
import importlib

# the sub-packages re-export functions named like their modules (`cell_list`, `dftd3`): import the MODULES explicitly
from nvalchemiops.neighborlist import _engine as E
from nvalchemiops.neighborlist import naive as _naive
from nvalchemiops.neighborlist import rebuild_detection as _rd

_d3 = importlib.import_module("nvalchemiops.interactions.dispersion.dftd3")
_bcl = importlib.import_module("nvalchemiops.neighborlist.batch_cell_list")
_cl = importlib.import_module("nvalchemiops.neighborlist.cell_list")

_CACHE = ("cells_per_dimension", "neighbor_search_radius", "atom_periodic_shifts", "atom_to_cell_mapping", "atoms_per_cell_count",
          "cell_atom_start_indices", "cell_atom_list")
_OUT = ("neighbor_matrix", "neighbor_matrix_shifts", "num_neighbors")


@torch.library.custom_op("nvalchemiops::build_cell_list", mutates_args=_CACHE)
def _build_cell_list_op(positions: torch.Tensor, cutoff: float, cell: torch.Tensor, pbc: torch.Tensor, cells_per_dimension: torch.Tensor,
                        neighbor_search_radius: torch.Tensor, atom_periodic_shifts: torch.Tensor, atom_to_cell_mapping: torch.Tensor,
                        atoms_per_cell_count: torch.Tensor, cell_atom_start_indices: torch.Tensor, cell_atom_list: torch.Tensor) -> None:
    _cl.build_cell_list(positions, cutoff, cell, pbc, cells_per_dimension, neighbor_search_radius, atom_periodic_shifts,
                        atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices, cell_atom_list)


@torch.library.custom_op("nvalchemiops::query_cell_list", mutates_args=_OUT)
def _query_cell_list_op(positions: torch.Tensor, cutoff: float, cell: torch.Tensor, pbc: torch.Tensor, cells_per_dimension: torch.Tensor,
                        neighbor_search_radius: torch.Tensor, atom_periodic_shifts: torch.Tensor, atom_to_cell_mapping: torch.Tensor,
                        atoms_per_cell_count: torch.Tensor, cell_atom_start_indices: torch.Tensor, cell_atom_list: torch.Tensor,
                        neighbor_matrix: torch.Tensor, neighbor_matrix_shifts: torch.Tensor, num_neighbors: torch.Tensor,
                        half_fill: bool = False) -> None:
    _cl.query_cell_list(positions, cutoff, cell, pbc, cells_per_dimension, neighbor_search_radius, atom_periodic_shifts,
                        atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices, cell_atom_list, neighbor_matrix,
                        neighbor_matrix_shifts, num_neighbors, half_fill)


@torch.library.custom_op("nvalchemiops::batch_build_cell_list", mutates_args=tuple(a for a in _CACHE if a != "neighbor_search_radius"))
def _batch_build_cell_list_op(positions: torch.Tensor, cutoff: float, cell: torch.Tensor, pbc: torch.Tensor, batch_idx: torch.Tensor,
                              cells_per_dimension: torch.Tensor, neighbor_search_radius: torch.Tensor,
                              atom_periodic_shifts: torch.Tensor, atom_to_cell_mapping: torch.Tensor, atoms_per_cell_count: torch.Tensor,
                              cell_atom_start_indices: torch.Tensor, cell_atom_list: torch.Tensor) -> None:
    _bcl.batch_build_cell_list(positions, cutoff, cell, pbc, batch_idx, cells_per_dimension, neighbor_search_radius,
                               atom_periodic_shifts, atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices, cell_atom_list)


@torch.library.custom_op("nvalchemiops::batch_query_cell_list", mutates_args=_OUT)
def _batch_query_cell_list_op(positions: torch.Tensor, cell: torch.Tensor, pbc: torch.Tensor, cutoff: float, batch_idx: torch.Tensor,
                              cells_per_dimension: torch.Tensor, neighbor_search_radius: torch.Tensor,
                              atom_periodic_shifts: torch.Tensor, atom_to_cell_mapping: torch.Tensor, atoms_per_cell_count: torch.Tensor,
                              cell_atom_start_indices: torch.Tensor, cell_atom_list: torch.Tensor, neighbor_matrix: torch.Tensor,
                              neighbor_matrix_shifts: torch.Tensor, num_neighbors: torch.Tensor, half_fill: bool = False) -> None:
    _bcl.batch_query_cell_list(positions, cell, pbc, cutoff, batch_idx, cells_per_dimension, neighbor_search_radius,
                               atom_periodic_shifts, atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices, cell_atom_list,
                               neighbor_matrix, neighbor_matrix_shifts, num_neighbors, half_fill)


def _device_of(t: torch.Tensor):
    """HIP launches go to the current device: guard for tensors on another GPU of the process."""
    return torch.cuda.device(t.device)


def _d3_scalars(a1, a2, s6, s8, k1, k3, on, off):
    return dict(a1=a1, a2=a2, s6=s6, s8=s8, k1=k1, k3=k3, s5_on=on, s5_off=off)


@torch.library.custom_op("nvalchemiops::dftd3_nm", mutates_args=("energy", "forces", "coord_num", "virial"))
def _dftd3_nm_op(positions: torch.Tensor, numbers: torch.Tensor, neighbor_matrix: torch.Tensor, covalent_radii: torch.Tensor,
                 r4r2: torch.Tensor, c6_reference: torch.Tensor, coord_num_ref: torch.Tensor, a1: float, a2: float, s8: float,
                 energy: torch.Tensor, forces: torch.Tensor, coord_num: torch.Tensor, virial: torch.Tensor, k1: float = 16.0,
                 k3: float = -4.0, s6: float = 1.0, s5_smoothing_on: float = 1e10, s5_smoothing_off: float = 1e10,
                 fill_value: int | None = None, batch_idx: torch.Tensor | None = None, cell: torch.Tensor | None = None,
                 neighbor_matrix_shifts: torch.Tensor | None = None, compute_virial: bool = False, device: str | None = None) -> None:
    n = positions.shape[0]
    if n == 0:
        return
    C.require_device(positions, numbers, neighbor_matrix, batch_idx, energy, forces, coord_num)
    nm = C.i32(neighbor_matrix)
    with _device_of(positions):
        _d3._launch(positions, numbers, nm, neighbor_matrix_shifts, None, nm.shape[1], n if fill_value is None else fill_value, cell, batch_idx,
                    energy.shape[0], (covalent_radii, r4r2, c6_reference, coord_num_ref),
                    _d3_scalars(a1, a2, s6, s8, k1, k3, s5_smoothing_on, s5_smoothing_off), compute_virial, energy, forces, coord_num, virial)


@torch.library.custom_op("nvalchemiops::dftd3_nl", mutates_args=("energy", "forces", "coord_num", "virial"))
def _dftd3_nl_op(positions: torch.Tensor, numbers: torch.Tensor, idx_j: torch.Tensor, neighbor_ptr: torch.Tensor,
                 covalent_radii: torch.Tensor, r4r2: torch.Tensor, c6_reference: torch.Tensor, coord_num_ref: torch.Tensor, a1: float,
                 a2: float, s8: float, energy: torch.Tensor, forces: torch.Tensor, coord_num: torch.Tensor, virial: torch.Tensor,
                 k1: float = 16.0, k3: float = -4.0, s6: float = 1.0, s5_smoothing_on: float = 1e10, s5_smoothing_off: float = 1e10,
                 batch_idx: torch.Tensor | None = None, cell: torch.Tensor | None = None, unit_shifts: torch.Tensor | None = None,
                 compute_virial: bool = False, device: str | None = None) -> None:
    if positions.shape[0] == 0:
        return
    C.require_device(positions, numbers, idx_j, neighbor_ptr, batch_idx, energy, forces, coord_num)
    with _device_of(positions):
        _d3._launch(positions, numbers, C.i32(idx_j), unit_shifts, C.i32(neighbor_ptr), 0, 0, cell, batch_idx, energy.shape[0],
                    (covalent_radii, r4r2, c6_reference, coord_num_ref), _d3_scalars(a1, a2, s6, s8, k1, k3, s5_smoothing_on, s5_smoothing_off),
                    compute_virial, energy, forces, coord_num, virial)


# ---- this build's own ops behind the high-level entry points -----------------------------------------------------------------

@torch.library.custom_op("nvalchemiops::neighbor_search", mutates_args=("neighbor_matrix", "neighbor_matrix_shifts", "num_neighbors"))
def _neighbor_search_op(positions: torch.Tensor, cell: torch.Tensor, pbc: torch.Tensor, batch_idx: torch.Tensor | None, cutoff: float,
                        flags: int, fill_value: int, neighbor_matrix: torch.Tensor, neighbor_matrix_shifts: torch.Tensor | None,
                        num_neighbors: torch.Tensor, origin: torch.Tensor | None = None) -> None:
    """The fused search (`mi_nl_neighbors`, mode matrix) into caller-owned outputs; `flags` = the NL_* bits of include/nvalchemiops_hip.h.
    Inputs are canonical already (`_engine.canon_geometry`): positions [N,3] f32/f64, cell [B,3,3] same dtype, pbc [B,3] bool."""
    C.require_device(positions, cell, pbc, batch_idx, neighbor_matrix, neighbor_matrix_shifts, num_neighbors, origin)
    pos, c, p = E.canon_geometry(positions, cell, pbc)
    with _device_of(pos):
        E.run(pos, c, p, None if batch_idx is None else C.i32(batch_idx), cutoff, C.NL_MATRIX, int(flags), nm=neighbor_matrix,
              nsh=neighbor_matrix_shifts, num=num_neighbors, max_neighbors=neighbor_matrix.shape[1], fill_value=int(fill_value),
              origin=None if origin is None else origin.contiguous())


@torch.library.custom_op("nvalchemiops::neighbor_search_dual",
                         mutates_args=("neighbor_matrix1", "neighbor_matrix_shifts1", "num_neighbors1", "neighbor_matrix2",
                                       "neighbor_matrix_shifts2", "num_neighbors2"))
def _neighbor_search_dual_op(positions: torch.Tensor, cell: torch.Tensor, pbc: torch.Tensor, batch_idx: torch.Tensor | None,
                             cutoff1: float, cutoff2: float, flags: int, fill_value: int, neighbor_matrix1: torch.Tensor,
                             neighbor_matrix_shifts1: torch.Tensor | None, num_neighbors1: torch.Tensor, neighbor_matrix2: torch.Tensor,
                             neighbor_matrix_shifts2: torch.Tensor | None, num_neighbors2: torch.Tensor,
                             origin: torch.Tensor | None = None) -> None:
    """Both matrices of a dual-cutoff search in one sweep (`mi_nl_neighbors_dual`; naive_dual_cutoff.py:283, :393 of the reference)."""
    C.require_device(positions, cell, pbc, batch_idx, neighbor_matrix1, num_neighbors1, neighbor_matrix2, num_neighbors2, origin)
    pos, c, p = E.canon_geometry(positions, cell, pbc)
    want_shifts = not (int(flags) & C.NL_NO_SHIFTS)
    with _device_of(pos):
        E.neighbor_matrix_dual(pos, c, p, None if batch_idx is None else C.i32(batch_idx), cutoff1, cutoff2, int(fill_value),
                               bool(int(flags) & C.NL_HALF_FILL), (neighbor_matrix1, neighbor_matrix_shifts1, num_neighbors1),
                               (neighbor_matrix2, neighbor_matrix_shifts2, num_neighbors2), naive=bool(int(flags) & C.NL_NAIVE_EXPR),
                               want_shifts=want_shifts, origin=None if origin is None else origin.contiguous())


@torch.library.custom_op("nvalchemiops::bounding_cells", mutates_args=())
def _bounding_cells_op(positions: torch.Tensor, batch_idx: torch.Tensor | None, n_systems: int) -> tuple[torch.Tensor, torch.Tensor]:
    C.require_device(positions, batch_idx)
    with _device_of(positions):
        return _naive._bounding_cell(positions.contiguous(), None if batch_idx is None else C.i32(batch_idx), int(n_systems))


@_bounding_cells_op.register_fake
def _(positions, batch_idx, n_systems):
    return positions.new_empty((n_systems, 3, 3)), positions.new_empty((n_systems, 3))


# ---- rebuild detection under the reference's op names ----------------------------------------------------------------------------

@torch.library.custom_op("nvalchemiops::_cell_list_needs_rebuild", mutates_args=())
def _cell_list_needs_rebuild_op(current_positions: torch.Tensor, atom_to_cell_mapping: torch.Tensor, cells_per_dimension: torch.Tensor,
                                cell: torch.Tensor, pbc: torch.Tensor) -> torch.Tensor:
    return _rd.cell_list_needs_rebuild(current_positions, atom_to_cell_mapping, cells_per_dimension, cell, pbc)


@_cell_list_needs_rebuild_op.register_fake
def _(current_positions, atom_to_cell_mapping, cells_per_dimension, cell, pbc):
    return current_positions.new_empty((1,), dtype=torch.bool)


@torch.library.custom_op("nvalchemiops::_neighbor_list_needs_rebuild", mutates_args=())
def _neighbor_list_needs_rebuild_op(reference_positions: torch.Tensor, current_positions: torch.Tensor,
                                    skin_distance_threshold: float) -> torch.Tensor:
    return _rd.neighbor_list_needs_rebuild(reference_positions, current_positions, skin_distance_threshold)


@_neighbor_list_needs_rebuild_op.register_fake
def _(reference_positions, current_positions, skin_distance_threshold):
    return current_positions.new_empty((1,), dtype=torch.bool)


__all__ = []

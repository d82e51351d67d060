"""`torch.library.custom_op` registration of the hot-path ops under the reference's names and argument lists, so that code which
reaches the seam directly -- `torch.ops.nvalchemiops.query_cell_list(...)`, or `torch.compile(fullgraph=True)` graphs that
need an opaque, mutation-annotated op -- finds the same ops as in the reference:

    nvalchemiops::build_cell_list / ::query_cell_list               (cell_list.py:725-736, 892-895)
    nvalchemiops::batch_build_cell_list / ::batch_query_cell_list   (batch_cell_list.py:739-749, 915-918)
    nvalchemiops::dftd3_nm / ::dftd3_nl                             (dftd3.py:1792-1795, 2125-2128)

All six mutate caller-owned tensors and return None, as in the reference, so no fake implementation is needed.  The Python
functional API does NOT route through these (a custom-op dispatch costs tens of microseconds per call, which matters on a
path whose kernels are that short); both enter the same ctypes layer.  Registration happens on first import of this module
(`import nvalchemiops._ops`, done by `nvalchemiops.neighborlist` / `.interactions.dispersion`).
"""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C
import importlib

# the sub-packages re-export functions named like their modules (`cell_list`, `dftd3`): import the MODULES explicitly
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
    nm = C.i32(neighbor_matrix)
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
    _d3._launch(positions, numbers, C.i32(idx_j), unit_shifts, C.i32(neighbor_ptr), 0, 0, cell, batch_idx, energy.shape[0],
                (covalent_radii, r4r2, c6_reference, coord_num_ref), _d3_scalars(a1, a2, s6, s8, k1, k3, s5_smoothing_on, s5_smoothing_off),
                compute_virial, energy, forces, coord_num, virial)


__all__ = []

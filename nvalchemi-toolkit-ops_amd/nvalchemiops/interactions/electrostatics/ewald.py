"""Real-space Ewald sum -- drop-in for `ewald_real_space` (interactions/electrostatics/ewald.py:2321-2628) and the 12
`alchemiops::_[batch_]ewald_real_space_*` ops behind it (:263-1365).

One HIP kernel family (csrc/ewald.hip, `mi_ewald_real`) covers matrix / CSR x single / batch x energy / +forces /
+charge gradients.  The neighbour list must be FULL (symmetric), as the reference's 1/2 prefactor assumes.
The explicit-k reciprocal half of ewald.py (`ewald_reciprocal_space`, `ewald_summation`) is outside this build's hot
path.
"""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C


def ewald_real_space(positions: torch.Tensor, charges: torch.Tensor, cell: torch.Tensor, alpha: torch.Tensor,
                     neighbor_list: torch.Tensor | None = None, neighbor_ptr: torch.Tensor | None = None,
                     neighbor_shifts: torch.Tensor | None = None, neighbor_matrix: torch.Tensor | None = None,
                     neighbor_matrix_shifts: torch.Tensor | None = None, mask_value: int = -1, batch_idx: torch.Tensor | None = None,
                     compute_forces: bool = False, compute_charge_gradients: bool = False):
    """E_i = 1/2 sum_j q_i q_j erfc(alpha r_ij)/r_ij over the listed neighbours (per-atom energies, input dtype).

    Returns ``energies`` | ``(energies, forces)`` | ``(energies, charge_grads)`` | ``(energies, forces, charge_grads)``."""
    if neighbor_list is None and neighbor_matrix is None:
        raise ValueError("Either neighbor_list or neighbor_matrix must be provided")
    if neighbor_list is not None and neighbor_ptr is None:
        raise ValueError("neighbor_ptr is required when using neighbor_list format")
    n, dev, dt = positions.shape[0], positions.device, positions.dtype
    code = C.dtype_code(dt)
    if n == 0:
        out = (torch.zeros(0, dtype=dt, device=dev),)
        if compute_forces:
            out += (torch.zeros((0, 3), dtype=dt, device=dev),)
        if compute_charge_gradients:
            out += (torch.zeros(0, dtype=dt, device=dev),)
        return out if len(out) > 1 else out[0]
    C.require_device(positions, charges, cell, neighbor_list, neighbor_ptr, neighbor_matrix, batch_idx)
    pos = positions.detach().contiguous()
    q = charges.detach().to(dt).contiguous()
    cells = cell.detach().to(dt).reshape(-1, 3, 3).contiguous()
    alpha_t = alpha if isinstance(alpha, torch.Tensor) else torch.tensor([float(alpha)], device=dev)
    alpha_t = alpha_t.detach().to(device=dev, dtype=dt).reshape(-1)
    if alpha_t.numel() == 1 and cells.shape[0] > 1:
        alpha_t = alpha_t.expand(cells.shape[0])
    alpha_t = alpha_t.contiguous()
    bi = None if batch_idx is None else C.i32(batch_idx)
    if neighbor_list is not None:
        idx, nptr, m = C.i32(neighbor_list[1]), C.i32(neighbor_ptr), 0
        sh = neighbor_shifts
        n_entries = idx.shape[0]
    else:
        idx, nptr, m = C.i32(neighbor_matrix), None, neighbor_matrix.shape[1]
        sh = neighbor_matrix_shifts
        n_entries = idx.numel()
    sh = torch.zeros((n_entries, 3), dtype=torch.int32, device=dev) if sh is None else C.i32(sh)
    energies = torch.empty(n, dtype=torch.float64, device=dev)
    forces = torch.empty((n, 3), dtype=dt, device=dev) if compute_forces else None
    cgrads = torch.empty(n, dtype=torch.float64, device=dev) if compute_charge_gradients else None
    if n_entries == 0:
        energies.zero_()
        forces = None if forces is None else forces.zero_()
        cgrads = None if cgrads is None else cgrads.zero_()
    else:
        flags = (C.EW_FORCES if compute_forces else 0) | (C.EW_CHARGE_GRAD if compute_charge_gradients else 0)
        rc = C.lib().mi_ewald_real(C.ptr(pos), C.ptr(q), C.ptr(cells), C.ptr(alpha_t), C.ptr(bi), n, code, C.ptr(idx), C.ptr(sh), C.ptr(nptr),
                                   int(m), int(mask_value), flags, C.ptr(energies), C.ptr(forces), C.ptr(cgrads), C.stream_of(pos))
        C.check(rc, "mi_ewald_real")
    out = (energies.to(dt),)  # ewald.py:577: energies are accumulated in float64 and returned in the input dtype
    if compute_forces:
        out += (forces,)
    if compute_charge_gradients:
        out += (cgrads.to(dt),)
    return out if len(out) > 1 else out[0]


__all__ = ["ewald_real_space"]

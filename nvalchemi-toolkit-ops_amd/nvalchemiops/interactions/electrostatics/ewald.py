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


class _EwaldRealEnergyFn(torch.autograd.Function):
    """Per-atom real-space energies with a hand-written adjoint kernel (`mi_ewald_real_bwd`) for positions, charges, cell and
    alpha -- the reference differentiates these ops through a recorded Warp tape (autograd.py:525-665)."""

    @staticmethod
    def forward(ctx, positions, charges, cells, alpha, idx, sh, nptr, m, mask_value, bi):
        pos, q = positions.detach().contiguous(), charges.detach().contiguous()
        c, al = cells.detach().contiguous(), alpha.detach().contiguous()
        n = pos.shape[0]
        energies = torch.empty(n, dtype=torch.float64, device=pos.device)
        rc = C.lib().mi_ewald_real(C.ptr(pos), C.ptr(q), C.ptr(c), C.ptr(al), C.ptr(bi), n, C.dtype_code(pos.dtype), C.ptr(idx), C.ptr(sh),
                                   C.ptr(nptr), int(m), int(mask_value), 0, C.ptr(energies), None, None, C.stream_of(pos))
        C.check(rc, "mi_ewald_real")
        empty = torch.empty(0, device=pos.device)
        ctx.save_for_backward(pos, q, c, al, idx, sh, nptr if nptr is not None else empty, bi if bi is not None else empty)
        ctx.meta = (m, mask_value, nptr is not None, bi is not None)
        return energies.to(pos.dtype)

    @staticmethod
    def backward(ctx, g_e):
        pos, q, c, al, idx, sh, nptr_t, bi_t = ctx.saved_tensors
        m, mask_value, has_ptr, has_bi = ctx.meta
        n, dt, dev = pos.shape[0], pos.dtype, pos.device
        g = g_e.detach().to(dt).contiguous()
        gpos = torch.empty((n, 3), dtype=dt, device=dev)
        gq = torch.empty(n, dtype=dt, device=dev)
        need = ctx.needs_input_grad
        gcell = torch.zeros(c.shape, dtype=torch.float64, device=dev) if need[2] else None
        galpha = torch.zeros(al.shape, dtype=torch.float64, device=dev) if need[3] else None
        rc = C.lib().mi_ewald_real_bwd(C.ptr(pos), C.ptr(q), C.ptr(c), C.ptr(al), C.ptr(bi_t if has_bi else None), n, C.dtype_code(dt),
                                       C.ptr(idx), C.ptr(sh), C.ptr(nptr_t if has_ptr else None), int(m), int(mask_value), C.ptr(g),
                                       C.ptr(gpos), C.ptr(gq), C.ptr(gcell), C.ptr(galpha), C.stream_of(pos))
        C.check(rc, "mi_ewald_real_bwd")
        return (gpos if need[0] else None, gq if need[1] else None, None if gcell is None else gcell.to(dt),
                None if galpha is None else galpha.to(dt), None, None, None, None, None, None)


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
    alpha_in = alpha if isinstance(alpha, torch.Tensor) else torch.tensor([float(alpha)], device=dev)
    alpha_in = alpha_in.to(device=dev, dtype=dt).reshape(-1)
    if alpha_in.numel() == 1 and cell.reshape(-1, 3, 3).shape[0] > 1:
        alpha_in = alpha_in.expand(cell.reshape(-1, 3, 3).shape[0])
    wants_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (positions, charges, cell, alpha_in))
    pos = positions.detach().contiguous()
    q = charges.detach().to(dt).contiguous()
    cells = cell.detach().to(dt).reshape(-1, 3, 3).contiguous()
    alpha_t = alpha_in.detach().contiguous()
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
    e_out = energies.to(dt)  # ewald.py:577: energies are accumulated in float64 and returned in the input dtype
    if wants_grad and n_entries > 0:
        # differentiable energies (explicit forces / charge gradients above stay plain outputs, as MD codes consume them)
        e_out = _EwaldRealEnergyFn.apply(positions, charges.to(dt), cell.to(dt).reshape(-1, 3, 3), alpha_in, idx, sh, nptr, m, mask_value, bi)
    out = (e_out,)
    if compute_forces:
        out += (forces,)
    if compute_charge_gradients:
        out += (cgrads.to(dt),)
    return out if len(out) > 1 else out[0]


__all__ = ["ewald_real_space"]

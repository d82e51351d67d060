"""Real-space Ewald sum -- drop-in for `ewald_real_space` (interactions/electrostatics/ewald.py:2321-2628) and the 12
`alchemiops::_[batch_]ewald_real_space_*` ops behind it (:263-1365).

One HIP kernel family (csrc/ewald.hip, `mi_ewald_real`) covers matrix / CSR x single / batch x energy / +forces /
+charge gradients.  Physically meaningful inputs are FULL (symmetric) lists -- the reference's 1/2 prefactor assumes one -- and those
take an owner-only fast path; any other list (half, truncated, one-sided) is detected on the device in the same pass and gets the
reference's i/j scatter (`ewald_kernels.py:518-544`), so results match the reference for every input.

The explicit-k reciprocal half (`ewald_reciprocal_space` :2631, `ewald_summation` :2798) runs on two HIP
kernels (`mi_ewald_structure_factors`, `mi_ewald_recip_gather`) that recompute the k.r phases in registers instead of the
reference's two float64 [K, N] phase tables.
"""
from __future__ import annotations

import torch

import ctypes
import math

import os

from nvalchemiops import _capi as C

# NVALCHEMIOPS_EWALD_TRUST_FULL_LISTS=0: always checksum the list for symmetry, also when it is provably the output of a full search (A/B, tests)
_TRUST_FULL_LISTS = os.environ.get("NVALCHEMIOPS_EWALD_TRUST_FULL_LISTS", "1") != "0"


def _real_space_inputs(positions, charges, cell, alpha, neighbor_list, neighbor_ptr, neighbor_shifts, neighbor_matrix, neighbor_matrix_shifts,
                       batch_idx):
    """Argument checks (ewald.py:2460-2470 messages) and the detached, contiguous launch tensors of the real-space sum."""
    if neighbor_list is None and neighbor_matrix is None:
        raise ValueError("Either neighbor_list or neighbor_matrix must be provided")
    if neighbor_list is not None and neighbor_ptr is None:
        raise ValueError("neighbor_ptr is required when using neighbor_list format")
    dev, dt = positions.device, positions.dtype
    C.dtype_code(dt)
    C.require_device(positions, charges, cell, neighbor_list, neighbor_ptr, neighbor_shifts, neighbor_matrix, neighbor_matrix_shifts, batch_idx)
    alpha_in = alpha if isinstance(alpha, torch.Tensor) else torch.tensor([float(alpha)], device=dev)
    alpha_in = alpha_in.to(device=dev, dtype=dt).reshape(-1)
    if alpha_in.numel() == 1 and cell.reshape(-1, 3, 3).shape[0] > 1:
        alpha_in = alpha_in.expand(cell.reshape(-1, 3, 3).shape[0])
    if neighbor_list is not None:
        idx, nptr, m = C.i32(neighbor_list[1]), C.i32(neighbor_ptr), 0
        sh, n_entries = neighbor_shifts, idx.shape[0]
    else:
        idx, nptr, m = C.i32(neighbor_matrix), None, neighbor_matrix.shape[1]
        sh, n_entries = neighbor_matrix_shifts, idx.numel()
    # a padded matrix that is provably the unmodified output of a full search of this package: its counts let the real-space kernel skip the
    # list-symmetry checksums (neighborlist/_engine.py::FullListRecord).  Not under torch.compile tracing (the record lives on the tensor object)
    counts = None
    if neighbor_list is None and neighbor_matrix_shifts is not None and _TRUST_FULL_LISTS and not C.tracing():
        from nvalchemiops.neighborlist import _engine as E

        counts = E.full_list_counts(neighbor_matrix, neighbor_matrix_shifts)
        if counts is not None and (idx is not neighbor_matrix or counts.shape[0] != positions.shape[0]):
            counts = None
    sh = torch.zeros((n_entries, 3), dtype=torch.int32, device=dev) if sh is None else C.i32(sh)
    if counts is not None and sh is not neighbor_matrix_shifts:
        counts = None
    return dict(counts=counts, pos=positions.detach().contiguous(), q=charges.detach().to(dt).contiguous(), cells=cell.detach().to(dt).reshape(-1, 3, 3).contiguous(),
                alpha=alpha_in.detach().contiguous(), alpha_in=alpha_in, bi=None if batch_idx is None else C.i32(batch_idx), idx=idx, sh=sh,
                nptr=nptr, m=m, n_entries=n_entries)


def _real_space_launch(p, mask_value: int, compute_forces: bool, compute_charge_gradients: bool):
    """One launch of the real-space family: (float64 energies, forces in the input dtype | None, float64 charge gradients | None)."""
    pos = p["pos"]
    n, dev, dt = pos.shape[0], pos.device, pos.dtype
    energies = torch.empty(n, dtype=torch.float64, device=dev)
    forces = torch.empty((n, 3), dtype=dt, device=dev) if compute_forces else None
    cgrads = torch.empty(n, dtype=torch.float64, device=dev) if compute_charge_gradients else None
    flags = (C.EW_FORCES if compute_forces else 0) | (C.EW_CHARGE_GRAD if compute_charge_gradients else 0)
    # scratch: list-symmetry checksums (zeroed by the library) + the {x,y,z,q} records the pair loop gathers
    nbytes = C.ewald_scratch_bytes(n, C.dtype_code(dt))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    counts = p.get("counts")
    # (a mask_value that is an atom's index would turn stored entries into padding: the list as the reference reads it is then not the search's list)
    if counts is not None and not (0 <= int(mask_value) < n):
        from nvalchemiops.neighborlist import _engine as E

        stride, phase = E.verify_args()
        rc = C.lib().mi_ewald_real_listed(C.ptr(pos), C.ptr(p["q"]), C.ptr(p["cells"]), C.ptr(p["alpha"]), C.ptr(p["bi"]), n, C.dtype_code(dt),
                                          C.ptr(p["idx"]), C.ptr(p["sh"]), C.ptr(p["nptr"]), int(p["m"]), int(mask_value), flags, C.ptr(energies),
                                          C.ptr(forces), C.ptr(cgrads), C.ptr(scratch), ctypes.c_size_t(nbytes), C.ptr(counts), int(stride),
                                          int(phase), C.stream_of(pos))
        C.check(rc, "mi_ewald_real_listed")
        return energies, forces, cgrads
    rc = C.lib().mi_ewald_real(C.ptr(pos), C.ptr(p["q"]), C.ptr(p["cells"]), C.ptr(p["alpha"]), C.ptr(p["bi"]), n, C.dtype_code(dt), C.ptr(p["idx"]),
                               C.ptr(p["sh"]), C.ptr(p["nptr"]), int(p["m"]), int(mask_value), flags, C.ptr(energies), C.ptr(forces), C.ptr(cgrads),
                               C.ptr(scratch), ctypes.c_size_t(nbytes), C.stream_of(pos))
    C.check(rc, "mi_ewald_real")
    return energies, forces, cgrads


@C.traceable
def ewald_real_space(positions: torch.Tensor, charges: torch.Tensor, cell: torch.Tensor, alpha: torch.Tensor,
                     neighbor_list: torch.Tensor | None = None, neighbor_ptr: torch.Tensor | None = None,
                     neighbor_shifts: torch.Tensor | None = None, neighbor_matrix: torch.Tensor | None = None,
                     neighbor_matrix_shifts: torch.Tensor | None = None, mask_value: int = -1, batch_idx: torch.Tensor | None = None,
                     compute_forces: bool = False, compute_charge_gradients: bool = False):
    """E_i = 1/2 sum_j q_i q_j erfc(alpha r_ij)/r_ij over the listed neighbours (per-atom energies, input dtype).

    Every stored entry (i, j, S) contributes to E_i, -f to F_i and +f to F_j, and to both charge gradients, exactly as in the
    reference (ewald_kernels.py:518-544, :864-873); pass a FULL (symmetric) list for physically meaningful totals -- that case runs
    without atomics.  Whether the list is symmetric is decided on the device by comparing two 64-bit checksums (entries as stored vs
    mirrored); a non-symmetric list passing as symmetric needs a 64-bit collision (~2^-64 per call) -- the one probabilistic step on
    this path.  Returns ``energies`` | ``(energies, forces)`` | ``(energies, charge_grads)`` | ``(energies, forces, charge_grads)``."""
    if neighbor_list is None and neighbor_matrix is None:
        raise ValueError("Either neighbor_list or neighbor_matrix must be provided")
    if neighbor_list is not None and neighbor_ptr is None:
        raise ValueError("neighbor_ptr is required when using neighbor_list format")
    n, dev, dt = positions.shape[0], positions.device, positions.dtype
    if n == 0:
        C.dtype_code(dt)
        out = (torch.zeros(0, dtype=dt, device=dev),)
        if compute_forces:
            out += (torch.zeros((0, 3), dtype=dt, device=dev),)
        if compute_charge_gradients:
            out += (torch.zeros(0, dtype=dt, device=dev),)
        return out if len(out) > 1 else out[0]
    alpha_t = alpha if isinstance(alpha, torch.Tensor) else torch.full((1,), float(alpha), dtype=dt, device=dev)
    if C.tracing() or (torch.is_grad_enabled() and any(t.requires_grad for t in (positions, charges, cell, alpha_t))):
        # the registered `alchemiops::_[batch_]ewald_real_space_*` op of this configuration (nvalchemiops/_eops.py): energies are
        # differentiable w.r.t. positions, charges, cell and alpha through the adjoint kernel; explicit forces / charge gradients are
        # outputs of the same launch (differentiating THEM raises NotImplementedError)
        from nvalchemiops import _eops

        batched = batch_idx is not None
        fmt = "list" if neighbor_list is not None else "matrix"
        op = _eops.real_space_op(batched, fmt, compute_forces, compute_charge_gradients)
        cells = cell.reshape(-1, 3, 3)
        al = alpha_t.to(dt).reshape(-1)
        if al.shape[0] == 1 and cells.shape[0] > 1:
            al = al.expand(cells.shape[0])
        head = (positions, charges.to(dt), cells.to(dt), al) + ((batch_idx,) if batched else ())
        if fmt == "list":
            n_entries = neighbor_list.shape[1]
            sh = neighbor_shifts if neighbor_shifts is not None else torch.zeros((n_entries, 3), dtype=torch.int32, device=dev)
            res = op(*head, neighbor_list, neighbor_ptr, sh)
        else:
            sh = neighbor_matrix_shifts if neighbor_matrix_shifts is not None else torch.zeros(tuple(neighbor_matrix.shape) + (3,), dtype=torch.int32, device=dev)
            res = op(*head, neighbor_matrix, sh, int(mask_value))
        res = res if isinstance(res, tuple) else (res,)
        out = (res[0],) + ((res[1],) if compute_forces else ()) + ((res[-1],) if compute_charge_gradients else ())
        return out if len(out) > 1 else out[0]
    p = _real_space_inputs(positions, charges, cell, alpha_t, neighbor_list, neighbor_ptr, neighbor_shifts, neighbor_matrix, neighbor_matrix_shifts,
                           batch_idx)
    if p["n_entries"] == 0:
        energies = torch.zeros(n, dtype=torch.float64, device=dev)
        forces = torch.zeros((n, 3), dtype=dt, device=dev) if compute_forces else None
        cgrads = torch.zeros(n, dtype=torch.float64, device=dev) if compute_charge_gradients else None
    else:
        energies, forces, cgrads = _real_space_launch(p, mask_value, compute_forces, compute_charge_gradients)
    out = (energies.to(dt),)  # ewald.py:577: energies are accumulated in float64 and returned in the input dtype
    if compute_forces:
        out += (forces,)
    if compute_charge_gradients:
        out += (cgrads.to(dt),)
    return out if len(out) > 1 else out[0]


def _prepare_alpha(alpha, num_systems: int, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    """float / 0-d / [B] alpha -> [B] tensor (ewald.py:190-235)."""
    if isinstance(alpha, (int, float)):
        return torch.full((num_systems,), float(alpha), dtype=dtype, device=device)
    if isinstance(alpha, torch.Tensor):
        if alpha.dim() == 0:
            return alpha.expand(num_systems).to(dtype=dtype, device=device)
        if alpha.shape[0] != num_systems:
            raise ValueError(f"alpha has {alpha.shape[0]} values but there are {num_systems} systems")
        return alpha.to(dtype=dtype, device=device)
    raise TypeError(f"alpha must be float or torch.Tensor, got {type(alpha)}")


def _prepare_cell(cell: torch.Tensor):
    if cell.dim() == 2:
        cell = cell.unsqueeze(0)
    return cell, cell.shape[0]


def _seg(x, batch_idx, n_sys):
    """Per-system sum of a per-atom float64 vector (wave-aggregated `mi_segment_sum`, not one same-address atomic per atom)."""
    out = torch.zeros(n_sys, dtype=x.dtype, device=x.device)
    xc = x.contiguous()
    if xc.shape[0]:
        C.check(C.lib().mi_segment_sum(C.ptr(xc), C.ptr(batch_idx), xc.shape[0], C.dtype_code(xc.dtype), C.ptr(out), C.stream_of(xc)), "mi_segment_sum")
    return out


def _structure_factors(pos, w, kv, cells, al, sptr, n_sys, n_k, max_atoms, want_charge=True):
    dev = pos.device
    sf = torch.empty((n_sys, n_k, 2), dtype=torch.float64, device=dev)
    tq = torch.empty(n_sys, dtype=torch.float64, device=dev) if want_charge else None
    rc = C.lib().mi_ewald_structure_factors(C.ptr(pos), C.ptr(w), C.ptr(kv), C.ptr(cells), C.ptr(al), C.ptr(sptr), pos.shape[0], n_sys, n_k,
                                            int(max_atoms), C.dtype_code(pos.dtype), C.ptr(sf), C.ptr(tq), C.stream_of(pos))
    C.check(rc, "mi_ewald_structure_factors")
    return sf, tq


def _recip_gather(pos, q, kv, al, bi, sf, tq, n_k, potential=False, kforce=False, energies=False, forces=False, cgrads=False):
    n, dev = pos.shape[0], pos.device
    f64 = dict(dtype=torch.float64, device=dev)
    out = dict(potential=torch.empty(n, **f64) if potential else None, kforce=torch.empty((n, 3), **f64) if kforce else None,
               energies=torch.empty(n, **f64) if energies else None,
               forces=torch.empty((n, 3), dtype=pos.dtype, device=dev) if forces else None,
               cgrads=torch.empty(n, **f64) if cgrads else None)
    rc = C.lib().mi_ewald_recip_gather(C.ptr(pos), C.ptr(q), C.ptr(kv), C.ptr(al), C.ptr(bi), C.ptr(sf), C.ptr(tq), n, n_k,
                                       C.dtype_code(pos.dtype), C.ptr(out["potential"]), C.ptr(out["kforce"]), C.ptr(out["energies"]),
                                       C.ptr(out["forces"]), C.ptr(out["cgrads"]), C.stream_of(pos))
    C.check(rc, "mi_ewald_recip_gather")
    return out


def _recip_inputs(positions, charges, cell, k_vectors, alpha, batch_idx):
    """Launch arguments of the explicit-k reciprocal sum: detached contiguous arrays in the positions dtype, [B,K,3] k-vectors, [B]
    alpha, and for a batch the CSR system pointer + the largest system (the structure-factor kernel's block shape)."""
    n, dev, dt = positions.shape[0], positions.device, positions.dtype
    cells, n_sys = _prepare_cell(cell)
    kv = k_vectors if k_vectors.dim() == 3 else k_vectors.unsqueeze(0)
    if kv.shape[0] != n_sys:
        kv = kv.expand(n_sys, -1, -1)
    p = dict(n_sys=n_sys, n_k=kv.shape[1], pos=positions.detach().contiguous(), q=charges.detach().to(dt).contiguous(),
             cells=cells.detach().to(dt).contiguous(), kv=kv.detach().to(dt).contiguous(),
             al=_prepare_alpha(alpha, n_sys, dt, dev).detach().contiguous(), bi=None, sptr=None, max_atoms=n)
    if batch_idx is not None and n > 0:
        p["bi"] = C.i32(batch_idx)
        counts = torch.bincount(batch_idx.long(), minlength=n_sys)
        sptr = torch.zeros(n_sys + 1, dtype=torch.int32, device=dev)
        sptr[1:] = torch.cumsum(counts, dim=0)
        p["sptr"], p["max_atoms"] = sptr, int(counts.max().item())
    return p


def _recip_forward(positions, charges, cell, k_vectors, alpha, batch_idx, forces: bool, cgrads: bool):
    """(energies [, forces] [, charge_grads]) in the positions dtype; no autograd graph (the `alchemiops::_[batch_]ewald_reciprocal_space_*`
    ops of nvalchemiops/_eops.py and the plain eager call both land here)."""
    n, dev, dt = positions.shape[0], positions.device, positions.dtype
    k3 = k_vectors if k_vectors.dim() == 3 else k_vectors.unsqueeze(0)
    # no k-vectors: every reference op except the single-system energy-only one returns zeros before its kernels (ewald.py:1507, :1655,
    # :1809, :1976, :2154); that one (:1365) still adds the self and background terms
    if n == 0 or (k3.shape[1] == 0 and (batch_idx is not None or forces or cgrads)):
        return (torch.zeros(n, dtype=dt, device=dev),) + ((torch.zeros((n, 3), dtype=dt, device=dev),) if forces else ()) + (
            (torch.zeros(n, dtype=dt, device=dev),) if cgrads else ())
    p = _recip_inputs(positions, charges, cell, k_vectors, alpha, batch_idx)
    sf, tq = _structure_factors(p["pos"], p["q"], p["kv"], p["cells"], p["al"], p["sptr"], p["n_sys"], p["n_k"], p["max_atoms"])
    res = _recip_gather(p["pos"], p["q"], p["kv"], p["al"], p["bi"], sf, tq, p["n_k"], energies=True, forces=forces, cgrads=cgrads)
    return (res["energies"].to(dt),) + ((res["forces"],) if forces else ()) + ((res["cgrads"].to(dt),) if cgrads else ())


def _recip_adjoint(p, g_e, need_atoms: bool, need_kv: bool, need_alpha: bool, need_vol: bool):
    """Adjoint of the per-atom reciprocal energies.  With L = sum_i g_i E_i and the g-weighted structure factors
    S^g = G sum_j g_j q_j exp(i k.r_j):
        dL/dr_m = -1/2 q_m (g_m kf_m[S] + kf_m[S^g]),
        dL/dq_m = 1/2 (g_m phi_m[S] + phi_m[S^g]) - 2 g_m alpha q_m/sqrt(pi) - pi/(2 alpha^2 V) (g_m Q + sum_i g_i q_i)
    -- two more passes of the forward kernels instead of a recorded tape (reference: autograd.py:525-665).
    Cell gradients (the reference's `test_cell_gradients`, test_ewald.py:2117) flow through `k_vectors` and the volume: with
    W_k = Re[conj(S^g) S]/G (the k-th term of 2 L) and G = 8 pi/V exp(-k^2/4 alpha^2)/k^2,
        dL/dk   = 1/2 W dlnG/dk + 1/(2G) d/dk Re[conj(A^g) A] G^2   (first moments sum_j w_j r_j exp(i k.r_j): six more
                  structure-factor passes with weights q r_c and g q r_c),
        dL/dalpha = 1/2 sum_k W k^2/(2 alpha^3) + sum_i g_i (-q_i^2/sqrt(pi) + pi q_i Q/(alpha^3 V)),
        dL/dV     = -1/(2V) sum_k W + sum_i g_i pi q_i Q/(2 alpha^2 V^2).
    Returns float64 (dL/dpositions, dL/dcharges, dL/dk_vectors [B,K,3], dL/dalpha [B], dL/dV [B]); entries not asked for are None."""
    pos, q, kv, cells, al, bi, sptr, max_atoms = (p[k] for k in ("pos", "q", "kv", "cells", "al", "bi", "sptr", "max_atoms"))
    n_sys, n_k = p["n_sys"], p["n_k"]
    batched = bi is not None
    sf, tq = _structure_factors(pos, q, kv, cells, al, sptr, n_sys, n_k, max_atoms)
    g = g_e.detach().to(torch.float64)
    q64, al64 = q.to(torch.float64), al.to(torch.float64)
    gq = (g * q64).to(pos.dtype).contiguous()
    sfg, tqg = _structure_factors(pos, gq, kv, cells, al, sptr, n_sys, n_k, max_atoms)
    sel = bi.long() if batched else torch.zeros(pos.shape[0], dtype=torch.long, device=pos.device)
    a_i = al64[sel]
    gpos = gch = gkv = gal = gvol = None
    if need_atoms:
        a = _recip_gather(pos, q, kv, al, bi, sf, None, n_k, potential=True, kforce=True)
        b = _recip_gather(pos, q, kv, al, bi, sfg, None, n_k, potential=True, kforce=True)
        gpos = (-0.5 * q64).unsqueeze(1) * (g.unsqueeze(1) * a["kforce"] + b["kforce"])
        gch = (0.5 * (g * a["potential"] + b["potential"]) - 2.0 * g * a_i * q64 / math.sqrt(math.pi)
               - math.pi / (2.0 * a_i * a_i) * (g * tq[sel] + tqg[sel]))
    if need_kv or need_alpha or need_vol:
        kv64 = kv.to(torch.float64)
        k2 = (kv64 * kv64).sum(-1)
        vol = torch.abs(torch.linalg.det(cells.to(torch.float64)))
        a2 = (al64 * al64).unsqueeze(1)
        ok = k2 >= 1e-10
        k2s = torch.where(ok, k2, torch.ones_like(k2))
        green = 8.0 * math.pi / vol.unsqueeze(1) * torch.exp(-k2s / (4.0 * a2)) / k2s
        ok = ok & (green > 1e-280)
        ginv = torch.where(ok, 1.0 / torch.where(ok, green, torch.ones_like(green)), torch.zeros_like(green))
        srq, siq, srg, sig = sf[..., 0], sf[..., 1], sfg[..., 0], sfg[..., 1]
        w = (srg * srq + sig * siq) * ginv
        if need_kv:
            gk = 0.5 * w.unsqueeze(-1) * (-kv64 * (0.5 / a2 + 2.0 / k2s).unsqueeze(-1))
            cross = []
            for c in range(3):
                rc_ = pos[:, c]
                m, _ = _structure_factors(pos, (q * rc_).contiguous(), kv, cells, al, sptr, n_sys, n_k, max_atoms, want_charge=False)
                mg, _ = _structure_factors(pos, (gq * rc_).contiguous(), kv, cells, al, sptr, n_sys, n_k, max_atoms, want_charge=False)
                cross.append((-mg[..., 1] * srq - srg * m[..., 1] + mg[..., 0] * siq + sig * m[..., 0]) * ginv)
            gkv = gk + 0.5 * torch.stack(cross, dim=-1)
        gi_q = g * q64
        seg = lambda x: _seg(x, bi, n_sys)  # noqa: E731
        if need_alpha:
            gal = 0.5 * (w * k2).sum(1) / (2.0 * al64**3) + seg(gi_q * (-q64 / math.sqrt(math.pi) + math.pi * tq[sel] / a_i**3))
        if need_vol:
            gvol = -0.5 * w.sum(1) / vol + seg(gi_q * math.pi * tq[sel] / (2.0 * a_i * a_i * vol[sel]))
    return gpos, gch, gkv, gal, gvol


def _recip_outputs_adjoint(p, g_f, g_c, need_alpha: bool, need_vol: bool, need_kv: bool = False):
    """Adjoint of the explicit FORCES and CHARGE GRADIENTS of the reciprocal sum, L = sum_i W_i . F_i + sum_i v_i cg_i (W = g_f, v = g_c).
    With c_ik / s_ik = cos / sin(k.r_i), Qc / Qs the charge sums, a_ik = q_i (W_i . k), As / Ac its sine / cosine sums, Vc / Vs those of v:
        L = sum_k G_k (Qc As - Qs Ac + Qc Vc + Qs Vs) - 2 alpha/sqrt(pi) sum v_i q_i - pi/alpha^2 (Q/V) sum v_i.
    Structure factors: S^q (weights q), S^v (weights v), S^A = sum_d k_d S[q W_d]  ->  S'' = (G As + G Vc, -G Ac + G Vs).  Then
        dL/dr_m = q_m KK_m[S^q; W_m] - q_m kf_m[S''] - v_m kf_m[S^q]                (KK: `mi_ewald_recip_gather_kk`)
        dL/dq_m = phi_m[S''] + W_m . kf_m[S^q] - 2 alpha v_m/sqrt(pi) - pi/alpha^2 sum v / V
        dL/dalpha = sum_k X_k k^2/(2 alpha^3) - 2/sqrt(pi) sum v_i q_i + 2 pi/alpha^3 (Q/V) sum v ,  X_k = Re[conj(S^q) S'']/G_k
        dL/dV     = -1/V sum_k X_k + pi/alpha^2 (Q/V^2) sum v.
    Five structure-factor passes and three gathers of the forward kernels' cost.
    k-vectors (`need_kv`; the reference lists them in `grad_arrays` of the force / charge-gradient ops, ewald.py:1481-1489, :1948, :2125 --
    the route by which a cell that generated the k-vectors receives the gradient of the forces): with L_k the k-th term of the sum above,
        dL/dk = L_k (-2 k)(1/(4 alpha^2) + 1/k^2)                                                              (through G_k)
              + G [ -S[q r] (As + Vc) + C[q r] (-Ac + Vs) + Qc (S[q W] + sum_e k_e C[q W_e r] - S[v r]) + Qs (-C[q W] + sum_e k_e S[q W_e r] + C[v r]) ]
    where C[w] / S[w] = sum_i w_i cos / sin(k.r_i) are structure factors with VECTOR weights (q r_d, v r_d, q W_e r_d: 15 more passes of
    `mi_ewald_structure_factors`; this is a second-order path, cf. the reference's test_cell_gradients).
    Returns float64 (dL/dpositions, dL/dcharges, dL/dalpha [B] | None, dL/dV [B] | None, dL/dk_vectors [B,K,3] | None)."""
    pos, q, kv, cells, al, bi, sptr, max_atoms = (p[k] for k in ("pos", "q", "kv", "cells", "al", "bi", "sptr", "max_atoms"))
    n_sys, n_k, n, dev = p["n_sys"], p["n_k"], pos.shape[0], pos.device
    f64 = dict(dtype=torch.float64, device=dev)
    W = torch.zeros((n, 3), **f64) if g_f is None else g_f.detach().to(torch.float64).contiguous()
    v = torch.zeros(n, **f64) if g_c is None else g_c.detach().to(torch.float64).contiguous()
    q64, al64, kv64 = q.to(torch.float64), al.to(torch.float64), kv.to(torch.float64)
    sel = bi.long() if bi is not None else torch.zeros(n, dtype=torch.long, device=dev)
    sf_of = lambda w, charge=False: _structure_factors(pos, w.to(pos.dtype).contiguous(), kv, cells, al, sptr, n_sys, n_k, max_atoms, want_charge=charge)  # noqa: E731
    sq, tq = sf_of(q64, True)                                       # tq = Q/V (0 when there is one k-vector: the reference's k_idx == 1 rule)
    s2 = sf_of(v)[0].clone()
    if g_f is not None:
        for d in range(3):
            sa = sf_of(q64 * W[:, d])[0]                             # (G Ac_d, G As_d)
            s2[..., 0] += kv64[..., d] * sa[..., 1]
            s2[..., 1] -= kv64[..., d] * sa[..., 0]
    a = _recip_gather(pos, q, kv, al, bi, sq, None, n_k, potential=False, kforce=True)
    b = _recip_gather(pos, q, kv, al, bi, s2.contiguous(), None, n_k, potential=True, kforce=True)
    a_i = al64[sel]
    vsum = _seg(v, bi, n_sys)
    vol = torch.abs(torch.linalg.det(cells.to(torch.float64)))
    gpos = -q64.unsqueeze(1) * b["kforce"] - v.unsqueeze(1) * a["kforce"]
    if g_f is not None:
        kk = torch.empty((n, 3), **f64)
        rc = C.lib().mi_ewald_recip_gather_kk(C.ptr(pos), C.ptr(kv), C.ptr(bi), C.ptr(sq), C.ptr(W), n, n_k, C.dtype_code(pos.dtype), C.ptr(kk),
                                              C.stream_of(pos))
        C.check(rc, "mi_ewald_recip_gather_kk")
        gpos = gpos + q64.unsqueeze(1) * kk
    # pi/alpha^2 (Q/V) sum v: d/dq_m of Q/V is 1/V wherever the forward accumulated the charge at all (tq != 0 or Q == 0 with K > 1)
    charged = 1.0 if n_k > 1 else 0.0
    gq = b["potential"] + (W * a["kforce"]).sum(-1) - 2.0 * a_i * v / math.sqrt(math.pi) - charged * math.pi / (a_i * a_i) * (vsum / vol)[sel]
    gal = gvol = gkv = None
    if need_alpha or need_vol or need_kv:
        k2 = (kv64 * kv64).sum(-1)
        a2 = (al64 * al64).unsqueeze(1)
        ok = k2 >= 1e-10
        k2s = torch.where(ok, k2, torch.ones_like(k2))
        green = 8.0 * math.pi / vol.unsqueeze(1) * torch.exp(-k2s / (4.0 * a2)) / k2s
        ok = ok & (green > 1e-280)
        ginv = torch.where(ok, 1.0 / torch.where(ok, green, torch.ones_like(green)), torch.zeros_like(green))
        x = (sq[..., 0] * s2[..., 0] + sq[..., 1] * s2[..., 1]) * ginv
        vq = _seg(v * q64, bi, n_sys)
        if need_alpha:
            gal = (x * k2).sum(1) / (2.0 * al64**3) - 2.0 / math.sqrt(math.pi) * vq + 2.0 * math.pi / al64**3 * tq * vsum
        if need_vol:
            gvol = -x.sum(1) / vol + math.pi / (al64 * al64) * tq / vol * vsum
        if need_kv:
            pos64 = pos.to(torch.float64)
            gi = ginv.unsqueeze(-1)                                    # 1/G (0 where the Green function is masked)
            # through G_k: x_k = L_k
            gkv = (x * (-2.0) * (1.0 / (4.0 * a2) + torch.where(ok, 1.0 / k2s, torch.zeros_like(k2s)))).unsqueeze(-1) * kv64
            cq, sq_ = sq[..., 0:1] * gi, sq[..., 1:2] * gi             # Qc, Qs
            for d in range(3):
                sqr = sf_of(q64 * pos64[:, d])[0]                      # G (C[q r_d], S[q r_d])
                term = (-sqr[..., 1] * s2[..., 0] + sqr[..., 0] * s2[..., 1]) * ginv
                inner_c = torch.zeros_like(k2)                         # G (dAs_d + dVc_d), G (-dAc_d + dVs_d)
                inner_s = torch.zeros_like(k2)
                if g_f is not None:
                    sw = sf_of(q64 * W[:, d])[0]
                    inner_c = inner_c + sw[..., 1]
                    inner_s = inner_s - sw[..., 0]
                    for e in range(3):
                        swr = sf_of(q64 * W[:, e] * pos64[:, d])[0]
                        inner_c = inner_c + kv64[..., e] * swr[..., 0]
                        inner_s = inner_s + kv64[..., e] * swr[..., 1]
                if g_c is not None:
                    svr = sf_of(v * pos64[:, d])[0]
                    inner_c = inner_c - svr[..., 1]
                    inner_s = inner_s + svr[..., 0]
                gkv[..., d] += term + cq[..., 0] * inner_c + sq_[..., 0] * inner_s
    return gpos, gq, gal, gvol, gkv


@C.traceable
def ewald_reciprocal_space(positions: torch.Tensor, charges: torch.Tensor, cell: torch.Tensor, k_vectors: torch.Tensor, alpha: torch.Tensor,
                           batch_idx: torch.Tensor | None = None, compute_forces: bool = False, compute_charge_gradients: bool = False):
    """Reciprocal-space Ewald over an explicit half-space k-vector set (ewald.py:2631-2795):
    E_i = q_i/2 sum_k Re[S(k) exp(-i k.r_i)] - alpha q_i^2/sqrt(pi) - pi q_i Q/(2 alpha^2 V), per-atom, input dtype.

    Returns ``energies`` | ``(energies, forces)`` | ``(energies, charge_grads)`` | ``(energies, forces, charge_grads)``.
    `k_vectors` is [K,3] (single system) or [B,K,3] (batch); `alpha` a [B] tensor (see `ewald_summation` for float input).
    Energies are differentiable w.r.t. positions, charges, k_vectors, alpha and cell (the latter through |det cell| here and
    through `k_vectors` when those were generated from a cell that requires grad, as the reference's tests do).  When something
    requires grad, or under torch.compile, the call goes through the `alchemiops::_[batch_]ewald_reciprocal_space_*` ops."""
    dt, dev = positions.dtype, positions.device
    if positions.shape[0] > 0:
        C.require_device(positions, charges, cell, k_vectors, batch_idx)
    cells, n_sys = _prepare_cell(cell)
    al_in = _prepare_alpha(alpha, n_sys, dt, dev)
    if C.tracing() or (torch.is_grad_enabled() and any(t.requires_grad for t in (positions, charges, k_vectors, cells, al_in))):
        from nvalchemiops import _eops

        op = _eops.reciprocal_space_op(batch_idx is not None, compute_forces or compute_charge_gradients, compute_charge_gradients)
        args = (positions, charges.to(dt), cells.to(dt), k_vectors.to(dt), al_in) + ((batch_idx,) if batch_idx is not None else ())
        out = op(*args)
        out = out if isinstance(out, tuple) else (out,)
        if compute_charge_gradients and not compute_forces:  # no (energies, charge_grads) op: the three-output op minus its forces (ewald.py:2756)
            out = (out[0], out[2])
    else:
        out = _recip_forward(positions, charges, cells, k_vectors, al_in, batch_idx, compute_forces, compute_charge_gradients)
    return out if len(out) > 1 else out[0]


@C.eager
def ewald_summation(positions: torch.Tensor, charges: torch.Tensor, cell: torch.Tensor, alpha=None, k_vectors: torch.Tensor | None = None,
                    k_cutoff: float | None = None, batch_idx: torch.Tensor | None = None, neighbor_list: torch.Tensor | None = None,
                    neighbor_ptr: torch.Tensor | None = None, neighbor_shifts: torch.Tensor | None = None,
                    neighbor_matrix: torch.Tensor | None = None, neighbor_matrix_shifts: torch.Tensor | None = None,
                    mask_value: int | None = None, compute_forces: bool = False, accuracy: float = 1e-6):
    """Real + reciprocal Ewald sum with automatic parameters (ewald.py:2798-3050): per-atom energies | (energies, forces)."""
    from nvalchemiops.interactions.electrostatics.k_vectors import generate_k_vectors_ewald_summation
    from nvalchemiops.interactions.electrostatics.parameters import estimate_ewald_parameters

    cells, n_sys = _prepare_cell(cell)
    if alpha is None or (k_cutoff is None and k_vectors is None):
        params = estimate_ewald_parameters(positions, cells, batch_idx, accuracy)
        if alpha is None:
            alpha = params.alpha
        if k_cutoff is None:
            k_cutoff = params.reciprocal_space_cutoff
    alpha_t = _prepare_alpha(alpha, n_sys, positions.dtype, positions.device)
    if k_vectors is None:
        k_vectors = generate_k_vectors_ewald_summation(cells, k_cutoff)
    if mask_value is None:
        mask_value = positions.shape[0]
    rs = ewald_real_space(positions, charges, cells, alpha_t, neighbor_list=neighbor_list, neighbor_ptr=neighbor_ptr,
                          neighbor_shifts=neighbor_shifts, neighbor_matrix=neighbor_matrix, neighbor_matrix_shifts=neighbor_matrix_shifts,
                          mask_value=mask_value, batch_idx=batch_idx, compute_forces=compute_forces)
    rec = ewald_reciprocal_space(positions, charges, cells, k_vectors, alpha_t, batch_idx=batch_idx, compute_forces=compute_forces)
    if compute_forces:
        return rs[0] + rec[0], rs[1] + rec[1]
    return rs + rec


__all__ = ["ewald_real_space", "ewald_reciprocal_space", "ewald_summation"]

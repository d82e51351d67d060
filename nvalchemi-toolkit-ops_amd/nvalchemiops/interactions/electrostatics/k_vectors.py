"""`generate_k_vectors_pme` -- drop-in for interactions/electrostatics/k_vectors.py:167-298.

k[(B,) nx, ny, nz/2+1, 3] = Miller(fftfreq, fftfreq, rfftfreq) . (2 pi cell^-1)^T and k^2 clamped to >= 1e-12.  Only needed
when a caller wants the arrays (e.g. to cache them); the PME path of this build evaluates k in registers inside the
convolution kernel (csrc/pme.hip) unless the caller passes these arrays in.
"""
from __future__ import annotations

import math

import torch


def generate_k_vectors_pme(cell: torch.Tensor, mesh_dimensions: tuple[int, int, int], reciprocal_cell: torch.Tensor | None = None):
    cells = cell if cell.dim() == 3 else cell.unsqueeze(0)
    if reciprocal_cell is None:
        reciprocal_cell = (2.0 * math.pi) * torch.linalg.inv(cells)
    nx, ny, nz = mesh_dimensions
    kw = dict(device=cell.device, dtype=cell.dtype)
    hx = torch.fft.fftfreq(nx, d=1.0, **kw) * nx
    hy = torch.fft.fftfreq(ny, d=1.0, **kw) * ny
    hz = torch.fft.rfftfreq(nz, d=1.0, **kw) * nz
    miller = torch.stack(torch.meshgrid(hx, hy, hz, indexing="ij"), dim=-1)
    k_vectors = torch.einsum("ijkd,bcd->bijkc", miller, reciprocal_cell).squeeze(0)
    k_sq = (k_vectors * k_vectors).sum(dim=-1)
    return k_vectors, torch.where(k_sq > 1e-12, k_sq, torch.tensor(1e-12, device=cell.device))


def _generate_miller_indices(cell: torch.Tensor, k_cutoff):
    """Per-axis Miller bound ceil(k_cutoff |a_d| / 2 pi), maximum over the batch (k_vectors.py:19-40)."""
    lengths = torch.norm(cell, dim=-1).max(dim=0).values / (2 * math.pi)
    return torch.ceil(k_cutoff * lengths).long()


def generate_k_vectors_ewald_summation(cell: torch.Tensor, k_cutoff) -> torch.Tensor:
    """Half-space k-vector set for the explicit Ewald sum (k_vectors.py:43-164): every Miller triple inside the
    per-axis bound with h>0 | (h==0,k>0) | (h==k==0,l>0), times 2 pi cell^-T.  [K,3] or [B,K,3] (same Miller set for all systems)."""
    if cell.ndim == 2:
        cell = cell.unsqueeze(0)
    kw = dict(device=cell.device, dtype=cell.dtype)
    if isinstance(k_cutoff, torch.Tensor):
        k_cutoff = k_cutoff.max() if k_cutoff.numel() > 1 else k_cutoff.reshape(())
    nh, nk, nl = (2 * _generate_miller_indices(cell, k_cutoff) + 1).tolist()
    h = torch.fft.fftfreq(nh, **kw) * nh
    k = torch.fft.fftfreq(nk, **kw) * nk
    m = torch.fft.fftfreq(nl, **kw) * nl
    miller = torch.stack([g.flatten() for g in torch.meshgrid(h, k, m, indexing="ij")], dim=1)
    hh, kk, mm = miller[:, 0], miller[:, 1], miller[:, 2]
    miller = miller[(hh > 0) | ((hh == 0) & (kk > 0)) | ((hh == 0) & (kk == 0) & (mm > 0))]
    reciprocal = (2.0 * math.pi) * torch.linalg.inv(cell.transpose(1, 2))
    return (miller.to(reciprocal.dtype) @ reciprocal).squeeze(0)


__all__ = ["generate_k_vectors_pme", "generate_k_vectors_ewald_summation"]

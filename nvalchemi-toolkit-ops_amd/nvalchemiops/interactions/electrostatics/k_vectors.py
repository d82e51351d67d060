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


__all__ = ["generate_k_vectors_pme"]

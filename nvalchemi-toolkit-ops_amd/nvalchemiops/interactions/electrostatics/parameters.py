"""Ewald / PME parameter estimation -- drop-in for interactions/electrostatics/parameters.py:68-437.

Kolafa-Perram balance: eta = (V^2/N)^(1/6)/sqrt(2 pi), alpha = 1/(sqrt(2) eta), r_cut = sqrt(-2 ln eps) eta,
k_cut = sqrt(-2 ln eps)/eta;  PME mesh n_d = 2 alpha L_d / (3 eps^(1/5)) rounded up to a power of two and maximised
over the batch.  Pure host-side torch arithmetic on (B,)-sized tensors, as in the reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass
class EwaldParameters:
    alpha: torch.Tensor
    real_space_cutoff: torch.Tensor
    reciprocal_space_cutoff: torch.Tensor


@dataclass
class PMEParameters:
    alpha: torch.Tensor
    mesh_dimensions: tuple[int, int, int]
    mesh_spacing: torch.Tensor
    real_space_cutoff: torch.Tensor


def _as_batch(cell: torch.Tensor) -> torch.Tensor:
    return cell.unsqueeze(0) if cell.ndim == 2 else cell


def _atoms_per_system(positions: torch.Tensor, num_systems: int, batch_idx: torch.Tensor | None) -> torch.Tensor:
    if batch_idx is None:
        return torch.tensor([positions.shape[0]], dtype=torch.int32, device=positions.device)
    return torch.zeros(num_systems, dtype=torch.int32, device=batch_idx.device).scatter_add_(0, batch_idx, torch.ones_like(batch_idx))


def _next_pow2(x: torch.Tensor) -> torch.Tensor:
    return torch.pow(2, torch.ceil(torch.log2(x))).to(torch.int32)


def estimate_ewald_parameters(positions: torch.Tensor, cell: torch.Tensor, batch_idx: torch.Tensor | None = None,
                              accuracy: float = 1e-6) -> EwaldParameters:
    cell = _as_batch(cell)
    volume = torch.abs(torch.linalg.det(cell)).squeeze(-1)
    n = _atoms_per_system(positions, cell.shape[0], batch_idx).to(positions.dtype)
    eta = (volume**2 / n) ** (1.0 / 6.0) / math.sqrt(2.0 * math.pi)
    spread = math.sqrt(-2.0 * math.log(accuracy))
    return EwaldParameters(alpha=1.0 / (math.sqrt(2.0) * eta), real_space_cutoff=spread * eta, reciprocal_space_cutoff=spread / eta)


def estimate_pme_mesh_dimensions(cell: torch.Tensor, alpha: torch.Tensor, accuracy: float = 1e-6) -> tuple[int, int, int]:
    lengths = torch.norm(_as_batch(cell), dim=2)
    wanted = 2 * alpha[:, None] * lengths / (3.0 * accuracy**0.2)
    dims = _next_pow2(torch.max(wanted, dim=0).values)
    return tuple(int(v) for v in dims.tolist())


def estimate_pme_parameters(positions: torch.Tensor, cell: torch.Tensor, batch_idx: torch.Tensor | None = None,
                            accuracy: float = 1e-6) -> PMEParameters:
    cell = _as_batch(cell)
    ew = estimate_ewald_parameters(positions, cell, batch_idx, accuracy)
    dims = estimate_pme_mesh_dimensions(cell, ew.alpha, accuracy)
    lengths = torch.norm(cell, dim=2)
    spacing = lengths / torch.tensor(dims, dtype=lengths.dtype, device=lengths.device)
    return PMEParameters(alpha=ew.alpha, mesh_dimensions=dims, mesh_spacing=spacing, real_space_cutoff=ew.real_space_cutoff)


def mesh_spacing_to_dimensions(cell: torch.Tensor, mesh_spacing: float | torch.Tensor) -> tuple[int, int, int]:
    cell = _as_batch(cell)
    lengths = torch.norm(cell, dim=2)
    if isinstance(mesh_spacing, float):
        counts = torch.ceil(lengths / mesh_spacing)
    elif mesh_spacing.ndim == 1:
        if mesh_spacing.shape[0] != cell.shape[0]:
            raise ValueError(f"mesh_spacing shape {mesh_spacing.shape} incompatible with cell batch size {cell.shape[0]}")
        counts = torch.ceil(lengths / mesh_spacing[:, None])
    else:
        if mesh_spacing.shape != lengths.shape:
            raise ValueError(f"mesh_spacing shape {mesh_spacing.shape} incompatible with cell_lengths shape {lengths.shape}")
        counts = torch.ceil(lengths / mesh_spacing)
    dims = torch.max(_next_pow2(counts), dim=0).values
    return tuple(int(v) for v in dims.tolist())


__all__ = ["EwaldParameters", "PMEParameters", "estimate_ewald_parameters", "estimate_pme_parameters", "estimate_pme_mesh_dimensions",
           "mesh_spacing_to_dimensions"]

"""B-spline charge spreading / gathering -- drop-in for the scalar-channel part of nvalchemiops/spline.py
(`spline_spread` :2581, `spline_gather` :2640, `spline_gather_vec3` :2684; ops `alchemiops::_[batch_]spline_*` :1500-2107).

HIP kernels in csrc/pme.hip behind `mi_spline_spread / _gather / _gather_vec3`.  Orders 1-4 reproduce the reference's
piecewise polynomials; orders 5 and 6 are true cardinal B-splines (the reference evaluates them as zero: SURVEY F2).
"""
from __future__ import annotations

import torch

from nvalchemiops import _capi as C


def _prep(positions: torch.Tensor, cell: torch.Tensor, batch_idx, cell_inv_t):
    pos = positions.detach().contiguous()
    c = cell.detach()
    c = (c if c.dim() == 3 else c.unsqueeze(0)).to(pos.dtype)
    if cell_inv_t is None:
        cell_inv_t = torch.linalg.inv(c).transpose(-1, -2)
    cit = cell_inv_t.detach().to(pos.dtype).reshape(-1, 3, 3).contiguous()
    bi = None if batch_idx is None else C.i32(batch_idx)
    return pos, c, cit, bi


def spline_spread(positions: torch.Tensor, values: torch.Tensor, cell: torch.Tensor, mesh_dims: tuple[int, int, int], spline_order: int = 4,
                  batch_idx: torch.Tensor | None = None, cell_inv_t: torch.Tensor | None = None) -> torch.Tensor:
    """mesh[(B,) nx, ny, nz] += values_i * M_n(x) M_n(y) M_n(z) over each atom's order^3 stencil (periodic wrap)."""
    C.require_device(positions, values, cell)
    pos, c, cit, bi = _prep(positions, cell, batch_idx, cell_inv_t)
    nx, ny, nz = (int(v) for v in mesh_dims)
    nsys = c.shape[0] if bi is not None else 1
    mesh = torch.zeros((nsys, nx, ny, nz), dtype=pos.dtype, device=pos.device)
    vals = values.detach().to(pos.dtype).contiguous()
    rc = C.lib().mi_spline_spread(C.ptr(pos), C.ptr(vals), C.ptr(bi), C.ptr(cit), pos.shape[0], nsys, nx, ny, nz, int(spline_order),
                                  int(bi is not None), C.dtype_code(pos.dtype), C.ptr(mesh), C.stream_of(pos))
    C.check(rc, "mi_spline_spread")
    return mesh if bi is not None else mesh[0]


def spline_gather(positions: torch.Tensor, mesh: torch.Tensor, cell: torch.Tensor, spline_order: int = 4,
                  batch_idx: torch.Tensor | None = None, cell_inv_t: torch.Tensor | None = None) -> torch.Tensor:
    """out_i = sum over the stencil of mesh[g] * w  (weights <= 1e-8 are skipped, spline.py:608)."""
    C.require_device(positions, mesh, cell)
    pos, c, cit, bi = _prep(positions, cell, batch_idx, cell_inv_t)
    m = mesh.detach().to(pos.dtype).contiguous()
    nx, ny, nz = m.shape[-3:]
    out = torch.empty(pos.shape[0], dtype=pos.dtype, device=pos.device)
    rc = C.lib().mi_spline_gather(C.ptr(pos), C.ptr(m), C.ptr(bi), C.ptr(cit), pos.shape[0], cit.shape[0], nx, ny, nz, int(spline_order),
                                  C.dtype_code(pos.dtype), C.ptr(out), C.stream_of(pos))
    C.check(rc, "mi_spline_gather")
    return out


def spline_gather_vec3(positions: torch.Tensor, charges: torch.Tensor, mesh: torch.Tensor, cell: torch.Tensor, spline_order: int = 4,
                       batch_idx: torch.Tensor | None = None, cell_inv_t: torch.Tensor | None = None) -> torch.Tensor:
    """out_i[3] = sum over the stencil of q_i * mesh[g, :] * w for a mesh of shape [(B,) nx, ny, nz, 3]."""
    C.require_device(positions, charges, mesh, cell)
    pos, c, cit, bi = _prep(positions, cell, batch_idx, cell_inv_t)
    m = mesh.detach().to(pos.dtype).contiguous()
    nx, ny, nz = m.shape[-4:-1]
    q = charges.detach().to(pos.dtype).contiguous()
    out = torch.empty((pos.shape[0], 3), dtype=pos.dtype, device=pos.device)
    rc = C.lib().mi_spline_gather_vec3(C.ptr(pos), C.ptr(q), C.ptr(m), C.ptr(bi), C.ptr(cit), pos.shape[0], cit.shape[0], nx, ny, nz,
                                       int(spline_order), C.dtype_code(pos.dtype), C.ptr(out), C.stream_of(pos))
    C.check(rc, "mi_spline_gather_vec3")
    return out


__all__ = ["spline_spread", "spline_gather", "spline_gather_vec3"]

"""B-spline charge spreading / gathering -- drop-in for the scalar-channel part of nvalchemiops/spline.py
(`spline_spread` :2581, `spline_gather` :2640, `spline_gather_vec3` :2684; ops `alchemiops::_[batch_]spline_*` :1500-2107).

HIP kernels in csrc/pme.hip behind `mi_spline_spread / _gather / _gather_vec3`.  Orders 1-4 reproduce the reference's
piecewise polynomials; orders 5 and 6 are true cardinal B-splines (the reference evaluates them as zero: SURVEY F2) unless
`reference_spline_orders()` / `set_reference_spline_orders(True)` asks for the reference's evaluation.
"""
from __future__ import annotations

import os

import torch

from nvalchemiops import _capi as C


def set_reference_spline_orders(enabled: bool) -> bool:
    """PROCESS-WIDE default of the one deliberate numerical deviation of this build; for a scoped
    change use the context manager `reference_spline_orders` below, which is local to the calling thread / task.

    False (default): spline orders 5 and 6 are true cardinal B-splines and the PME structure factor uses exponent = order.
    True: they are evaluated exactly as the reference evaluates them -- its weight function has cases for orders 1-4 only and returns 0
    otherwise (spline.py:150-193), and its structure-factor exponent is min(order, 4) (pme_kernels.py:213-225) -- so an order-5 PME
    call returns the reference's numbers (reciprocal potential identically zero, self / background corrections only).  Orders 1-4 are
    unaffected.  Also settable at import time with NVALCHEMIOPS_REFERENCE_SPLINE_ORDERS=1.  Returns the previous default.  Read at launch
    time (inside the custom ops as well), so it also applies to graphs compiled earlier.  Meant to be set once at start-up: flipping it
    while other threads launch is a data race on their results, which is what the context manager avoids."""
    prev = C._REFERENCE_SPLINE_ORDERS
    C._REFERENCE_SPLINE_ORDERS = bool(enabled)
    return prev


class reference_spline_orders:
    """``with reference_spline_orders(): particle_mesh_ewald(..., spline_order=5)`` -- the reference's evaluation of orders 5 / 6 for the
    calls made inside the block BY THIS THREAD / TASK (a `contextvars.ContextVar`: nothing process-wide changes, another thread's launches
    keep their own setting, and leaving the block -- normally or by an exception -- restores exactly what this context had before).
    ``reference_spline_orders(False)`` scopes the true B-splines inside a process whose default is the reference's evaluation."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled)
        self._token = None

    def __enter__(self):
        self._token = C._REFERENCE_SPLINE_ORDERS_CTX.set(self.enabled)
        return self

    def __exit__(self, *exc):
        C._REFERENCE_SPLINE_ORDERS_CTX.reset(self._token)
        return False


def _prep(positions: torch.Tensor, cell: torch.Tensor, batch_idx, cell_inv_t):
    pos = positions.detach().contiguous()
    c = cell.detach()
    c = (c if c.dim() == 3 else c.unsqueeze(0)).to(pos.dtype)
    if cell_inv_t is None:
        cell_inv_t = torch.linalg.inv(c).transpose(-1, -2)
    cit = cell_inv_t.detach().to(pos.dtype).reshape(-1, 3, 3).contiguous()
    bi = None if batch_idx is None else C.i32(batch_idx)
    if bi is not None and c.shape[0] == 1 and bi.numel():
        # one cell for the whole batch (spline.py:2256, :2775: `cell.unsqueeze(0).expand(num_systems, ...)`)
        nsys = int(bi.max().item()) + 1
        c, cit = c.expand(nsys, 3, 3), cit.expand(nsys, 3, 3).contiguous()
    return pos, c, cit, bi


# ---- raw launchers (detached, contiguous tensors) ------------------------------------------------------------------
# "auto": the library's measured policy (mi_spline_spread_prefers_tiles); "tile": the tile pipeline wherever the mesh allows it; "atomic":
# never.  The override exists for A/B runs and for the parity tests, which drive the tile kernels with small inputs.
_SPREAD_PATH = os.environ.get("NVALCHEMIOPS_SPREAD_PATH", "auto")


def _launch_spread(pos, vals, cit, bi, nsys, dims, order, batched, want_order=False, out=None):
    """mesh[nsys,nx,ny,nz] = spread of `vals` at `pos`.  The library runs tile-owned (no global atomics) when every mesh dimension
    has a divisor in [max(order - 1, 2), 8] and falls back to atomic adds into the zeroed mesh otherwise; the scratch buffer covers the former.
    want_order: also return the scratch buffer of a tile-owned run (None after the atomic kernel): atoms grouped by mesh tile, stencil starts
    and fractional offsets, which `mi_pme_gather_finish` of the same step takes as `spread_workspace` for its tile-staged gather."""
    import ctypes

    nx, ny, nz = (int(v) for v in dims)
    n = pos.shape[0]
    # the tile-owned kernel writes every mesh point exactly once: no zero-fill pass then.  Small systems take the atomic kernel: one
    # zero-fill + one launch instead of the five launches of the tile pipeline (launch latency is all there is to pay below ~12k atoms)
    if _SPREAD_PATH == "auto":
        tiled = n > 0 and bool(C.lib().mi_spline_spread_prefers_tiles(n, nsys, nx, ny, nz, C.spline_order_arg(order)))
    else:
        tiled = _SPREAD_PATH == "tile" and n > 0 and bool(C.lib().mi_spline_spread_is_tiled(nsys, nx, ny, nz, C.spline_order_arg(order)))
    if out is not None:  # a caller's contiguous [nsys, nx, ny, nz] block (one of several meshes that share a batched transform)
        mesh = out
        if not tiled:
            mesh.zero_()
    else:
        mesh = (torch.empty if tiled else torch.zeros)((nsys, nx, ny, nz), dtype=pos.dtype, device=pos.device)
    ws_bytes = int(C.lib().mi_spline_spread_workspace_bytes_for(n, nsys, nx, ny, nz, C.spline_order_arg(order), C.dtype_code(pos.dtype))) if tiled else 0
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=pos.device)
    rc = C.lib().mi_spline_spread(C.ptr(pos), C.ptr(vals), C.ptr(bi), C.ptr(cit), n, nsys, nx, ny, nz, C.spline_order_arg(order), int(batched),
                                  C.dtype_code(pos.dtype), C.ptr(mesh), C.ptr(ws), ctypes.c_size_t(ws_bytes), C.stream_of(pos))
    C.check(rc, "mi_spline_spread")
    if not want_order:
        return mesh
    return mesh, (ws if tiled else None)


def _launch_gather(pos, mesh, cit, bi, order, grad=False):
    nx, ny, nz = mesh.shape[-3:]
    out = torch.empty((pos.shape[0], 3) if grad else (pos.shape[0],), dtype=pos.dtype, device=pos.device)
    fn = C.lib().mi_spline_gather_grad if grad else C.lib().mi_spline_gather
    rc = fn(C.ptr(pos), C.ptr(mesh), C.ptr(bi), C.ptr(cit), pos.shape[0], cit.shape[0], nx, ny, nz, C.spline_order_arg(order), C.dtype_code(pos.dtype),
            C.ptr(out), C.stream_of(pos))
    C.check(rc, "mi_spline_gather_grad" if grad else "mi_spline_gather")
    return out


def _wants_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def _op_inputs(positions, cell, batch_idx, cell_inv_t):
    """Arguments of the `alchemiops::_[batch_]spline_*` ops (nvalchemiops/_eops.py): [B,3,3] cell, a DIFFERENTIABLE cell_inv_t (so
    cell gradients flow through torch.linalg.inv, as in the reference where `cell_inv_t` is the tracked array) and the system count."""
    from nvalchemiops import _eops  # noqa: F401  (registers the ops)

    dt = positions.dtype
    c = (cell if cell.dim() == 3 else cell.unsqueeze(0)).to(dt)
    cit = torch.linalg.inv(c).transpose(-1, -2) if cell_inv_t is None else cell_inv_t.to(dt).reshape(-1, 3, 3)
    nsys = 1
    if batch_idx is not None:
        # one cell for the whole batch (spline.py:2256, :2775): the system count then comes from the indices
        nsys = c.shape[0] if c.shape[0] > 1 else int(batch_idx.max().item()) + 1
    return c, cit, nsys


@C.traceable
def spline_spread(positions: torch.Tensor, values: torch.Tensor, cell: torch.Tensor, mesh_dims: tuple[int, int, int], spline_order: int = 4,
                  batch_idx: torch.Tensor | None = None, cell_inv_t: torch.Tensor | None = None) -> torch.Tensor:
    """mesh[(B,) nx, ny, nz] += values_i * M_n(x) M_n(y) M_n(z) over each atom's order^3 stencil (periodic wrap).
    Differentiable w.r.t. positions, values and cell (hand-written adjoint kernels)."""
    spline_order = C.resolve_spline_order(spline_order)  # read the reference-orders switch once, here (backward passes reuse the value)
    C.require_device(positions, values, cell)
    if C.tracing() or _wants_grad(positions, values, cell, cell_inv_t):
        c, cit, nsys = _op_inputs(positions, cell, batch_idx, cell_inv_t)
        nx, ny, nz = (int(v) for v in mesh_dims)
        if batch_idx is None:
            return torch.ops.alchemiops._spline_spread(positions, values.to(positions.dtype), c[0], nx, ny, nz, int(spline_order), cit)
        return torch.ops.alchemiops._batch_spline_spread(positions, values.to(positions.dtype), batch_idx, c, nsys, nx, ny, nz, int(spline_order), cit)
    pos, c, cit, bi = _prep(positions, cell, batch_idx, cell_inv_t)
    nx, ny, nz = (int(v) for v in mesh_dims)
    nsys = c.shape[0] if bi is not None else 1
    vals = values.detach().to(pos.dtype).contiguous()
    mesh = _launch_spread(pos, vals, cit, bi, nsys, (nx, ny, nz), int(spline_order), bi is not None)
    return mesh if bi is not None else mesh[0]


@C.traceable
def spline_gather(positions: torch.Tensor, mesh: torch.Tensor, cell: torch.Tensor, spline_order: int = 4,
                  batch_idx: torch.Tensor | None = None, cell_inv_t: torch.Tensor | None = None) -> torch.Tensor:
    """out_i = sum over the stencil of mesh[g] * w  (weights <= 1e-8 are skipped, spline.py:608).
    Differentiable w.r.t. positions, mesh and cell."""
    spline_order = C.resolve_spline_order(spline_order)  # read the reference-orders switch once, here (backward passes reuse the value)
    C.require_device(positions, mesh, cell)
    if C.tracing() or _wants_grad(positions, mesh, cell, cell_inv_t):
        c, cit, _ = _op_inputs(positions, cell, batch_idx, cell_inv_t)
        if batch_idx is None:
            return torch.ops.alchemiops._spline_gather(positions, mesh.to(positions.dtype), c[0], int(spline_order), cit)
        return torch.ops.alchemiops._batch_spline_gather(positions, mesh.to(positions.dtype), batch_idx, c, int(spline_order), cit)
    pos, c, cit, bi = _prep(positions, cell, batch_idx, cell_inv_t)
    m = mesh.detach().to(pos.dtype).contiguous()
    nx, ny, nz = m.shape[-3:]
    out = torch.empty(pos.shape[0], dtype=pos.dtype, device=pos.device)
    rc = C.lib().mi_spline_gather(C.ptr(pos), C.ptr(m), C.ptr(bi), C.ptr(cit), pos.shape[0], cit.shape[0], nx, ny, nz, C.spline_order_arg(spline_order),
                                  C.dtype_code(pos.dtype), C.ptr(out), C.stream_of(pos))
    C.check(rc, "mi_spline_gather")
    return out


@C.traceable
def spline_gather_vec3(positions: torch.Tensor, charges: torch.Tensor, mesh: torch.Tensor, cell: torch.Tensor, spline_order: int = 4,
                       batch_idx: torch.Tensor | None = None, cell_inv_t: torch.Tensor | None = None) -> torch.Tensor:
    """out_i[3] = sum over the stencil of q_i * mesh[g, :] * w for a mesh of shape [(B,) nx, ny, nz, 3].
    Differentiable w.r.t. positions, charges, mesh and cell (adjoint in nvalchemiops/_eops.py)."""
    spline_order = C.resolve_spline_order(spline_order)  # read the reference-orders switch once, here (backward passes reuse the value)
    C.require_device(positions, charges, mesh, cell)
    if C.tracing() or _wants_grad(positions, charges, mesh, cell, cell_inv_t):
        c, cit, _ = _op_inputs(positions, cell, batch_idx, cell_inv_t)
        if batch_idx is None:
            return torch.ops.alchemiops._spline_gather_vec3(positions, charges.to(positions.dtype), mesh.to(positions.dtype), c[0], int(spline_order), cit)
        return torch.ops.alchemiops._batch_spline_gather_vec3(positions, charges.to(positions.dtype), mesh.to(positions.dtype), batch_idx, c,
                                                              int(spline_order), cit)
    pos, c, cit, bi = _prep(positions, cell, batch_idx, cell_inv_t)
    m = mesh.detach().to(pos.dtype).contiguous()
    nx, ny, nz = m.shape[-4:-1]
    q = charges.detach().to(pos.dtype).contiguous()
    out = torch.empty((pos.shape[0], 3), dtype=pos.dtype, device=pos.device)
    rc = C.lib().mi_spline_gather_vec3(C.ptr(pos), C.ptr(q), C.ptr(m), C.ptr(bi), C.ptr(cit), pos.shape[0], cit.shape[0], nx, ny, nz,
                                       C.spline_order_arg(spline_order), C.dtype_code(pos.dtype), C.ptr(out), C.stream_of(pos))
    C.check(rc, "mi_spline_gather_vec3")
    return out


@C.traceable
def spline_gather_gradient(positions: torch.Tensor, charges: torch.Tensor, mesh: torch.Tensor, cell: torch.Tensor, spline_order: int = 4,
                           batch_idx: torch.Tensor | None = None, cell_inv_t: torch.Tensor | None = None) -> torch.Tensor:
    """F_i = -q_i sum_g mesh[g] grad_r w(r_i, g): the fractional-coordinate gradient (scaled by the mesh dimensions) mapped to Cartesian
    with cell_inv_t^T (spline.py:2733-2785; kernels :678-755).  Differentiable w.r.t. positions, charges, mesh and cell: the adjoint uses
    the second derivatives of the spline weights (nvalchemiops/_eops.py `_gather_gradient_backward`)."""
    spline_order = C.resolve_spline_order(spline_order)  # read the reference-orders switch once, here (backward passes reuse the value)
    C.require_device(positions, charges, mesh, cell)
    if C.tracing() or _wants_grad(positions, charges, mesh, cell, cell_inv_t):
        c, cit, _ = _op_inputs(positions, cell, batch_idx, cell_inv_t)
        if batch_idx is None:
            return torch.ops.alchemiops._spline_gather_gradient(positions, charges.to(positions.dtype), mesh.to(positions.dtype), c[0], int(spline_order), cit)
        return torch.ops.alchemiops._batch_spline_gather_gradient(positions, charges.to(positions.dtype), mesh.to(positions.dtype), batch_idx, c,
                                                                  int(spline_order), cit)
    pos, c, cit, bi = _prep(positions, cell, batch_idx, cell_inv_t)
    gfrac = _launch_gather(pos, mesh.detach().to(pos.dtype).contiguous(), cit, bi, int(spline_order), grad=True)
    cit_i = cit[bi.long()] if bi is not None else cit[0].expand(pos.shape[0], 3, 3)
    return -charges.detach().to(pos.dtype).unsqueeze(-1) * torch.einsum("na,nab->nb", gfrac, cit_i)


@C.traceable
def spline_spread_channels(positions: torch.Tensor, values: torch.Tensor, cell: torch.Tensor, mesh_dims: tuple[int, int, int],
                           spline_order: int = 4, batch_idx: torch.Tensor | None = None) -> torch.Tensor:
    """values[N, C] -> mesh[(B,) C, nx, ny, nz] (spline.py:2788-2860).  One tile-owned scalar spread per channel -- the channels
    share nothing but the per-atom weights, and the multipole path that uses many channels is outside this build's hot path."""
    chans = [spline_spread(positions, values[:, ch], cell, mesh_dims, spline_order, batch_idx) for ch in range(values.shape[1])]
    if not chans:
        lead = () if batch_idx is None else ((cell.shape[0] if cell.dim() == 3 else int(batch_idx.max().item()) + 1),)
        return torch.zeros(lead + (0,) + tuple(int(v) for v in mesh_dims), dtype=positions.dtype, device=positions.device)
    return torch.stack(chans, dim=0 if batch_idx is None else 1)


@C.traceable
def spline_gather_channels(positions: torch.Tensor, mesh: torch.Tensor, cell: torch.Tensor, spline_order: int = 4,
                           batch_idx: torch.Tensor | None = None) -> torch.Tensor:
    """mesh[(B,) C, nx, ny, nz] -> values[N, C] (spline.py:2863-2910); one scalar gather per channel."""
    nch = mesh.shape[0] if batch_idx is None else mesh.shape[1]
    cols = [spline_gather(positions, mesh[ch] if batch_idx is None else mesh[:, ch], cell, spline_order, batch_idx) for ch in range(nch)]
    if not cols:
        return torch.zeros((positions.shape[0], 0), dtype=positions.dtype, device=positions.device)
    return torch.stack(cols, dim=1)


def _bspline_modulus_sq(n: int, order: int, device) -> torch.Tensor:
    """|b(k)|^2 of the order-n cardinal B-spline on an n-point grid, b(k) = sum_j M_order(j+1) exp(2 pi i j k/n), k = fftfreq order;
    float64 (spline.py:2913-2980, Essmann et al. 1995 eq. 4.7)."""
    # The reference's coefficient table for "order n" holds the cardinal B-spline of order n+1 at the integers 1..n
    # ([1/6, 4/6, 1/6] for n = 3, ..., spline.py:2999-3022); beyond n = 6 it continues with its own two-term recursion (:3024-3035).
    base = min(order, 6)
    m = torch.zeros(base + 2, dtype=torch.float64)
    m[0] = 1.0  # order 1: indicator of [0, 1) at the integers 0..base+1
    u = torch.arange(base + 2, dtype=torch.float64)
    for o in range(2, base + 2):  # Cox-de Boor: M_o(u) = [u M_{o-1}(u) + (o - u) M_{o-1}(u - 1)] / (o - 1)
        m = (u * m + (o - u) * torch.cat([torch.zeros(1, dtype=torch.float64), m[:-1]])) / (o - 1)
    coeff = m[1:base + 1]
    for o in range(7, order + 1):
        uu = torch.arange(1, o + 1, dtype=torch.float64)
        cur = torch.cat([coeff, torch.zeros(1, dtype=torch.float64)])
        prev = torch.cat([torch.zeros(1, dtype=torch.float64), coeff])
        coeff = (uu * cur + (o - uu) * prev) / (o - 1)
    coeff = coeff.to(device)
    k = torch.fft.fftfreq(n, device=device) * n
    w = 2.0 * torch.pi * k.to(torch.float32).to(torch.float64) / n
    j = torch.arange(order, dtype=torch.float64, device=device)
    phase = w.unsqueeze(-1) * j
    b2 = (coeff * torch.cos(phase)).sum(-1) ** 2 + (coeff * torch.sin(phase)).sum(-1) ** 2
    return torch.where(k != 0, b2, torch.ones_like(b2))


def compute_bspline_deconvolution_1d(n: int, spline_order: int = 4, device=None) -> torch.Tensor:
    """1 / max(|b(k)|^2, 1e-15) along one mesh axis (spline.py:3117-3148)."""
    device = torch.device("cpu") if device is None else device
    return 1.0 / torch.clamp(_bspline_modulus_sq(int(n), int(spline_order), device), min=1e-15)


def compute_bspline_deconvolution(mesh_dims: tuple[int, int, int], spline_order: int = 4, device=None) -> torch.Tensor:
    """deconv[nx, ny, nz] = 1 / max(|b(kx)|^2 |b(ky)|^2 |b(kz)|^2, 1e-15): multiply the FFT of a spread mesh by it to undo the
    B-spline smoothing (spline.py:3038-3114).  Pure torch on `device` (default CPU, as the reference)."""
    device = torch.device("cpu") if device is None else device
    nx, ny, nz = (int(v) for v in mesh_dims)
    bx, by, bz = (_bspline_modulus_sq(n, int(spline_order), device) for n in (nx, ny, nz))
    return 1.0 / torch.clamp(bx.view(nx, 1, 1) * by.view(1, ny, 1) * bz.view(1, 1, nz), min=1e-15)


__all__ = ["spline_spread", "spline_gather", "spline_gather_vec3", "spline_gather_gradient", "spline_spread_channels", "spline_gather_channels",
           "compute_bspline_deconvolution", "compute_bspline_deconvolution_1d", "set_reference_spline_orders", "reference_spline_orders"]

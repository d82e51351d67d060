"""CPU oracle -- TEST INFRASTRUCTURE ONLY (not part of the product).

numpy/ctypes front-end of ``oracle/nvalchemi_oracle.cpp`` (a single-threaded C++ restatement of the
reference's Warp kernels) plus the few torch-level steps of the reference path restated in numpy
(FFT composition of PME, k-vector grid, matrix->COO conversion).  Also holds two *independent*
checkers that do not share code with the restatement: an O(N^2 * images) brute-force neighbour
enumerator and an explicit structure-factor Ewald sum.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  Nothing under ``nvalchemi-toolkit-ops_amd/`` does.

Reference files followed (all under /root/reference/nvalchemiops, v0.2.0):
  neighborlist/cell_list.py:35-556, batch_cell_list.py:36-569, neighbor_utils.py:26-441, naive.py:37-182
  interactions/dispersion/dftd3.py:341-1615, 1911-2122, 2306-2465
  interactions/electrostatics/ewald_kernels.py:150-1495, pme_kernels.py:93-657, pme.py:1166-1479,
  k_vectors.py:167-298, spline.py:127-959, math/math.py:41-93
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    """Compile the oracle shared library with g++ (make) and return its path."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return os.path.join(_HERE, "libnvalchemi_oracle.so")


def _load(name: str) -> ctypes.CDLL:
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "nvalchemi_oracle.cpp")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build()
    L = ctypes.CDLL(path)
    L.orc_erfc.restype = ctypes.c_double
    L.orc_erfc.argtypes = [ctypes.c_int, ctypes.c_double]
    L.orc_cell_list.restype = ctypes.c_int
    return L


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = _load("libnvalchemi_oracle.so")
    return _LIB


_FFT_WORKERS = None  # None: numpy's single-threaded pocketfft; int: scipy.fft with that many workers (all-core baseline leg only)


class openmp:
    """Context manager: run the restatement on `threads` host cores (the -fopenmp build of the same source, scipy.fft workers for
    the FFTs).  For the all-core leg of bench.py's cpu_baseline ONLY -- parity tests use the serial library, whose accumulation
    order is the reference's."""

    def __init__(self, threads: int | None = None):
        self.threads = threads or os.cpu_count() or 1

    def __enter__(self):
        global _LIB, _FFT_WORKERS
        self._saved = (_LIB, _FFT_WORKERS)
        ext = lib().orc_set_extended(0)
        lib().orc_set_extended(ext)
        wide = lib().orc_set_d3_wide_sums(0)
        lib().orc_set_d3_wide_sums(wide)
        _LIB = _load("libnvalchemi_oracle_omp.so")
        _LIB.orc_set_extended(ext)
        _LIB.orc_set_d3_wide_sums(wide)
        self.threads = int(_LIB.orc_set_threads(int(self.threads)))
        _FFT_WORKERS = self.threads
        return self

    def __exit__(self, *exc):
        global _LIB, _FFT_WORKERS
        _LIB, _FFT_WORKERS = self._saved
        return False


def _rfftn(x, axes):
    if _FFT_WORKERS:
        import scipy.fft

        return scipy.fft.rfftn(x, axes=axes, workers=_FFT_WORKERS)
    return np.fft.rfftn(x, axes=axes)


def _irfftn(x, s, axes):
    if _FFT_WORKERS:
        import scipy.fft

        return scipy.fft.irfftn(x, s=s, axes=axes, workers=_FFT_WORKERS)
    return np.fft.irfftn(x, s=s, axes=axes)


class extended_splines:
    """Context manager: "beyond reference" mode of the spline / structure-factor restatement (true order-5/6 cardinal B-splines by
    the closed-form truncated-power sum, structure-factor exponent = order).  The reference itself evaluates orders 5/6 as zero
    (SURVEY F2/F3), so this mode is NOT pinned by the reference; it is pinned by the NaCl Madelung constant and the independent
    explicit Ewald sum (tests/test_oracle_golden.py) and is what the product's order-5/6 path is compared with."""

    def __enter__(self):
        self._old = lib().orc_set_extended(1)
        return self

    def __exit__(self, *exc):
        lib().orc_set_extended(self._old)
        return False


class d3_wide_sums:
    """Context manager: D3 restatement with every fp32 accumulation of the reference (CN, dE/dCN, per-system energy / virial) carried
    in double and rounded once -- the reference's arithmetic without its summation-order noise (error budget)."""

    def __enter__(self):
        self._old = lib().orc_set_d3_wide_sums(1)
        return self

    def __exit__(self, *exc):
        lib().orc_set_d3_wide_sums(self._old)
        return False


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _dt(a) -> int:
    if a.dtype == np.float32:
        return 0
    if a.dtype == np.float64:
        return 1
    raise ValueError(f"unsupported dtype {a.dtype}")


def _c(a, dtype=None):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


# --------------------------------------------------------------------------------------
# neighbour lists
# --------------------------------------------------------------------------------------
def estimate_max_neighbors(cutoff: float, atomic_density: float = 0.35, safety_factor: float = 5.0) -> int:
    """neighbor_utils.py:296-340."""
    if cutoff <= 0:
        return 0
    expected = max(1, safety_factor * atomic_density * (4.0 / 3.0) * math.pi * cutoff**3)
    return int(math.ceil(expected / 16)) * 16


def cell_list(positions, cutoff, cell, pbc, batch_idx=None, max_neighbors=None, fill_value=None, half_fill=False,
              max_nbins=1000, return_grid=False):
    """Reference cell-list pipeline (single system when batch_idx is None, batch kernels otherwise)."""
    pos = _c(positions)
    n = pos.shape[0]
    cell = _c(cell, pos.dtype).reshape(-1, 3, 3)
    nsys = cell.shape[0]
    pbc = _c(np.asarray(pbc).reshape(-1, 3), np.uint8)
    if pbc.shape[0] != nsys:
        pbc = np.ascontiguousarray(np.broadcast_to(pbc, (nsys, 3)))
    m = estimate_max_neighbors(cutoff) if max_neighbors is None else int(max_neighbors)
    fv = n if fill_value is None else int(fill_value)
    nm = np.empty((n, m), np.int32)
    sh = np.empty((n, m, 3), np.int32)
    num = np.empty((n,), np.int32)
    bi = _c(batch_idx, np.int32)
    cpd = np.zeros((nsys, 3), np.int32)
    rad = np.zeros((nsys, 3), np.int32)
    if n > 0 and cutoff > 0:
        lib().orc_cell_list(_dt(pos), _p(pos), n, _p(cell), _p(pbc), _p(bi), nsys, ctypes.c_double(cutoff),
                            int(max_nbins), m, fv, int(half_fill), _p(nm), _p(sh), _p(num), _p(cpd), _p(rad))
    else:
        nm[:] = fv
        sh[:] = 0
        num[:] = 0
    if return_grid:
        return nm, num, sh, cpd, rad
    return nm, num, sh


def build_cell_cache(positions, cutoff, cell, pbc, max_total_cells, batch_idx=None):
    """What build_cell_list / batch_build_cell_list leave in the caller's cache tensors (cell_list.py:725-889, batch_cell_list.py:739-912):
    (cells_per_dimension, atom_periodic_shifts, atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices, cell_atom_list),
    atoms of a cell listed in ascending index."""
    pos = _c(positions)
    n = pos.shape[0]
    cell = _c(cell, pos.dtype).reshape(-1, 3, 3)
    nsys = cell.shape[0]
    pbc = _c(np.broadcast_to(np.asarray(pbc).reshape(-1, 3), (nsys, 3)), np.uint8)
    bi = _c(batch_idx, np.int32)
    c = int(max_total_cells)
    cpd, shift, amap = np.zeros((nsys, 3), np.int32), np.zeros((n, 3), np.int32), np.zeros((n, 3), np.int32)
    counts, starts, atoms = np.zeros(c, np.int32), np.zeros(c, np.int32), np.zeros(n, np.int32)
    lib().orc_cell_cache(_dt(pos), _p(pos), n, _p(cell), _p(pbc), _p(bi), nsys, ctypes.c_double(cutoff), c, _p(cpd), _p(shift), _p(amap),
                         _p(counts), _p(starts), _p(atoms))
    return (cpd if batch_idx is not None else cpd[0]), shift, amap, counts, starts, atoms


def cells_changed(positions, cell, atom_to_cell_mapping, cells_per_dimension, pbc) -> bool:
    """rebuild_detection.py:37-110."""
    pos = _c(positions)
    return bool(lib().orc_cells_changed(_dt(pos), _p(pos), pos.shape[0], _p(_c(cell, pos.dtype).reshape(3, 3)), _p(_c(atom_to_cell_mapping, np.int32)),
                                        _p(_c(np.asarray(cells_per_dimension).reshape(3), np.int32)), _p(_c(np.asarray(pbc).reshape(3), np.uint8))))


def moved_beyond_skin(reference_positions, current_positions, threshold) -> bool:
    """rebuild_detection.py:113-170."""
    ref = _c(reference_positions)
    cur = _c(current_positions, ref.dtype)
    return bool(lib().orc_moved_beyond_skin(_dt(ref), _p(ref), _p(cur), ref.shape[0], ctypes.c_double(threshold)))


def naive(positions, cutoff, cell=None, pbc=None, max_neighbors=None, fill_value=None, half_fill=False, image_range_cutoff=None):
    """`image_range_cutoff`: the cutoff the periodic-image table is built for when it is not `cutoff` itself -- list 1 of the reference's
    dual-cutoff kernels uses the table of cutoff2 (naive_dual_cutoff.py:835, :215-226)."""
    pos = _c(positions)
    n = pos.shape[0]
    m = estimate_max_neighbors(cutoff) if max_neighbors is None else int(max_neighbors)
    fv = n if fill_value is None else int(fill_value)
    nm = np.empty((n, m), np.int32)
    num = np.empty((n,), np.int32)
    sh = np.zeros((n, m, 3), np.int32) if cell is not None else None
    c = None if cell is None else _c(cell, pos.dtype).reshape(3, 3)
    pb = None if pbc is None else _c(np.asarray(pbc).reshape(3), np.uint8)
    if image_range_cutoff is None:
        lib().orc_naive(_dt(pos), _p(pos), n, _p(c), _p(pb), ctypes.c_double(cutoff), m, fv, int(half_fill), _p(nm), _p(sh), _p(num))
    else:
        lib().orc_naive_range(_dt(pos), _p(pos), n, _p(c), _p(pb), ctypes.c_double(cutoff), ctypes.c_double(image_range_cutoff), m, fv,
                              int(half_fill), _p(nm), _p(sh), _p(num))
    return (nm, num, sh) if cell is not None else (nm, num)


class NeighborOverflow(Exception):
    pass


def matrix_to_coo(nm, num, shifts=None, fill_value=-1):
    """neighbor_utils.py:362-441: boolean-mask conversion, row-major."""
    if num.size and num.max() > nm.shape[1]:
        raise NeighborOverflow(f"{num.max()} > {nm.shape[1]}")
    mask = nm != fill_value
    i_idx = np.nonzero(mask)[0].astype(np.int32)
    j_idx = nm[mask].astype(np.int32)
    ptr = np.zeros(num.shape[0] + 1, np.int32)
    np.cumsum(num, out=ptr[1:])
    out = (np.stack([i_idx, j_idx], 0), ptr)
    if shifts is not None:
        out = out + (shifts[mask],)
    return out


def canonical_pairs(nm, num, shifts, fill_value=None):
    """Sorted (i, j, Sx, Sy, Sz) rows of a neighbour matrix -- the order-free comparison key."""
    n, m = nm.shape
    cols = np.arange(m)[None, :]
    mask = cols < np.minimum(num, m)[:, None]
    i = np.nonzero(mask)[0]
    rows = np.column_stack([i, nm[mask], shifts[mask]]).astype(np.int64)
    return rows[np.lexsort(rows.T[::-1])]


def brute_force_pairs(positions, cutoff, cell, pbc):
    """Independent checker: every directed (i, j, S) with |r_j + S.cell - r_i| < cutoff, S over all images.

    Same floating-point expression order as the cell-list distance test so boundary pairs agree.
    """
    pos = np.asarray(positions)
    t = pos.dtype.type
    cell = np.asarray(cell, pos.dtype).reshape(3, 3)
    pbc = np.asarray(pbc, bool).reshape(3)
    inv = np.linalg.inv(cell.astype(np.float64))
    # atoms may sit outside the cell: extend the image range by the fractional spread
    frac = pos.astype(np.float64) @ inv
    spread = np.ceil(frac.max(0) - frac.min(0)).astype(int) if len(pos) else np.zeros(3, int)
    rng = []
    for d in range(3):
        face = 1.0 / np.linalg.norm(inv[:, d])
        rng.append(int(math.ceil(cutoff / face)) + int(spread[d]) + 1 if pbc[d] else 0)
    rc2 = t(cutoff) * t(cutoff)
    rows = []
    n = len(pos)
    for sx in range(-rng[0], rng[0] + 1):
        for sy in range(-rng[1], rng[1] + 1):
            for sz in range(-rng[2], rng[2] + 1):
                s = np.array([sx, sy, sz], pos.dtype)
                cart = np.empty(3, pos.dtype)
                for c in range(3):
                    r = cell[0, c] * s[0]
                    r = r + cell[1, c] * s[1]
                    r = r + cell[2, c] * s[2]
                    cart[c] = r
                dr = (pos[None, :, :] - pos[:, None, :]) + cart[None, None, :]
                d2 = dr[..., 0] * dr[..., 0] + dr[..., 1] * dr[..., 1] + dr[..., 2] * dr[..., 2]
                hit = d2 < rc2
                if sx == 0 and sy == 0 and sz == 0:
                    hit[np.arange(n), np.arange(n)] = False
                ii, jj = np.nonzero(hit)
                if len(ii):
                    rows.append(np.column_stack([ii, jj, np.full(len(ii), sx), np.full(len(ii), sy), np.full(len(ii), sz)]))
    if not rows:
        return np.zeros((0, 5), np.int64)
    rows = np.concatenate(rows).astype(np.int64)
    return rows[np.lexsort(rows.T[::-1])]


# --------------------------------------------------------------------------------------
# DFT-D3
# --------------------------------------------------------------------------------------
def d3_test_tables(z_max: int = 17, seed: int | None = None):
    """The analytic D3 test tables (an INPUT generator, not a checker): lives in tests/systems.py; re-exported here for the tests
    that reach it through the oracle module."""
    from tests.systems import d3_test_tables as gen

    return gen(z_max, seed)


def dftd3(positions, numbers, params, a1, a2, s8, k1=16.0, k3=-4.0, s6=1.0, s5_on=1e10, s5_off=1e10, fill_value=None,
          neighbor_matrix=None, neighbor_matrix_shifts=None, idx_j=None, neighbor_ptr=None, unit_shifts=None, cell=None,
          batch_idx=None, compute_virial=False, num_systems=None):
    pos = _c(positions)
    n = pos.shape[0]
    numbers = _c(numbers, np.int32)
    bi = _c(batch_idx, np.int32)
    if num_systems is None:
        num_systems = 1 if bi is None else (np.asarray(cell).reshape(-1, 3, 3).shape[0] if cell is not None else int(bi.max()) + 1)
    rcov = _c(params["rcov"], np.float32)
    r4r2 = _c(params["r4r2"], np.float32)
    c6ab = _c(params["c6ab"], np.float32)
    cnref = _c(params["cn_ref"], np.float32)
    energy = np.zeros(num_systems, np.float32)
    forces = np.zeros((n, 3), np.float32)
    cn = np.zeros(n, np.float32)
    virial = np.zeros((num_systems, 3, 3), np.float32)
    if neighbor_matrix is not None:
        j = _c(neighbor_matrix, np.int32)
        m = j.shape[1]
        ptr = None
        sh = _c(neighbor_matrix_shifts, np.int32)
        fv = n if fill_value is None else int(fill_value)
    else:
        j = _c(idx_j, np.int32)
        m = 0
        ptr = _c(neighbor_ptr, np.int32)
        sh = _c(unit_shifts, np.int32)
        fv = 0
    c = None if (cell is None or sh is None) else _c(cell, pos.dtype).reshape(-1, 3, 3)
    if c is None:
        sh = None
    if n:
        lib().orc_dftd3(_dt(pos), _p(pos), _p(numbers), n, _p(j), _p(sh), _p(ptr), m, fv, _p(c), _p(bi), num_systems,
                        _p(rcov), _p(r4r2), _p(c6ab), _p(cnref), rcov.shape[0], *(ctypes.c_double(x) for x in
                        (a1, a2, s6, s8, k1, k3, s5_on, s5_off)), int(compute_virial), _p(energy), _p(forces), _p(cn), _p(virial))
    return (energy, forces, cn, virial) if compute_virial else (energy, forces, cn)


# --------------------------------------------------------------------------------------
# electrostatics
# --------------------------------------------------------------------------------------
def erfc_as(x, dtype=np.float64):
    return lib().orc_erfc(0 if dtype == np.float32 else 1, float(x))


def ewald_real_space(positions, charges, cell, alpha, neighbor_matrix=None, neighbor_matrix_shifts=None, idx_j=None,
                     neighbor_ptr=None, neighbor_shifts=None, mask_value=-1, batch_idx=None, compute_forces=False,
                     compute_charge_gradients=False):
    pos = _c(positions)
    n = pos.shape[0]
    q = _c(charges, pos.dtype)
    c = _c(cell, pos.dtype).reshape(-1, 3, 3)
    al = _c(np.broadcast_to(np.asarray(alpha, pos.dtype).reshape(-1), (c.shape[0],)), pos.dtype)
    bi = _c(batch_idx, np.int32)
    if neighbor_matrix is not None:
        j, sh, ptr, m = _c(neighbor_matrix, np.int32), _c(neighbor_matrix_shifts, np.int32), None, neighbor_matrix.shape[1]
    else:
        j, sh, ptr, m = _c(idx_j, np.int32), _c(neighbor_shifts, np.int32), _c(neighbor_ptr, np.int32), 0
    e = np.zeros(n, np.float64)
    f = np.zeros((n, 3), pos.dtype)
    cg = np.zeros(n, np.float64)
    if n:
        lib().orc_ewald_real(_dt(pos), _p(pos), _p(q), _p(c), _p(al), _p(bi), n, _p(j), _p(sh), _p(ptr), m, int(mask_value),
                             int(compute_forces), int(compute_charge_gradients), _p(e), _p(f), _p(cg))
    out = (e.astype(pos.dtype),)
    if compute_forces:
        out += (f,)
    if compute_charge_gradients:
        out += (cg.astype(pos.dtype),)
    return out if len(out) > 1 else out[0]


def _cell_inv_t(cell):
    return np.ascontiguousarray(np.swapaxes(np.linalg.inv(cell), -1, -2))


def _spline(mode, positions, values, cell, dims, order, batch_idx, mesh):
    pos = _c(positions)
    n = pos.shape[0]
    cell = _c(cell, pos.dtype).reshape(-1, 3, 3)
    nsys = cell.shape[0]
    cit = _c(_cell_inv_t(cell), pos.dtype)
    bi = _c(batch_idx, np.int32)
    d = np.asarray(dims, np.int32)
    vals = _c(values, pos.dtype) if values is not None else None
    if mode == 0:
        mesh = np.zeros((nsys,) + tuple(dims), pos.dtype)
        out = None
    else:
        mesh = _c(mesh, pos.dtype)
        out = np.zeros((n,) if mode == 1 else (n, 3), pos.dtype)
    if n:
        lib().orc_spline(_dt(pos), mode, _p(pos), _p(vals), _p(bi), _p(cit), n, nsys, _p(d), int(order), _p(mesh), _p(out))
    if mode == 0:
        return mesh if batch_idx is not None else mesh[0]
    return out


def spline_spread(positions, values, cell, mesh_dims, spline_order=4, batch_idx=None):
    return _spline(0, positions, values, cell, mesh_dims, spline_order, batch_idx, None)


def spline_gather(positions, mesh, cell, spline_order=4, batch_idx=None):
    dims = mesh.shape[-3:]
    return _spline(1, positions, None, cell, dims, spline_order, batch_idx, mesh)


def spline_gather_vec3(positions, charges, mesh, cell, spline_order=4, batch_idx=None):
    dims = mesh.shape[-4:-1]
    return _spline(2, positions, charges, cell, dims, spline_order, batch_idx, mesh)


def generate_k_vectors_pme(cell, mesh_dimensions):
    """k_vectors.py:167-298 in numpy."""
    cell = np.asarray(cell).reshape(-1, 3, 3)
    dt = cell.dtype
    recip = (2.0 * math.pi) * np.linalg.inv(cell)
    nx, ny, nz = mesh_dimensions
    kx = (np.fft.fftfreq(nx, d=1.0) * nx).astype(dt)
    ky = (np.fft.fftfreq(ny, d=1.0) * ny).astype(dt)
    kz = (np.fft.rfftfreq(nz, d=1.0) * nz).astype(dt)
    grid = np.stack(np.meshgrid(kx, ky, kz, indexing="ij"), -1)
    kvec = np.einsum("ijkd,bcd->bijkc", grid, recip.astype(dt))
    if kvec.shape[0] == 1:
        kvec = kvec[0]
    k2 = np.sum(kvec**2, -1)
    k2 = np.where(k2 > 1e-12, k2, np.float32(1e-12).astype(dt))
    return kvec, k2


def pme_green_structure_factor(k_squared, mesh_dimensions, alpha, cell, spline_order=4, batched=False):
    k2 = _c(k_squared)
    dt = k2.dtype
    cell = np.asarray(cell, dt).reshape(-1, 3, 3)
    nsys = cell.shape[0]
    vol = _c(np.abs(np.linalg.det(cell)), dt)
    al = _c(np.broadcast_to(np.asarray(alpha, dt).reshape(-1), (nsys,)), dt)
    nx, ny, nz = mesh_dimensions
    g = np.zeros(k2.shape, dt)
    sf2 = np.zeros((nx, ny, nz // 2 + 1), dt)
    lib().orc_green_sf(_dt(k2), _p(k2), _p(al), _p(vol), nsys, nx, ny, nz, int(spline_order), _p(g), _p(sf2))
    return g, sf2


def pme_energy_corrections(raw, charges, cell, alpha, batch_idx=None, with_charge_grad=False):
    raw = _c(raw)
    dt = raw.dtype
    q = _c(charges, dt)
    cell = np.asarray(cell, dt).reshape(-1, 3, 3)
    nsys = cell.shape[0]
    vol = _c(np.abs(np.linalg.det(cell)), dt)
    al = _c(np.broadcast_to(np.asarray(alpha, dt).reshape(-1), (nsys,)), dt)
    bi = _c(batch_idx, np.int32)
    if bi is None:
        qtot = np.array([q.sum()], dt)
    else:
        qtot = np.zeros(nsys, dt)
        np.add.at(qtot, bi, q)
    e = np.zeros_like(raw)
    cg = np.zeros_like(raw) if with_charge_grad else None
    lib().orc_corrections(_dt(raw), _p(raw), _p(q), _p(bi), _p(vol), _p(al), _p(qtot), raw.shape[0], _p(e), _p(cg))
    return (e, cg) if with_charge_grad else e


def pme_reciprocal_space(positions, charges, cell, alpha, mesh_dimensions, spline_order=4, batch_idx=None,
                         compute_forces=False, compute_charge_gradients=False):
    """pme.py:1338-1479 with numpy FFTs (rfftn unscaled forward, irfftn norm='forward' == unscaled inverse)."""
    pos = _c(positions)
    dt = pos.dtype
    cell = _c(cell, dt).reshape(-1, 3, 3)
    batched = batch_idx is not None
    axes = (-3, -2, -1)
    mesh = spline_spread(pos, charges, cell, mesh_dimensions, spline_order, batch_idx)
    mesh_fft = _rfftn(mesh, axes)
    kvec, k2 = generate_k_vectors_pme(cell, mesh_dimensions)
    if batched and kvec.ndim == 4:
        kvec, k2 = kvec[None], k2[None]
    g, sf2 = pme_green_structure_factor(k2, mesh_dimensions, alpha, cell, spline_order)
    conv = (mesh_fft / sf2) * g
    ntot = float(np.prod(mesh_dimensions))
    phi = (_irfftn(conv, mesh_dimensions, axes) * ntot).astype(dt)
    raw = spline_gather(pos, phi, cell, spline_order, batch_idx)
    corr = pme_energy_corrections(raw, charges, cell, alpha, batch_idx, with_charge_grad=compute_charge_gradients)
    energies, cgrads = (corr if compute_charge_gradients else (corr, None))
    out = (energies,)
    if compute_forces:
        comps = [_irfftn(-1j * kvec[..., d] * conv, mesh_dimensions, axes) * ntot for d in range(3)]
        efield = np.stack(comps, -1).astype(dt)
        out += ((2.0 * spline_gather_vec3(pos, charges, efield, cell, spline_order, batch_idx)).astype(dt),)
    if compute_charge_gradients:
        out += (cgrads,)
    return out if len(out) > 1 else out[0]


def particle_mesh_ewald(positions, charges, cell, alpha, mesh_dimensions, spline_order=4, batch_idx=None,
                        neighbor_matrix=None, neighbor_matrix_shifts=None, idx_j=None, neighbor_ptr=None,
                        neighbor_shifts=None, mask_value=None, compute_forces=False, compute_charge_gradients=False):
    """pme.py:1673-1994 (real + reciprocal)."""
    n = np.asarray(positions).shape[0]
    mv = n if mask_value is None else mask_value
    rs = ewald_real_space(positions, charges, cell, alpha, neighbor_matrix, neighbor_matrix_shifts, idx_j, neighbor_ptr,
                          neighbor_shifts, mv, batch_idx, compute_forces, compute_charge_gradients)
    rec = pme_reciprocal_space(positions, charges, cell, alpha, mesh_dimensions, spline_order, batch_idx, compute_forces,
                               compute_charge_gradients)
    if isinstance(rs, tuple):
        return tuple(a + b for a, b in zip(rs, rec))
    return rs + rec


def generate_k_vectors_ewald_summation(cell, k_cutoff):
    """Half-space Miller set times 2 pi cell^-T (interactions/electrostatics/k_vectors.py:19-164); no spherical cut, as the reference."""
    cell = np.asarray(cell, np.float64).reshape(3, 3)
    bound = np.ceil(k_cutoff * np.linalg.norm(cell, axis=-1) / (2 * math.pi)).astype(int)
    axes = [np.fft.fftfreq(2 * b + 1) * (2 * b + 1) for b in bound]
    mil = np.stack([g.ravel() for g in np.meshgrid(*axes, indexing="ij")], axis=1)
    h, k, l = mil.T
    mil = mil[(h > 0) | ((h == 0) & (k > 0)) | ((h == 0) & (k == 0) & (l > 0))]
    return mil @ (2 * math.pi * np.linalg.inv(cell.T))


def ewald_reciprocal_space(positions, charges, cell, k_vectors, alpha):
    """Explicit-k reciprocal Ewald, single system, float64 (ewald_kernels.py:1496-1980 + the charge-gradient corrections of
    ewald.py:1761-1766).  Returns per-atom (energies, forces, charge_gradients)."""
    pos, q = np.asarray(positions, np.float64), np.asarray(charges, np.float64)
    cell, kv = np.asarray(cell, np.float64).reshape(3, 3), np.asarray(k_vectors, np.float64).reshape(-1, 3)
    vol = abs(np.linalg.det(cell))
    k2 = (kv * kv).sum(1)
    ok = k2 >= 1e-10                                    # :1578 skips k ~ 0 (structure factor stays 0)
    green = np.zeros_like(k2)
    green[ok] = np.exp(-k2[ok] * 0.25 / alpha**2) / k2[ok] * 8.0 * math.pi / vol   # :1583, 8 pi: half-space set
    ph = kv @ pos.T                                     # [K, N]
    c, s = np.cos(ph), np.sin(ph)
    s_re, s_im = green * (c @ q), green * (s @ q)       # :1610-1615
    s_re[~ok] = 0.0
    s_im[~ok] = 0.0
    phi = s_re @ c + s_im @ s                           # :1650-1655
    fs = s_re[:, None] * s - s_im[:, None] * c          # :1815
    total_charge = q.sum() / vol if len(kv) > 1 else 0.0  # accumulated by the k_idx == 1 thread only (:1594)
    e = 0.5 * q * phi - alpha * q * q / math.sqrt(math.pi) - math.pi * q * total_charge / (2 * alpha**2)   # :1692-1720
    f = q[:, None] * (fs.T @ kv)
    cg = phi - 2 * alpha / math.sqrt(math.pi) * q - math.pi / alpha**2 * total_charge
    return e, f, cg


def _erfc_as_np(x):
    """Abramowitz-Stegun 7.1.26 in float64, x >= 0 (math/math.py:52-93)."""
    t = 1.0 / (1.0 + 0.3275911 * x)
    poly = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))))
    return poly * np.exp(-x * x)


def coulomb(positions, charges, cell, cutoff, alpha=0.0, neighbor_list=None, neighbor_ptr=None, neighbor_shifts=None,
            neighbor_matrix=None, neighbor_matrix_shifts=None, fill_value=None, batch_idx=None, compute_forces=True):
    """Cut-off Coulomb over stored entries, float64 (interactions/electrostatics/coulomb.py:133-708).  Returns (energies, forces)
    -- forces None when compute_forces is False.  Entry (i, j): r_ij = r_i - r_j - cell^T S (:185, :248), skipped when r >= cutoff
    or r < 1e-10 (:189); matrix padding is j >= fill_value or j >= N (:325).  Prefactor 1/2 q_i q_j, except q_i q_j in the
    energy-only matrix kernels (:340, :623).  Forces: +f_ij on i, -f_ij on j for every entry (:282-286)."""
    pos, q = np.asarray(positions, np.float64), np.asarray(charges, np.float64)
    cells = np.asarray(cell, np.float64).reshape(-1, 3, 3)
    n = pos.shape[0]
    if neighbor_list is not None:
        ptr = np.asarray(neighbor_ptr, np.int64)
        j = np.asarray(neighbor_list, np.int64)[1]
        i = np.repeat(np.arange(n), np.diff(ptr))     # the kernels walk rows by neighbor_ptr; neighbor_list[0] is not read (:176)
        j, sh = j[: len(i)], np.asarray(neighbor_shifts, np.float64)[: len(i)]
        pref = 0.5
    else:
        nm = np.asarray(neighbor_matrix, np.int64)
        fv = n if fill_value is None else int(fill_value)
        i = np.repeat(np.arange(n), nm.shape[1])
        j, sh = nm.ravel(), np.asarray(neighbor_matrix_shifts, np.float64).reshape(-1, 3)
        keep = ~((j >= fv) | (j >= n))
        i, j, sh = i[keep], j[keep], sh[keep]
        pref = 0.5 if compute_forces else 1.0
    sys_i = np.zeros(len(i), np.int64) if batch_idx is None else np.asarray(batch_idx, np.int64)[i]
    shift_vec = np.einsum("eab,ea->eb", cells[sys_i], sh)           # transpose(cell) @ S
    rij = pos[i] - pos[j] - shift_vec
    r = np.sqrt((rij * rij).sum(1))
    keep = ~((r >= cutoff) | (r < 1e-10))
    i, j, rij, r = i[keep], j[keep], rij[keep], r[keep]
    qq = q[i] * q[j]
    if alpha > 0.0:
        ar = alpha * r
        ec = _erfc_as_np(ar)
        phi = ec / r
        fmr = ec / (r * r * r) + 1.1283791670955126 * alpha * np.exp(-ar * ar) / (r * r)
    else:
        phi = 1.0 / r
        fmr = 1.0 / (r * r * r)
    e = np.zeros(n)
    np.add.at(e, i, pref * qq * phi)
    if not compute_forces:
        return e, None
    fij = (0.5 * qq * fmr)[:, None] * rij
    f = np.zeros((n, 3))
    np.add.at(f, i, fij)
    np.add.at(f, j, -fij)
    return e, f


def explicit_ewald(positions, charges, cell, alpha, kmax, rcut_images=None, exact_erfc=True):
    """Independent float64 Ewald sum (structure-factor reciprocal part + direct real-space image sum).

    Returns (total energy, forces[N,3]).  Coulomb constant 1.  Used only to pin the PME restatement.
    """
    pos = np.asarray(positions, np.float64)
    q = np.asarray(charges, np.float64)
    cell = np.asarray(cell, np.float64).reshape(3, 3)
    vol = abs(np.linalg.det(cell))
    n = len(q)
    recip = 2.0 * math.pi * np.linalg.inv(cell)  # columns are reciprocal vectors
    e_rec = 0.0
    f = np.zeros((n, 3))
    rng = range(-kmax, kmax + 1)
    for h in rng:
        for k in rng:
            for l in rng:
                if h == 0 and k == 0 and l == 0:
                    continue
                kv = recip @ np.array([h, k, l], float)
                k2 = kv @ kv
                ph = pos @ kv
                sr, si = (q * np.cos(ph)).sum(), (q * np.sin(ph)).sum()
                a = (2.0 * math.pi / vol) * math.exp(-k2 / (4 * alpha * alpha)) / k2
                e_rec += a * (sr * sr + si * si)
                f += (2.0 * a) * (q * (np.sin(ph) * sr - np.cos(ph) * si))[:, None] * kv[None, :]
    e_self = -alpha / math.sqrt(math.pi) * (q * q).sum()
    e_bg = -math.pi * q.sum() ** 2 / (2 * alpha * alpha * vol)
    # real space over images
    from math import erfc as _erfc
    e_real = 0.0
    inv = np.linalg.inv(cell)
    rc = rcut_images if rcut_images is not None else 6.0 / alpha
    reps = [int(math.ceil(rc * np.linalg.norm(inv[:, d]))) for d in range(3)]
    for sx in range(-reps[0], reps[0] + 1):
        for sy in range(-reps[1], reps[1] + 1):
            for sz in range(-reps[2], reps[2] + 1):
                sh = np.array([sx, sy, sz], float) @ cell
                dr = pos[None, :, :] - pos[:, None, :] + sh[None, None, :]
                d = np.sqrt((dr * dr).sum(-1))
                mask = (d > 1e-10) & (d < rc)
                dd = d[mask]
                er = np.array([_erfc(alpha * x) for x in dd])
                qq = (q[:, None] * q[None, :])[mask]
                e_real += 0.5 * (qq * er / dd).sum()
                fm = qq * (er / dd**3 + 2 * alpha / math.sqrt(math.pi) * np.exp(-(alpha * dd) ** 2) / dd**2)
                fv = np.zeros_like(dr)
                fv[mask] = -fm[:, None] * dr[mask]
                f += fv.sum(1)
    return e_rec + e_self + e_bg + e_real, f

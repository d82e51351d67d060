"""Seeded synthetic systems shared by tests and bench.py (numpy only; no reference code is read at run time).

Generators restate the *recipes* of the reference's test/benchmark inputs:
  * HoTlPd / SiCu crystals           test/neighborlist/test_utils.py:252-301 (data values)
  * random / triclinic boxes         test/neighborlist/test_utils.py:85-250 (recipe: uniform in the cell, seed 42)
  * jittered FCC boxes, molecules    benchmarks/systems.py:874-1059 (recipe)
  * D3 test molecules                test/interactions/dispersion/conftest.py:216-507 (data values)
"""
import numpy as np

HOTLPD_POS = np.array([
    [4.64882481e00, 0.0, 1.87730266e00], [1.56295308e00, 2.70711418e00, 1.87730266e00],
    [-2.32441241e00, 4.02600045e00, 1.87730266e00], [2.08725046e00, 0.0, 0.0],
    [2.84374025e00, 4.92550268e00, 0.0], [-1.04362523e00, 1.80761195e00, 0.0],
    [-3.88994531e-06, 4.48874533e00, 0.0], [3.88736937e00, 2.24436930e00, 0.0], [0.0, 0.0, 1.87730266e00]])
HOTLPD_CELL = np.array([[7.77473097, 0.0, 0.0], [-3.88736549, 6.73311463, 0.0], [0.0, 0.0, 3.75460533]])
SICU_POS = np.array([[0.0, 0.0, 0.0], [3.0, 0.0, 0.0]])
SICU_CELL = np.array([[0.0, 3.0, 3.0], [3.0, 0.0, 3.0], [3.0, 3.0, 0.0]])
# known answers: test/neighborlist/test_cell_list.py:391-419, test_batch_cell_list.py:516-541
HOTLPD_COUNTS = {1.0: [0] * 9, 4.0: [13, 13, 13, 14, 14, 14, 11, 11, 11], 6.0: [42, 42, 42, 36, 36, 36, 41, 41, 44]}
SICU_COUNTS = {1.0: [0, 0], 4.0: [6, 6], 6.0: [26, 26]}


def random_box(n, box=10.0, seed=42, dtype=np.float32, triclinic=False, outside=False):
    g = np.random.default_rng(seed)
    cell = np.eye(3) * box
    if triclinic:
        cell = np.array([[box, 0.0, 0.0], [0.25 * box, 0.9 * box, 0.0], [0.1 * box, -0.2 * box, 1.1 * box]])
    frac = g.uniform(0.0, 1.0, (n, 3))
    if outside:  # atoms outside the cell exercise the per-atom periodic wrap bookkeeping
        frac += g.integers(-2, 3, (n, 3))
    return (frac @ cell).astype(dtype), cell.astype(dtype)


def fcc_box(n_atoms, a=4.0, jitter=0.05, seed=1234, dtype=np.float32):
    """Jittered FCC box, first n_atoms sites, alternating +-1 charges, Z in {6, 8}."""
    nc = int(np.ceil((n_atoms / 4.0) ** (1.0 / 3.0)))
    basis = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    ijk = np.stack(np.meshgrid(np.arange(nc), np.arange(nc), np.arange(nc), indexing="ij"), -1).reshape(-1, 3)
    sites = (ijk[:, None, :] + basis[None, :, :]).reshape(-1, 3)[:n_atoms]
    g = np.random.default_rng(seed)
    pos = sites * a + g.normal(0.0, jitter, sites.shape)
    cell = np.eye(3) * (nc * a)
    # FCC is not bipartite: alternate by (i + j + k + basis index) as the reference benchmark generator does
    par = (ijk.sum(1)[:, None] + np.arange(4)[None, :]).reshape(-1)[:n_atoms] % 2
    q = np.where(par == 0, 1.0, -1.0)
    q[-1] -= q.sum()  # neutral
    numbers = np.where(par == 0, 6, 8).astype(np.int32)
    pos = np.mod(pos, nc * a)
    return pos.astype(dtype), cell.astype(dtype), q.astype(dtype), numbers


def molecule(n_atoms=512, density=0.05, min_dist=1.0, seed=2000, dtype=np.float32):
    """Uniform random atoms in a cube of side (n/density)^(1/3) with min-distance rejection, Z in {1,6,7,8}."""
    side = (n_atoms / density) ** (1.0 / 3.0)
    g = np.random.default_rng(seed)
    pts = np.empty((0, 3))
    while len(pts) < n_atoms:
        cand = g.uniform(0, side, (n_atoms, 3))
        for c in cand:
            if len(pts) == 0 or np.min(np.sum((pts - c) ** 2, 1)) >= min_dist**2:
                pts = np.vstack([pts, c])
                if len(pts) == n_atoms:
                    break
    numbers = g.choice(np.array([1, 6, 7, 8], np.int32), n_atoms)
    return pts.astype(dtype), numbers.astype(np.int32), side


def d3_test_tables(z_max: int = 17, seed: int | None = None):
    """Analytic test tables of the reference's own test-suite (test/interactions/dispersion/conftest.py:38-160):
    c6ab = 10 Zi Zj (1 + 0.1p + 0.1q), cn_ref = (p/4) cnmax[Zi].  For z_max > 17 (benchmarks) the element
    vectors are extended with a seeded generator (real Grimme tables are not in the reference repo: SURVEY F8)."""
    nz = z_max + 1
    rcov = np.zeros(nz, np.float32)
    cnmax = np.zeros(nz, np.float32)
    r4r2 = np.zeros(nz, np.float32)
    rcov[:10] = [0.0, 0.6, 0.8, 2.8, 2.0, 1.6, 1.4, 1.3, 1.2, 1.5]
    cnmax[:10] = [0.0, 1.5, 1.0, 6.0, 4.0, 4.0, 4.0, 4.0, 2.5, 1.5]
    r4r2[:10] = [0.0, 2.0, 1.5, 10.0, 6.0, 5.0, 4.5, 4.0, 3.5, 3.0]
    if nz > 10:
        rcov[10], cnmax[10], r4r2[10] = 1.5, 1.0, 4.5
    if nz > 17:
        rcov[17], cnmax[17], r4r2[17] = 1.8, 2.0, 8.0
    if seed is not None:
        g = np.random.default_rng(seed)
        for z in range(1, nz):
            if rcov[z] == 0.0:
                rcov[z] = g.uniform(1.0, 3.0)
                cnmax[z] = g.uniform(1.0, 6.0)
                r4r2[z] = g.uniform(2.0, 10.0)
    p = np.arange(5, dtype=np.float32)
    zi = np.arange(nz, dtype=np.float32)
    c6ab = (10.0 * zi[:, None, None, None] * zi[None, :, None, None]
            * (1.0 + 0.1 * p[None, None, :, None] + 0.1 * p[None, None, None, :])).astype(np.float32)
    cn_ref = np.broadcast_to(((p / 4.0)[None, None, :, None] * cnmax[:, None, None, None]), (nz, nz, 5, 5)).astype(np.float32).copy()
    return {"rcov": rcov, "r4r2": r4r2, "c6ab": c6ab, "cn_ref": cn_ref}

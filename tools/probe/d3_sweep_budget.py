"""Diagnostic: for the seeds of tests/test_sweep_gpu.py::test_d3_sweep, force error of the product / IEEE build / reference-order oracle
against the wide-sum oracle, plus a fp64-pair-arithmetic reference (the oracle run on float64 positions)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from oracle import oracle as O
import tests.test_sweep_gpu as T
from tests.test_d3_gpu import _ieee_lib
from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
from nvalchemiops.neighborlist import batch_cell_list
d3mod = sys.modules["nvalchemiops.interactions.dispersion.dftd3"]
_t = T._t
for seed in range(10):
    g = np.random.default_rng(3000 + seed)
    dtype = np.float64 if seed % 4 == 3 else np.float32
    periodic = seed % 3 != 2
    nsys = int(g.integers(1, 5))
    sizes = [int(g.choice([1, 5, 40, 150, 400])) for _ in range(nsys)]
    pos, cell, pbc, bi = T._random_batch(g, dtype, periodic, sizes)
    zmax = [17, 17, 30, 9][seed % 4]
    t = O.d3_test_tables(zmax, seed=50 + seed)
    if seed % 5 == 1:
        t["cn_ref"] = t["cn_ref"] * (1.0 + 0.05 * g.random(t["cn_ref"].shape).astype(np.float32))
    p = D3Parameters(rcov=_t(t["rcov"]), r4r2=_t(t["r4r2"]), c6ab=_t(t["c6ab"]), cn_ref=_t(t["cn_ref"]))
    z = g.integers(1, zmax + 1, len(pos)).astype(np.int32)
    if seed % 3 == 0:
        z[g.integers(0, len(z))] = 0
    cutoff = float(g.uniform(6.0, 14.0))
    fp = dict(a1=float(g.uniform(0.3, 0.5)), a2=float(g.uniform(3.5, 5.0)), s8=float(g.uniform(0.7, 2.0)), k1=16.0, k3=-4.0, s6=1.0)
    nm, num, sh = batch_cell_list(_t(pos), cutoff, _t(cell), _t(pbc), _t(bi), max_neighbors=2048)
    kw = dict(neighbor_matrix=nm) | (dict(neighbor_matrix_shifts=sh, cell=_t(cell)) if periodic else {})
    okw = dict(neighbor_matrix=nm.cpu().numpy()) | (dict(neighbor_matrix_shifts=sh.cpu().numpy(), cell=cell) if periodic else {})
    vir = dict(compute_virial=True) if periodic else {}
    fast = dftd3(_t(pos), _t(z), d3_params=p, batch_idx=_t(bi), num_systems=nsys, **kw, **fp, **vir)
    d3mod._LIB_OVERRIDE = _ieee_lib()
    try:
        ieee = dftd3(_t(pos), _t(z), d3_params=p, batch_idx=_t(bi), num_systems=nsys, **kw, **fp, **vir)
    finally:
        d3mod._LIB_OVERRIDE = None
    ref = O.dftd3(pos, z, t, batch_idx=bi, num_systems=nsys, **okw, **fp, **vir)
    with O.d3_wide_sums():
        wide = O.dftd3(pos, z, t, batch_idx=bi, num_systems=nsys, **okw, **fp, **vir)
    F = np.asarray(wide[1], np.float64)
    cn = np.asarray(wide[2], np.float64)
    row = {"seed": seed, "atoms": len(pos), "maxF": float(np.abs(F).max()), "maxCN": float(cn.max()),
           "fast": float(np.abs(fast[1].cpu().numpy() - F).max()), "ieee": float(np.abs(ieee[1].cpu().numpy() - F).max()),
           "ref_order": float(np.abs(np.asarray(ref[1], np.float64) - F).max()),
           "cn_fast": float(np.abs(fast[2].cpu().numpy() - cn).max()), "E_fast_rel": float(np.abs(fast[0].cpu().numpy() - wide[0]).max() / max(np.abs(wide[0]).max(), 1e-30))}
    if periodic:
        V = np.asarray(wide[3], np.float64)
        row.update(maxV=float(np.abs(V).max()), vir_fast=float(np.abs(fast[3].cpu().numpy() - V).max()), vir_ieee=float(np.abs(ieee[3].cpu().numpy() - V).max()),
                   vir_ref=float(np.abs(np.asarray(ref[3], np.float64) - V).max()))
    print(json.dumps(row), flush=True)

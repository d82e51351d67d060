"""BASELINE config 5 (1024 x 2000-atom periodic boxes, full nlist + D3 + PME step, 128 systems per GPU) -- parity of what
`bench.py --workload c5` runs.  The boxes are SMALLER than twice the D3 cutoff (L = 32 A = 60.5 Bohr < 80 Bohr), so every row
holds several periodic images of the same neighbour (entries (i, j, S) and (i, j, S') with S != S').

  * an 8-system shard against the CPU oracle, stage by stage: both batch neighbour lists bit-exact, batch PME (mesh 32^3 per system,
    spline order 5 vs the oracle's extended mode, fp64) to 1e-10, batch D3 (E + F + CN + virial, fp32) at the D3 tolerances;
  * the full 128-system shard through size-independent properties: batch == the same systems run alone, zero net force per system,
    symmetric virial, counts equal for the replicated geometry."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import systems as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BOHR = 1.8897261246
BJ = dict(a1=0.4289, a2=4.4407, s8=0.7875)


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def _shard(nsys, atoms=2000, seed=1234):
    parts = [S.fcc_box(atoms, seed=seed + 17 * b, dtype=np.float64) for b in range(nsys)]
    pos = np.concatenate([p[0] for p in parts])
    cell = np.stack([p[1] for p in parts])
    q = np.concatenate([p[2] for p in parts])
    z = np.concatenate([p[3] for p in parts])
    bi = np.repeat(np.arange(nsys, dtype=np.int32), atoms)
    return pos, cell, q, z, bi


def _step(pos, cell, q, z, bi, nsys, params, md=2560):
    from nvalchemiops.interactions.dispersion import dftd3
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
    from nvalchemiops.neighborlist import batch_cell_list

    pbc = torch.ones((nsys, 3), dtype=torch.bool, device=DEV)
    tp, tc, tb = _t(pos), _t(cell), _t(bi)
    nm, num, sh = batch_cell_list(tp, 9.0, tc, pbc, tb, max_neighbors=256)
    e_pme, f_pme = particle_mesh_ewald(tp, _t(q), tc, alpha=torch.full((nsys,), 0.35, dtype=torch.float64, device=DEV), mesh_dimensions=(32, 32, 32),
                                       spline_order=5, batch_idx=tb, neighbor_matrix=nm, neighbor_matrix_shifts=sh, compute_forces=True)
    pb, cb = _t((pos * BOHR).astype(np.float32)), _t((cell * BOHR).astype(np.float32))
    dm, dnum, dsh = batch_cell_list(pb, 40.0, cb, pbc, tb, max_neighbors=md)
    assert int(dnum.max()) <= md and int(num.max()) <= 256
    d3 = dftd3(pb, _t(z), d3_params=params, neighbor_matrix=dm, neighbor_matrix_shifts=dsh, cell=cb, batch_idx=tb, compute_virial=True,
               num_systems=nsys, fill_value=len(pos), **BJ)
    return (nm, num, sh), (e_pme, f_pme), (dm, dnum, dsh), d3


def _params():
    from nvalchemiops.interactions.dispersion import D3Parameters

    t = O.d3_test_tables(94, seed=7)
    return t, D3Parameters(rcov=_t(t["rcov"]), r4r2=_t(t["r4r2"]), c6ab=_t(t["c6ab"]), cn_ref=_t(t["cn_ref"]))


def _pairs(nm, num, sh):
    return O.canonical_pairs(nm.cpu().numpy(), num.cpu().numpy(), sh.cpu().numpy())


def test_c5_shard_of_8_systems_vs_oracle():
    nsys = 8
    pos, cell, q, z, bi = _shard(nsys)
    tables, params = _params()
    (nm, num, sh), (e_pme, f_pme), (dm, dnum, dsh), d3 = _step(pos, cell, q, z, bi, nsys, params)
    pbc = np.ones((nsys, 3), bool)
    # 1. neighbour lists: bit-exact sets and counts, multi-image rows included
    onm, onum, osh = O.cell_list(pos, 9.0, cell, pbc, batch_idx=bi, max_neighbors=256)
    assert np.array_equal(num.cpu().numpy(), onum) and np.array_equal(_pairs(nm, num, sh), O.canonical_pairs(onm, onum, osh))
    pb, cb = (pos * BOHR).astype(np.float32), (cell * BOHR).astype(np.float32)
    odm, odnum, odsh = O.cell_list(pb, 40.0, cb, pbc, batch_idx=bi, max_neighbors=2560)
    assert np.array_equal(dnum.cpu().numpy(), odnum) and np.array_equal(_pairs(dm, dnum, dsh), O.canonical_pairs(odm, odnum, odsh))
    row0 = odm[0, : odnum[0]]
    assert len(np.unique(row0)) < len(row0), "L < 2 rc: a row must list several periodic images of the same neighbour"
    # 2. batch PME, order 5, fp64 (extended oracle)
    with O.extended_splines():
        oe, of = O.particle_mesh_ewald(pos, q, cell, np.full(nsys, 0.35), (32, 32, 32), 5, batch_idx=bi, neighbor_matrix=onm,
                                       neighbor_matrix_shifts=osh, mask_value=len(pos), compute_forces=True)
    assert np.abs(e_pme.cpu().numpy() - oe).max() <= 1e-10 * np.abs(oe).max() + 1e-12
    assert np.abs(f_pme.cpu().numpy() - of).max() <= 1e-10 * np.abs(of).max() + 1e-12
    # 3. batch D3 with virial, fp32: vs the wide-sum oracle at the reference's rtol = atol = 1e-6
    with O.d3_wide_sums():
        ref = O.dftd3(pb, z, tables, neighbor_matrix=odm, neighbor_matrix_shifts=odsh, cell=cb, batch_idx=bi, compute_virial=True,
                      num_systems=nsys, **BJ)
    for got, want, what in zip(d3, ref, ("energy", "forces", "coord_num", "virial")):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-6, atol=1e-6, err_msg=what)


def test_c5_full_shard_128_systems_properties():
    nsys = 128
    pos, cell, q, z, bi = _shard(nsys)
    tables, params = _params()
    (nm, num, sh), (e_pme, f_pme), (dm, dnum, dsh), (e, f, cn, vir) = _step(pos, cell, q, z, bi, nsys, params)
    assert all(torch.isfinite(x).all() for x in (e_pme, f_pme, e, f, cn, vir))
    seg = torch.as_tensor(bi, device=DEV).long()
    # zero net force per system (D3: exact pair antisymmetry up to fp32 rounding; PME: mesh accuracy)
    net = torch.zeros((nsys, 3), dtype=torch.float64, device=DEV).index_add_(0, seg, f.double())
    assert float(net.abs().max()) < 2e-4 * float(f.abs().max()) * 2000 ** 0.5
    assert float((vir - vir.transpose(1, 2)).abs().max()) <= 1e-5 * float(vir.abs().max())
    # a few systems run ALONE give what they gave inside the batch (D3: rtol = atol = 1e-6, test_dftd3.py:2386-2391; PME: 1e-10)
    for b in (0, 77, 127):
        sl = slice(2000 * b, 2000 * (b + 1))
        _, (e1, f1), _, (ed, fd, cnd, vd) = _step(pos[sl], cell[b:b + 1], q[sl], z[sl], np.zeros(2000, np.int32), 1, params)
        np.testing.assert_allclose(e_pme[sl].cpu().numpy(), e1.cpu().numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(f_pme[sl].cpu().numpy(), f1.cpu().numpy(), rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(e[b].cpu().numpy(), ed[0].cpu().numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(f[sl].cpu().numpy(), fd.cpu().numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(cn[sl].cpu().numpy(), cnd.cpu().numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(vir[b].cpu().numpy(), vd[0].cpu().numpy(), rtol=1e-6, atol=2e-6)

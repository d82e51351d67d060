"""`@torch.compile(mode="default", fullgraph=True)` (Inductor) over the PUBLIC neighbour-list / DFT-D3 API on the device, results equal to
the eager calls.  Restates the compiled MD step of the reference's example
(examples/neighborlist/04_neighbors_list_torch_compile_performance.py:218-289 high-level `cell_list` with pre-allocated outputs,
:296-399 low-level `build_cell_list` + `query_cell_list`), which the reference supports because its wrappers are thin Python over
`torch.library.custom_op`s (neighborlist/cell_list.py:725-736, 892-895, 1037-1192; dftd3.py:1792-1796)."""
import numpy as np
import pytest
import torch

from tests import systems as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CUTOFF, DT = 3.0, 1e-3


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def _lj_forces(positions, nm, sh, cell):
    """Lennard-Jones forces / energy over a padded neighbour matrix (padding -1), plain torch: the part Inductor compiles."""
    mask = nm >= 0
    dr = positions[nm.long()] - positions.unsqueeze(1) + sh.to(positions.dtype) @ cell[0]
    r2 = (dr * dr).sum(-1).clamp(min=1e-10)
    s6 = (1.0 / r2) ** 3
    fmag = torch.where(mask, 24.0 / r2 * (s6 - 2 * s6 * s6), torch.zeros_like(r2))
    return (fmag.unsqueeze(-1) * dr).sum(1), torch.where(mask, 4 * (s6 * s6 - s6), torch.zeros_like(r2)).sum() * 0.5


def _md_system(n=500, box=15.0):
    pos, cell, _, _ = S.fcc_box(n, a=box / 5.0, jitter=0.02, seed=9, dtype=np.float32)  # 5 x 5 x 5 FCC cells = 500 atoms
    cell = np.eye(3, dtype=np.float32) * box
    g = torch.Generator(device="cpu").manual_seed(3)
    vel = torch.randn(n, 3, generator=g).to(DEV)
    return _t(pos % box), _t(cell).reshape(1, 3, 3), torch.tensor([True, True, True], device=DEV), vel - vel.mean(0)


def test_low_level_md_step_fullgraph_inductor_equals_eager():
    from nvalchemiops.neighborlist import (allocate_cell_list, build_cell_list, estimate_cell_list_sizes, estimate_max_neighbors,
                                           query_cell_list)

    pos, cell, pbc, vel = _md_system()
    n = pos.shape[0]
    ncell, radius = estimate_cell_list_sizes(cell, pbc, CUTOFF)
    m = estimate_max_neighbors(CUTOFF)

    def make_step():
        cache = allocate_cell_list(n, ncell, radius, pos.device)
        nm = torch.full((n, m), -1, dtype=torch.int32, device=DEV)
        sh = torch.zeros((n, m, 3), dtype=torch.int32, device=DEV)
        num = torch.zeros(n, dtype=torch.int32, device=DEV)

        def md_step(positions, velocities):
            build_cell_list(positions, CUTOFF, cell, pbc, *cache)
            nm.fill_(-1), sh.fill_(0), num.fill_(0)
            query_cell_list(positions, CUTOFF, cell, pbc, *cache, nm, sh, num)
            f, _ = _lj_forces(positions, nm, sh, cell)
            velocities = velocities + 0.5 * DT * f
            positions = (positions + DT * velocities) % cell[0, 0, 0]
            build_cell_list(positions, CUTOFF, cell, pbc, *cache)
            nm.fill_(-1), sh.fill_(0), num.fill_(0)
            query_cell_list(positions, CUTOFF, cell, pbc, *cache, nm, sh, num)
            f, potential = _lj_forces(positions, nm, sh, cell)
            velocities = velocities + 0.5 * DT * f
            return positions, velocities, potential, 0.5 * (velocities**2).sum()

        return md_step, (nm, sh, num, cache)

    eager_step, eager_state = make_step()
    traced_step, traced_state = make_step()
    compiled = torch.compile(traced_step, mode="default", fullgraph=True)  # Inductor, as the reference's example
    pe, ve, pc, vc = pos.clone(), vel.clone(), pos.clone(), vel.clone()
    for _ in range(3):  # a few steps: the second call re-uses the compiled graph on new positions
        pe, ve, ue, ke = eager_step(pe, ve)
        pc, vc, uc, kc = compiled(pc, vc)
    assert int(eager_state[2].sum()) > 10 * n, "the system must have neighbours"
    # the lists the two runs left behind are the same lists, bit for bit (deterministic row order on this build)
    assert torch.equal(eager_state[2], traced_state[2]) and torch.equal(eager_state[0], traced_state[0]) and torch.equal(eager_state[1], traced_state[1])
    for a, b in zip(eager_state[3], traced_state[3]):
        assert torch.equal(a, b)
    assert torch.allclose(pe, pc, rtol=1e-5, atol=1e-5) and torch.allclose(ve, vc, rtol=1e-4, atol=1e-4)
    assert torch.allclose(ue, uc, rtol=1e-4) and torch.allclose(ke, kc, rtol=1e-5)


def test_high_level_entry_points_fullgraph_inductor_equal_eager():
    from nvalchemiops.neighborlist import batch_cell_list, cell_list, neighbor_list

    pos, cell, pbc, vel = _md_system()
    n, m = pos.shape[0], 96

    def outputs():
        return (torch.full((n, m), -1, dtype=torch.int32, device=DEV), torch.zeros((n, m, 3), dtype=torch.int32, device=DEV),
                torch.zeros(n, dtype=torch.int32, device=DEV))

    @torch.compile(mode="default", fullgraph=True)
    def step(positions, velocities, nm, sh, num):  # the reference example's first variant (:218-289)
        positions = (positions + DT * velocities) % cell[0, 0, 0]
        nm2, num2, sh2 = cell_list(positions, CUTOFF, cell, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num, fill_value=-1)
        f, u = _lj_forces(positions, nm2, sh2, cell)
        return positions, f, u, num2.sum()

    o = outputs()
    p1, f1, u1, tot = step(pos, vel, *o)
    e = outputs()
    pe = (pos + DT * vel) % cell[0, 0, 0]
    cell_list(pe, CUTOFF, cell, pbc, neighbor_matrix=e[0], neighbor_matrix_shifts=e[1], num_neighbors=e[2], fill_value=-1)
    fe, ue = _lj_forces(pe, e[0], e[1], cell)
    assert all(torch.equal(a, b) for a, b in zip(o, e)) and int(tot) == int(e[2].sum()) > 0
    assert torch.allclose(f1, fe, rtol=1e-4, atol=1e-4) and torch.allclose(u1, ue, rtol=1e-5)

    # allocating forms: dispatcher (cell list, naive, free space) and a batch of two systems
    bi = torch.tensor([0] * 300 + [1] * (n - 300), dtype=torch.int32, device=DEV)
    cells, pbcs = cell.expand(2, 3, 3).contiguous(), pbc.reshape(1, 3).expand(2, 3).contiguous()
    cases = [lambda p: neighbor_list(p, CUTOFF, cell=cell, pbc=pbc, method="cell_list", max_neighbors=m),
             lambda p: neighbor_list(p, CUTOFF, cell=cell, pbc=pbc, max_neighbors=m, half_fill=True),
             lambda p: neighbor_list(p, CUTOFF, max_neighbors=m),
             lambda p: batch_cell_list(p, CUTOFF, cells, pbcs, bi, max_neighbors=m)]
    for fn in cases:
        torch._dynamo.reset()
        got = torch.compile(fn, mode="default", fullgraph=True)(pos)
        want = fn(pos)
        assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))


def test_dftd3_fullgraph_inductor_equals_eager():
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
    from nvalchemiops.neighborlist import batch_cell_list

    t = S.d3_test_tables(17)
    params = D3Parameters(rcov=_t(t["rcov"]), r4r2=_t(t["r4r2"]), c6ab=_t(t["c6ab"]), cn_ref=_t(t["cn_ref"]))
    p0, c0 = S.random_box(300, 22.0, seed=35, dtype=np.float32)
    p1, c1 = S.random_box(200, 18.0, seed=36, dtype=np.float32, triclinic=True)
    pos, cell = _t(np.concatenate([p0, p1])), _t(np.stack([c0, c1]))
    pbc = torch.ones((2, 3), dtype=torch.bool, device=DEV)
    bi = torch.tensor([0] * 300 + [1] * 200, dtype=torch.int32, device=DEV)
    numbers = _t(np.random.default_rng(0).choice([1, 6, 7, 8], 500).astype(np.int32))
    bj = dict(a1=0.4289, a2=4.4407, s8=0.7875)

    @torch.compile(mode="default", fullgraph=True)
    def energy_and_forces(positions):  # list AND dispersion inside one graph, some torch work on either side
        positions = positions * 1.0
        nm, num, sh = batch_cell_list(positions, 10.0, cell, pbc, bi, max_neighbors=192)
        e, f, cn, vir = dftd3(positions, numbers, d3_params=params, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=cell, batch_idx=bi,
                              num_systems=2, compute_virial=True, **bj)
        return e, f - f.mean(0), cn, vir, num.max()

    e, f, cn, vir, worst = energy_and_forces(pos)
    assert int(worst) <= 192
    nm, num, sh = batch_cell_list(pos, 10.0, cell, pbc, bi, max_neighbors=192)
    e0, f0, cn0, v0 = dftd3(pos, numbers, d3_params=params, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=cell, batch_idx=bi, num_systems=2,
                            compute_virial=True, **bj)
    assert e.shape == (2,) and f.shape == (500, 3) and vir.shape == (2, 3, 3) and e.dtype == torch.float32
    assert torch.equal(e, e0) and torch.equal(cn, cn0) and torch.equal(vir, v0)  # same kernels on the same list: the same bits
    assert torch.allclose(f, f0 - f0.mean(0), rtol=1e-6, atol=1e-9)

    # CSR form + dict parameters through `nvalchemiops::dftd3_nl`
    from nvalchemiops.neighborlist import neighbor_list

    lst, ptr, shl = neighbor_list(pos, 10.0, cell=cell, pbc=pbc, batch_idx=bi, method="batch_cell_list", max_neighbors=192, return_neighbor_list=True)
    tables = {"rcov": params.rcov, "r4r2": params.r4r2, "c6ab": params.c6ab, "cn_ref": params.cn_ref}
    csr = lambda p: dftd3(p, numbers, d3_params=tables, neighbor_list=lst, neighbor_ptr=ptr, unit_shifts=shl, cell=cell, batch_idx=bi,  # noqa: E731
                          num_systems=2, **bj)
    got, want = torch.compile(csr, mode="default", fullgraph=True)(pos), csr(pos)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    assert torch.allclose(got[0], e0, rtol=1e-6)

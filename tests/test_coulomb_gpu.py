"""GPU parity of the cut-off Coulomb ops (`coulomb_energy`, `coulomb_forces`, `coulomb_energy_forces`) against the CPU oracle,
on the cases the reference tests (test/interactions/electrostatics/test_coulomb.py: pair known answers, matrix vs list, batches,
damping, half lists, autograd, float32 inputs, empty inputs, default fill value).

Tolerance: the kernels compute in float64 like the reference (inputs are upcast), so HIP vs oracle differs only by summation
order: 1e-11 of the largest |value|; float32 inputs are compared after the same upcast and the final cast (1e-6 relative)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV) if dtype is None else torch.as_tensor(np.ascontiguousarray(a), device=DEV, dtype=dtype)


def _close(got, ref, what, rel=1e-11):
    got = got.detach().cpu().numpy().astype(np.float64)
    tol = rel * max(np.abs(ref).max(), 1e-30) + 1e-14
    err = np.abs(got - ref).max() if ref.size else 0.0
    assert err <= tol, f"{what}: max err {err:.3e} > {tol:.3e}"


def _system(n, seed=0, box=14.0, triclinic=True):
    g = np.random.default_rng(seed)
    cell = np.eye(3) * box
    if triclinic:
        cell = np.array([[box, 0, 0], [0.2 * box, 0.9 * box, 0], [0.1 * box, -0.15 * box, 1.1 * box]])
    pos = g.uniform(0, 1, (n, 3)) @ cell
    q = g.normal(size=n)
    return pos, cell, q


def _lists(pos, cell, cutoff, half_fill=False, batch_idx=None, batch_ptr=None, max_neighbors=320):
    from nvalchemiops.neighborlist import neighbor_list

    cells = _t(cell).reshape(-1, 3, 3)
    pbc = torch.ones((cells.shape[0], 3), dtype=torch.bool, device=DEV)
    kw = dict(batch_idx=batch_idx, batch_ptr=batch_ptr, method="batch_cell_list") if batch_idx is not None else dict(method="cell_list")
    nm, num, sh = neighbor_list(_t(pos), cutoff, cell=cells, pbc=pbc, max_neighbors=max_neighbors, half_fill=half_fill, **kw)
    assert int(num.max()) <= max_neighbors
    nl, ptr, lsh = neighbor_list(_t(pos), cutoff, cell=cells, pbc=pbc, max_neighbors=max_neighbors, half_fill=half_fill,
                                 return_neighbor_list=True, **kw)
    return nm, sh, nl, ptr, lsh


@pytest.mark.parametrize("alpha", [0.0, 0.3])
@pytest.mark.parametrize("half_fill", [False, True])
def test_list_and_matrix_match_oracle(alpha, half_fill):
    from nvalchemiops.interactions.electrostatics.coulomb import coulomb_energy, coulomb_energy_forces, coulomb_forces

    pos, cell, q = _system(700, seed=3)
    cutoff = 5.0
    nm, sh, nl, ptr, lsh = _lists(pos, cell, cutoff + 0.8, half_fill)  # list reaches beyond the cutoff: the kernel's own r >= cutoff test matters
    P, Q, Cc = _t(pos), _t(q), _t(cell).reshape(1, 3, 3)
    n = pos.shape[0]
    # list format
    e, f = coulomb_energy_forces(P, Q, Cc, cutoff, alpha, neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=lsh)
    oe, of = O.coulomb(pos, q, cell, cutoff, alpha, neighbor_list=nl.cpu().numpy(), neighbor_ptr=ptr.cpu().numpy(), neighbor_shifts=lsh.cpu().numpy())
    _close(e, oe, "list energies"); _close(f, of, "list forces")
    assert float(f.sum(0).abs().max()) < 1e-9 * max(1.0, float(f.abs().max()))  # +f on i, -f on j: momentum is conserved for any list
    _close(coulomb_energy(P, Q, Cc, cutoff, alpha, neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=lsh), oe, "list energy-only")
    _close(coulomb_forces(P, Q, Cc, cutoff, alpha, neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=lsh), of, "forces-only wrapper")
    # matrix format, explicit and default fill value
    for fv in (n, None):
        e2, f2 = coulomb_energy_forces(P, Q, Cc, cutoff, alpha, neighbor_matrix=nm, neighbor_matrix_shifts=sh, fill_value=fv)
        oe2, of2 = O.coulomb(pos, q, cell, cutoff, alpha, neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy(), fill_value=fv)
        _close(e2, oe2, "matrix energies"); _close(f2, of2, "matrix forces")
        _close(e2, oe, "matrix vs list energies", 1e-10); _close(f2, of, "matrix vs list forces", 1e-10)
    # energy-only matrix kernels: q_i q_j without the 1/2 (coulomb.py:340) -- twice the energy+forces energies
    e3 = coulomb_energy(P, Q, Cc, cutoff, alpha, neighbor_matrix=nm, neighbor_matrix_shifts=sh, fill_value=n)
    oe3, _ = O.coulomb(pos, q, cell, cutoff, alpha, neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy(), fill_value=n,
                       compute_forces=False)
    _close(e3, oe3, "matrix energy-only"); _close(e3, 2.0 * oe, "matrix energy-only quirk", 1e-10)
    if not half_fill and alpha > 0:
        # full list + damping == the real-space Ewald op restricted to r < cutoff (same erfc polynomial)
        full = O.ewald_real_space(pos, q, cell, alpha, neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy(), mask_value=n)
        far, _ = O.coulomb(pos, q, cell, 1e9, alpha, neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy(), fill_value=n)
        assert np.abs(full - far).max() < 1e-10 * np.abs(full).max()


@pytest.mark.parametrize("alpha", [0.0, 0.35])
def test_batched_match_oracle_and_single(alpha):
    from nvalchemiops.interactions.electrostatics.coulomb import coulomb_energy, coulomb_energy_forces

    p0, c0, q0 = _system(300, seed=5, box=12.0)
    p1, c1, q1 = _system(420, seed=6, box=15.0, triclinic=False)
    pos, q = np.concatenate([p0, p1]), np.concatenate([q0, q1])
    cells = np.stack([c0, c1])
    bi = np.concatenate([np.zeros(300, np.int32), np.ones(420, np.int32)])
    bptr = np.array([0, 300, 720], np.int32)
    cutoff = 4.5
    nm, sh, nl, ptr, lsh = _lists(pos, cells, cutoff, batch_idx=_t(bi), batch_ptr=_t(bptr))
    P, Q, Cc, B = _t(pos), _t(q), _t(cells), _t(bi)
    e, f = coulomb_energy_forces(P, Q, Cc, cutoff, alpha, neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=lsh, batch_idx=B)
    oe, of = O.coulomb(pos, q, cells, cutoff, alpha, neighbor_list=nl.cpu().numpy(), neighbor_ptr=ptr.cpu().numpy(),
                       neighbor_shifts=lsh.cpu().numpy(), batch_idx=bi)
    _close(e, oe, "batch list energies"); _close(f, of, "batch list forces")
    e2, f2 = coulomb_energy_forces(P, Q, Cc, cutoff, alpha, neighbor_matrix=nm, neighbor_matrix_shifts=sh, batch_idx=B)
    _close(e2, oe, "batch matrix energies", 1e-10); _close(f2, of, "batch matrix forces", 1e-10)
    e3 = coulomb_energy(P, Q, Cc, cutoff, alpha, neighbor_matrix=nm, neighbor_matrix_shifts=sh, batch_idx=B)
    _close(e3, 2.0 * oe, "batch matrix energy-only", 1e-10)  # coulomb.py:623
    # each system alone gives the same numbers (test_coulomb.py:697 single_batch_matches_unbatched)
    nm0, sh0, *_ = _lists(p0, c0, cutoff)
    e0, f0 = coulomb_energy_forces(_t(p0), _t(q0), _t(c0).reshape(1, 3, 3), cutoff, alpha, neighbor_matrix=nm0, neighbor_matrix_shifts=sh0)
    _close(e[:300], e0.cpu().numpy(), "system 0 alone", 1e-10); _close(f[:300], f0.cpu().numpy(), "system 0 alone forces", 1e-10)
    for s, sl in enumerate((slice(0, 300), slice(300, 720))):
        assert float(f[sl].sum(0).abs().max()) < 1e-9 * float(f.abs().max()), f"momentum of system {s}"


def test_reference_pair_known_answers():
    """The analytic expectations of test_coulomb.py:58-92 (energy), :191-221 (half list, short neighbor_ptr), :353-383 (cutoff)."""
    from nvalchemiops.interactions.electrostatics.coulomb import coulomb_energy, coulomb_energy_forces

    pos = torch.tensor([[0.0, 0, 0], [3.0, 0, 0]], dtype=torch.float64, device=DEV)
    q = torch.tensor([1.0, -1.0], dtype=torch.float64, device=DEV)
    cell = torch.eye(3, dtype=torch.float64, device=DEV).reshape(1, 3, 3) * 100
    nl = torch.tensor([[0, 1], [1, 0]], dtype=torch.int32, device=DEV)
    ptr = torch.tensor([0, 1, 2], dtype=torch.int32, device=DEV)
    sh = torch.zeros((2, 3), dtype=torch.int32, device=DEV)
    e = coulomb_energy(pos, q, cell, 10.0, 0.0, neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=sh)
    assert abs(float(e.sum()) + 1.0 / 3.0) < 1e-12
    _, f = coulomb_energy_forces(pos, q, cell, 10.0, 0.0, neighbor_list=nl[:, :1].contiguous(), neighbor_ptr=ptr[:2].contiguous(), neighbor_shifts=sh[:1])
    assert abs(float(f[0, 0]) - 1.0 / 18.0) < 1e-12 and torch.allclose(f[0], -f[1]) and float(f[:, 1:].abs().max()) == 0.0
    far = pos.clone(); far[1, 0] = 15.0
    e, f = coulomb_energy_forces(far, q, cell, 10.0, 0.0, neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=sh)
    assert float(e.abs().max()) == 0.0 and float(f.abs().max()) == 0.0
    # fill_value = -1 pads everything away (j >= fill_value, test_coulomb.py:2255) ; minimum image through the shift (:654)
    nm = torch.tensor([[1], [0]], dtype=torch.int32, device=DEV)
    e = coulomb_energy(pos, q, cell, 10.0, 0.0, neighbor_matrix=nm, neighbor_matrix_shifts=sh.reshape(2, 1, 3), fill_value=-1)
    assert float(e.abs().max()) == 0.0
    p2 = torch.tensor([[0.5, 5, 5], [9.5, 5, 5]], dtype=torch.float64, device=DEV)
    sh2 = torch.tensor([[-1, 0, 0], [1, 0, 0]], dtype=torch.int32, device=DEV)
    e = coulomb_energy(p2, q, cell / 10, 5.0, 0.0, neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=sh2)
    assert abs(float(e.sum()) + 1.0) < 1e-12  # image distance 1.0


@pytest.mark.parametrize("fmt", ["list", "matrix"])
@pytest.mark.parametrize("alpha", [0.0, 0.3])
def test_energy_autograd(fmt, alpha):
    """-dE/dr equals the explicit forces (test_coulomb.py:1062-1160), charge / cell gradients against central differences (:1318)."""
    from nvalchemiops.interactions.electrostatics.coulomb import coulomb_energy_forces

    p0, c0, q0 = _system(90, seed=8, box=9.0)
    p1, c1, q1 = _system(70, seed=9, box=10.0, triclinic=False)
    pos, q, cells = np.concatenate([p0, p1]), np.concatenate([q0, q1]), np.stack([c0, c1])
    bi = _t(np.concatenate([np.zeros(90, np.int32), np.ones(70, np.int32)]))
    cutoff = 4.0
    nm, sh, nl, ptr, lsh = _lists(pos, cells, cutoff + 0.5, batch_idx=bi, batch_ptr=_t(np.array([0, 90, 160], np.int32)))
    kw = dict(neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=lsh) if fmt == "list" else dict(neighbor_matrix=nm, neighbor_matrix_shifts=sh)

    def total(P, Q, Cc):
        e, f = coulomb_energy_forces(P, Q, Cc, cutoff, alpha, batch_idx=bi, **kw)
        return e.sum(), f

    P, Q, Cc = _t(pos).requires_grad_(True), _t(q).requires_grad_(True), _t(cells).requires_grad_(True)
    E, f = total(P, Q, Cc)
    E.backward(retain_graph=True)
    _close(-P.grad, f.detach().cpu().numpy(), "-dE/dr vs explicit forces", 1e-10)
    g = np.random.default_rng(0)
    with torch.no_grad():
        for _ in range(3):
            k, h = int(g.integers(0, 160)), 1e-5
            dq = torch.zeros_like(Q); dq[k] = h
            fd = float(total(P, Q + dq, Cc)[0] - total(P, Q - dq, Cc)[0]) / (2 * h)
            assert abs(fd - float(Q.grad[k])) < 1e-7 * max(1.0, abs(fd)), "charge gradient"
            s, a, b = int(g.integers(0, 2)), int(g.integers(0, 3)), int(g.integers(0, 3))
            dc = torch.zeros_like(Cc); dc[s, a, b] = h
            fd = float(total(P, Q, Cc + dc)[0] - total(P, Q, Cc - dc)[0]) / (2 * h)
            assert abs(fd - float(Cc.grad[s, a, b])) < 1e-6 * max(1.0, abs(fd)), "cell gradient"


def test_float32_inputs_and_empty():
    from nvalchemiops.interactions.electrostatics.coulomb import coulomb_energy, coulomb_energy_forces

    pos, cell, q = _system(256, seed=11)
    pos32, cell32, q32 = pos.astype(np.float32), cell.astype(np.float32), q.astype(np.float32)
    nm, sh, nl, ptr, lsh = _lists(pos32, cell32, 5.0)
    e, f = coulomb_energy_forces(_t(pos32), _t(q32), _t(cell32).reshape(1, 3, 3), 5.0, 0.3, neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=lsh)
    assert e.dtype == torch.float32 and f.dtype == torch.float32  # coulomb.py:1691
    oe, of = O.coulomb(pos32.astype(np.float64), q32.astype(np.float64), cell32.astype(np.float64), 5.0, 0.3, neighbor_list=nl.cpu().numpy(),
                       neighbor_ptr=ptr.cpu().numpy(), neighbor_shifts=lsh.cpu().numpy())
    _close(e, oe, "f32 energies", 1e-6); _close(f, of, "f32 forces", 1e-6)
    # empty neighbour data (test_coulomb.py:964, :1954-2175)
    P, Q, Cc = _t(pos), _t(q), _t(cell).reshape(1, 3, 3)
    n = pos.shape[0]
    z_nl = torch.zeros((2, 0), dtype=torch.int32, device=DEV)
    z_ptr = torch.zeros(n + 1, dtype=torch.int32, device=DEV)
    z_sh = torch.zeros((0, 3), dtype=torch.int32, device=DEV)
    e, f = coulomb_energy_forces(P, Q, Cc, 5.0, 0.0, neighbor_list=z_nl, neighbor_ptr=z_ptr, neighbor_shifts=z_sh)
    assert e.shape == (n,) and f.shape == (n, 3) and float(e.abs().max()) == 0.0 and float(f.abs().max()) == 0.0
    z_nm = torch.full((n, 4), n, dtype=torch.int32, device=DEV)
    e = coulomb_energy(P, Q, Cc, 5.0, 0.0, neighbor_matrix=z_nm, neighbor_matrix_shifts=torch.zeros((n, 4, 3), dtype=torch.int32, device=DEV))
    assert float(e.abs().max()) == 0.0
    e, f = coulomb_energy_forces(P, Q, Cc, 5.0, 0.0, neighbor_matrix=z_nm[:, :0], neighbor_matrix_shifts=torch.zeros((n, 0, 3), dtype=torch.int32, device=DEV))
    assert float(e.abs().max()) == 0.0 and float(f.abs().max()) == 0.0


@pytest.mark.parametrize("fmt", ["list", "matrix"])
@pytest.mark.parametrize("alpha", [0.0, 0.3])
def test_forces_can_be_differentiated(fmt, alpha):
    """Force-matching on the explicit Coulomb forces (the reference lists `forces` in the ops' grad_arrays, coulomb.py:785-790, :937-945):
    d(sum_k w_k . F_k)/d(positions, charges, cell) through `mi_coulomb_forces_bwd` against central differences, full matrix and half list."""
    from nvalchemiops.interactions.electrostatics.coulomb import coulomb_energy_forces

    p0, c0, q0 = _system(60, seed=18, box=9.0)
    p1, c1, q1 = _system(40, seed=19, box=10.0, triclinic=False)
    pos, q, cells = np.concatenate([p0, p1]), np.concatenate([q0, q1]), np.stack([c0, c1])
    bi = _t(np.concatenate([np.zeros(60, np.int32), np.ones(40, np.int32)]))
    cutoff = 4.0
    # the list variant is a HALF list: the scatter (+f on i, -f on j) must carry the adjoint too
    nm, sh, nl, ptr, lsh = _lists(pos, cells, cutoff + 0.5, half_fill=(fmt == "list"), batch_idx=bi, batch_ptr=_t(np.array([0, 60, 100], np.int32)))
    if fmt == "list":
        kw = dict(neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=lsh)
    else:
        kw = dict(neighbor_matrix=nm, neighbor_matrix_shifts=sh)
    w = _t(np.random.default_rng(3).normal(size=(100, 3)))

    def loss(P, Q, Cc):
        return (w * coulomb_energy_forces(P, Q, Cc, cutoff, alpha, batch_idx=bi, **kw)[1]).sum()

    P, Q, Cc = _t(pos).requires_grad_(True), _t(q).requires_grad_(True), _t(cells).requires_grad_(True)
    loss(P, Q, Cc).backward()
    g = np.random.default_rng(1)
    h = 1e-6
    with torch.no_grad():
        for _ in range(4):
            k, d = int(g.integers(0, 100)), int(g.integers(0, 3))
            dp = torch.zeros_like(P); dp[k, d] = h
            fd = float(loss(P + dp, Q, Cc) - loss(P - dp, Q, Cc)) / (2 * h)
            assert abs(fd - float(P.grad[k, d])) < 2e-6 * max(1.0, abs(fd)), ("position", fd, float(P.grad[k, d]))
            dq = torch.zeros_like(Q); dq[k] = h
            fd = float(loss(P, Q + dq, Cc) - loss(P, Q - dq, Cc)) / (2 * h)
            assert abs(fd - float(Q.grad[k])) < 1e-6 * max(1.0, abs(fd)), "charge"
            s_, a, b = int(g.integers(0, 2)), int(g.integers(0, 3)), int(g.integers(0, 3))
            dc = torch.zeros_like(Cc); dc[s_, a, b] = h
            fd = float(loss(P, Q, Cc + dc) - loss(P, Q, Cc - dc)) / (2 * h)
            assert abs(fd - float(Cc.grad[s_, a, b])) < 5e-6 * max(1.0, abs(fd)), ("cell", fd, float(Cc.grad[s_, a, b]))

"""Scenario-by-scenario restatement of the reference's API-level neighbour-list tests (test/neighborlist/test_neighborlist.py,
test_cell_list.py, test_batch_cell_list.py, test_rebuild_detection.py) on the HIP path: the SAME situations and the SAME assertions
(tuple layouts, shapes, dtypes, identity of pre-allocated outputs, exception types), written against this package and, where the
reference compares with an external neighbour finder (vesin / ASE, absent here), against the CPU oracle instead.  Each test names the
reference test it restates; no reference code is used."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.float32, torch.float64]


def _random_system(n, box, dtype, seed=42):
    """Uniform positions in a cubic periodic box (the reference's `create_random_system` fixture shape: positions, [1,3,3] cell, pbc[3])."""
    g = torch.Generator().manual_seed(seed)
    pos = torch.rand((n, 3), generator=g, dtype=dtype) * box
    return pos.to(DEV), (torch.eye(3, dtype=dtype) * box).reshape(1, 3, 3).to(DEV), torch.tensor([True, True, True], device=DEV)


def _two_system_batch(dtype, n0=50, n1=30, box=10.0):
    p0, c0, b0 = _random_system(n0, box, dtype, seed=1)
    p1, c1, b1 = _random_system(n1, box, dtype, seed=2)
    bi = torch.cat([torch.zeros(n0, dtype=torch.int32), torch.ones(n1, dtype=torch.int32)]).to(DEV)
    return (torch.cat([p0, p1]), torch.cat([c0, c1]), torch.stack([b0, b1]), bi, torch.tensor([0, n0, n0 + n1], dtype=torch.int32, device=DEV))


def _pairs(nl, sh=None):
    nl = nl.cpu().numpy()
    s = np.zeros((nl.shape[1], 3), np.int32) if sh is None else sh.cpu().numpy()
    return sorted(zip(nl[0].tolist(), nl[1].tolist(), map(tuple, s.tolist())))


def _oracle_pairs(pos, cell, pbc, cutoff, half=False):
    nm, num, sh = O.cell_list(pos.cpu().numpy(), cutoff, cell.cpu().numpy().reshape(3, 3), np.asarray(pbc.cpu().numpy()).reshape(3),
                              max_neighbors=1024, half_fill=half)
    out = []
    for i in range(nm.shape[0]):
        for k in range(int(num[i])):
            out.append((i, int(nm[i, k]), tuple(int(v) for v in sh[i, k])))
    return sorted(out)


# ------------------------------------------------------------------------------------------------------------ test_neighborlist.py
@pytest.mark.parametrize("dtype", DTYPES)
def test_auto_selection_and_tuple_layouts(dtype):
    """TestNeighborListAutoSelection (test_neighborlist.py:43-292): which method the dispatcher picks is visible in the tuple it returns."""
    from nvalchemiops.neighborlist import neighbor_list

    g = torch.Generator().manual_seed(0)
    # < 5000 atoms, no cell: naive, free space -> (list, ptr) without shifts (:43)
    pos = (torch.rand((100, 3), generator=g, dtype=dtype) * (100 / 0.25) ** (1 / 3)).to(DEV)
    out = neighbor_list(pos, 2.0, return_neighbor_list=True)
    assert len(out) == 2 and out[0].shape[0] == 2 and out[1].shape[0] == 101 and int(out[1][0]) == 0
    # < 5000 atoms with cell and pbc: naive with shifts (:68)
    p, c, b = _random_system(100, 10.0, dtype)
    out = neighbor_list(p, 2.0, cell=c, pbc=b, return_neighbor_list=True)
    assert len(out) == 3 and out[0].shape[0] == 2 and out[1].shape[0] == 101 and out[2].shape[1] == 3
    assert _pairs(out[0], out[2]) == _oracle_pairs(p, c, b, 2.0)
    # >= 5000 atoms, nothing else given: cell list over an automatic cell -> shifts are part of the result (:93)
    big = (torch.randn((5000, 3), generator=g, dtype=dtype) * 50.0).to(DEV)
    out = neighbor_list(big, 2.0, return_neighbor_list=True)
    assert len(out) == 3 and out[0].shape[0] == 2 and out[1].shape[0] == 5001 and int(out[1][0]) == 0 and out[2].shape[1] == 3
    assert int(out[2].abs().max()) == 0 if out[2].numel() else True
    # cutoff2 -> naive_dual_cutoff: six outputs (:118)
    out = neighbor_list(p, 2.5, cell=c, pbc=b, cutoff2=3.5, max_neighbors1=50, max_neighbors2=50, return_neighbor_list=True)
    assert len(out) == 6
    assert out[0].shape[0] == 2 and out[3].shape[0] == 2 and out[1].shape[0] == 101 and out[4].shape[0] == 101
    assert out[2].shape[1] == 3 and out[5].shape[1] == 3
    assert _pairs(out[0], out[2]) == _oracle_pairs(p, c, b, 2.5) and _pairs(out[3], out[5]) == _oracle_pairs(p, c, b, 3.5)
    # batch_idx -> batch_naive (:155), batch + cutoff2 -> batch_naive_dual_cutoff (:242)
    P, Cc, B, bi, bp = _two_system_batch(dtype)
    out = neighbor_list(P, 2.0, cell=Cc, pbc=B, batch_idx=bi, batch_ptr=bp, return_neighbor_list=True)
    assert len(out) == 3 and out[0].shape[0] == 2 and out[1].shape[0] == 81 and int(out[1][0]) == 0
    out = neighbor_list(P, 2.5, cell=Cc, pbc=B, batch_idx=bi, batch_ptr=bp, cutoff2=3.5, max_neighbors1=50, max_neighbors2=50,
                        return_neighbor_list=True)
    assert len(out) == 6 and out[1].shape[0] == 81 and out[4].shape[0] == 81 and out[2].shape[1] == 3 and out[5].shape[1] == 3
    # batch with >= 5000 atoms in total -> batch_cell_list (:199)
    bigp = (torch.randn((5500, 3), generator=g, dtype=dtype) * 50.0).to(DEV)
    cells = (torch.eye(3, dtype=dtype) * 60.0).repeat(2, 1, 1).to(DEV)
    bb = torch.ones((2, 3), dtype=torch.bool, device=DEV)
    bi2 = torch.cat([torch.zeros(3000, dtype=torch.int32), torch.ones(2500, dtype=torch.int32)]).to(DEV)
    out = neighbor_list(bigp, 2.0, cell=cells, pbc=bb, batch_idx=bi2, batch_ptr=torch.tensor([0, 3000, 5500], dtype=torch.int32, device=DEV),
                        return_neighbor_list=True)
    assert len(out) == 3 and out[0].shape[0] == 2 and out[1].shape[0] == 5501 and int(out[1][0]) == 0
    # no pair crosses the two systems
    src, dst = out[0][0].long(), out[0][1].long()
    assert bool((bi2[src] == bi2[dst]).all())


@pytest.mark.parametrize("dtype", DTYPES)
def test_explicit_methods_formats_half_fill(dtype):
    """TestNeighborListExplicitMethod / BatchProcessing / ReturnFormats / HalfFill / NoPBC (test_neighborlist.py:294-717)."""
    from nvalchemiops.neighborlist import neighbor_list

    p, c, b = _random_system(100, 10.0, dtype)
    want = _oracle_pairs(p, c, b, 3.0)
    for method in ("naive", "cell_list"):
        nl, ptr, sh = neighbor_list(p, 3.0, cell=c, pbc=b, method=method, return_neighbor_list=True)
        assert nl.shape[0] == 2 and ptr.shape[0] == 101 and nl.dtype == torch.int32 and sh.dtype == torch.int32
        assert _pairs(nl, sh) == want
        nm, num, nsh = neighbor_list(p, 3.0, cell=c, pbc=b, method=method, return_neighbor_list=False)
        assert nm.dim() == 2 and nm.shape[0] == 100 and num.shape == (100,) and nsh.shape == nm.shape + (3,)
        assert int(num.sum()) == len(want)
    # half_fill: each unordered pair once (:656)
    full = neighbor_list(p, 3.0, cell=c, pbc=b, method="cell_list", return_neighbor_list=True)[0]
    half = neighbor_list(p, 3.0, cell=c, pbc=b, method="cell_list", half_fill=True, return_neighbor_list=True)[0]
    assert full.shape[1] == 2 * half.shape[1]
    # no pbc, naive (:698): two outputs in either format's free-space variant
    out = neighbor_list(p, 3.0, method="naive", return_neighbor_list=False)
    assert len(out) == 2 and out[0].shape[0] == 100
    # batch methods, explicit (:365, :420)
    P, Cc, B, bi, bp = _two_system_batch(dtype)
    for method in ("batch_naive", "batch_cell_list"):
        nl, ptr, sh = neighbor_list(P, 3.0, cell=Cc, pbc=B, batch_idx=bi, batch_ptr=bp, method=method, return_neighbor_list=True)
        assert ptr.shape[0] == 81
        first = _oracle_pairs(P[:50], Cc[0], B[0], 3.0)
        second = [(i + 50, j + 50, s) for i, j, s in _oracle_pairs(P[50:], Cc[1], B[1], 3.0)]
        assert _pairs(nl, sh) == sorted(first + second)
    with pytest.raises(ValueError):
        neighbor_list(p, 3.0, method="invalid_method")  # :722


def test_kwargs_forwarding_and_edge_cases():
    """TestNeighborListKwargs / EdgeCases / PrepareBatchIdxPtr (test_neighborlist.py:731-985)."""
    from nvalchemiops.neighborlist import neighbor_list
    from nvalchemiops.neighborlist.neighbor_utils import _prepare_batch_idx_ptr

    p, c, b = _random_system(50, 10.0, torch.float32)
    # 50 atoms, cutoff 5 in a box of 10 has rows longer than 20: the matrix keeps the requested width, the counts keep counting (:735)
    nm20, num20, _ = neighbor_list(p, 5.0, cell=c, pbc=b, method="naive", max_neighbors=20)
    assert nm20.shape[1] == 20 and int(num20.sum()) == len(_oracle_pairs(p, c, b, 5.0))
    p100, c15, _ = _random_system(100, 15.0, torch.float32)
    assert neighbor_list(p100, 2.0, cell=c15, pbc=b, method="cell_list", max_neighbors=30)[0].shape[1] == 30
    out = neighbor_list(p, 2.5, cell=c, pbc=b, cutoff2=3.5, method="naive_dual_cutoff", max_neighbors1=15, max_neighbors2=25)
    assert out[0].shape[1] == 15 and out[3].shape[1] == 25
    assert neighbor_list(p, 2.0, cell=c, pbc=b, max_neighbors=25)[0].shape[1] == 25  # auto selection keeps the kwarg (:876)
    # pre-allocated outputs come back as the SAME objects (:818)
    nm = torch.full((50, 40), 50, dtype=torch.int32, device=DEV)
    num = torch.zeros(50, dtype=torch.int32, device=DEV)
    nsh = torch.zeros((50, 40, 3), dtype=torch.int32, device=DEV)
    r = neighbor_list(p, 3.0, cell=c, pbc=b, method="naive", neighbor_matrix=nm, num_neighbors=num, neighbor_matrix_shifts=nsh)
    assert r[0] is nm and r[1] is num and r[2] is nsh and int(num.sum()) == len(_oracle_pairs(p, c, b, 3.0))
    with pytest.raises(TypeError):  # :858
        neighbor_list(torch.randn((50, 3), device=DEV), 2.0, method="naive", invalid_parameter_name=123)
    # empty and single-atom inputs (:905, :925)
    out = neighbor_list(torch.empty((0, 3), dtype=torch.float32, device=DEV), 2.0, method="naive", return_neighbor_list=True)
    assert len(out) == 2 and out[0].shape[1] == 0 and out[1].shape[0] == 1 and int(out[1][0]) == 0
    out = neighbor_list(torch.randn((1, 3), dtype=torch.float32, device=DEV), 2.0, method="naive", return_neighbor_list=True)
    assert len(out) == 2 and out[0].shape[1] == 0 and out[1].shape[0] == 2 and int(out[1][0]) == 0
    # _prepare_batch_idx_ptr (:955): either input reconstructs the other; neither -> ValueError with the reference's message
    bi = torch.tensor([0, 0, 1, 1, 1, 2, 2], dtype=torch.int32, device=DEV)
    bp = torch.tensor([0, 2, 5, 7], dtype=torch.int32, device=DEV)
    with pytest.raises(ValueError, match="Either batch_idx or batch_ptr must be provided."):
        _prepare_batch_idx_ptr(None, None, 7, torch.device(DEV))
    for a, bb in ((bi, None), (None, bp), (bi, bp)):
        i2, p2 = _prepare_batch_idx_ptr(a, bb, 7, torch.device(DEV))
        assert torch.equal(i2.cpu().int(), bi.cpu()) and torch.equal(p2.cpu().int(), bp.cpu())


# ------------------------------------------------------------------------------------------------------------ test_cell_list.py
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("as_list", [True, False])
def test_cell_list_small_systems_and_cutoff_extremes(dtype, as_list):
    """TestCellListAPI (test_cell_list.py:62-330): one atom, two atoms, simple cubic, cutoff larger than the box, zero cutoff."""
    from nvalchemiops.neighborlist import cell_list

    pbc = torch.tensor([True, True, True], device=DEV)
    cell = (torch.eye(3, dtype=dtype) * 2.0).reshape(1, 3, 3).to(DEV)
    one = torch.tensor([[0.5, 0.5, 0.5]], dtype=dtype, device=DEV)
    out = cell_list(one, 1.0, cell, pbc, return_neighbor_list=as_list)
    assert (out[0].shape[1] if as_list else int(out[1].sum())) == 0                       # :62 no self pair at cutoff < box
    two = torch.tensor([[0.0, 0.0, 0.0], [0.5, 0.0, 0.0]], dtype=dtype, device=DEV)
    out = cell_list(two, 1.0, cell, pbc, return_neighbor_list=as_list)
    assert (out[0].shape[1] if as_list else int(out[1].sum())) == 2                       # :83 the pair, both directions
    # simple cubic 2x2x2, spacing 1, box 2: 6 neighbours at distance 1 through the boundary (:106)
    g = torch.arange(2, dtype=dtype)
    sc = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).to(DEV)
    out = cell_list(sc, 1.1, cell, pbc, return_neighbor_list=as_list)
    assert (out[0].shape[1] if as_list else int(out[1].sum())) == 8 * 6
    # cutoff 5 in a box of 2: images several cells away, self images included (:269); compare with the oracle
    p, c, b = _random_system(10, 2.0, dtype, seed=123)
    want = _oracle_pairs(p, c, b, 5.0)
    if as_list:
        nl, ptr, sh = cell_list(p, 5.0, c, b, return_neighbor_list=True)
        assert _pairs(nl, sh) == want
    else:
        nm, num, sh = cell_list(p, 5.0, c, b, max_neighbors=max(len(want) // 10 + 64, 64))
        assert int(num.sum()) == len(want)
    # zero cutoff: empty results with the documented shapes (:292)
    out = cell_list(sc, 0.0, cell, pbc, return_neighbor_list=as_list)
    assert len(out) == 3
    if as_list:
        assert out[0].shape == (2, 0) and out[1].shape == (9,) and out[2].shape == (0, 3)
    else:
        assert out[0].shape[0] == 8 and int(out[1].sum()) == 0


def test_cell_list_edge_cases_and_dtypes():
    """TestEdgeCases (test_cell_list.py:427-505): estimates for an empty / negative-cutoff input, empty system, output dtypes, devices."""
    from nvalchemiops.neighborlist import cell_list, estimate_cell_list_sizes

    for dtype in DTYPES:
        mc, rad = estimate_cell_list_sizes(torch.zeros((0, 3, 3), dtype=dtype, device=DEV), torch.zeros((0, 3), dtype=torch.bool, device=DEV), 1.0)
        assert mc == 1 and rad.shape == (3,) and rad.dtype == torch.int32 and rad.device == torch.device(DEV)
        mc, rad = estimate_cell_list_sizes(torch.eye(3, dtype=dtype, device=DEV).reshape(1, 3, 3), torch.ones((1, 3), dtype=torch.bool, device=DEV), -1.0)
        assert mc == 1 and rad.shape == (3,) and rad.dtype == torch.int32
    pbc = torch.tensor([True, True, True], device=DEV)
    for as_list in (True, False):
        out = cell_list(torch.empty((0, 3), dtype=torch.float32, device=DEV), 1.0, torch.eye(3, dtype=torch.float32, device=DEV), pbc,
                        return_neighbor_list=as_list)
        assert len(out) == 3
        if as_list:
            assert out[0].shape == (2, 0) and out[1].shape == (1,) and out[2].shape == (0, 3)
        else:
            assert out[0].shape[0] == 0 and out[1].shape == (0,) and out[2].shape[0] == 0 and out[2].shape[2] == 3
    for dtype in DTYPES:
        p, c, b = _random_system(20, 4.0, dtype)
        for as_list in (True, False):
            out = cell_list(p, 1.5, c, b, return_neighbor_list=as_list)
            assert all(t.dtype == torch.int32 and t.device == p.device for t in out)      # :474, :490


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("pbc_flags", [[True, True, True], [False, False, False], [True, False, True], [False, False, True]])
@pytest.mark.parametrize("shape", ["random", "nonorthorhombic"])
def test_scaling_correctness_vs_oracle(dtype, pbc_flags, shape):
    """test_scaling_correctness (test_cell_list.py:333-390; there against vesin): 10-100 atoms x cutoffs 1 / 3 / 5 x pbc patterns x
    cubic and sheared cells, pair multiset == the oracle's."""
    from nvalchemiops.neighborlist import cell_list

    for n in (10, 50, 100):
        g = torch.Generator().manual_seed(n)
        if shape == "random":
            cell = torch.eye(3, dtype=dtype) * 6.0
        else:
            cell = torch.tensor([[6.0, 0.0, 0.0], [1.5, 5.5, 0.0], [0.8, 1.1, 5.0]], dtype=dtype)
        pos = (torch.rand((n, 3), generator=g, dtype=dtype) @ cell).to(DEV)
        cell = cell.reshape(1, 3, 3).to(DEV)
        pbc = torch.tensor(pbc_flags, device=DEV)
        for cutoff in (1.0, 3.0, 5.0):
            nl, ptr, sh = cell_list(pos, cutoff, cell, pbc, return_neighbor_list=True)
            assert _pairs(nl, sh) == _oracle_pairs(pos, cell, pbc, cutoff), (n, cutoff)


# ------------------------------------------------------------------------------------------------------ test_batch_cell_list.py
@pytest.mark.parametrize("dtype", DTYPES)
def test_batch_cell_list_scenarios(dtype):
    """TestBatchCellListAPI / TestBatchEdgeCases (test_batch_cell_list.py:70-740): identical systems give identical rows, different
    systems do not mix, non-periodic and mixed-pbc batches, zero cutoff, empty batch."""
    from nvalchemiops.neighborlist import batch_cell_list, cell_list, estimate_batch_cell_list_sizes

    p, c, b = _random_system(40, 6.0, dtype, seed=5)
    P = torch.cat([p, p])
    Cc = torch.cat([c, c])
    B = torch.stack([b, b])
    bi = torch.cat([torch.zeros(40, dtype=torch.int32), torch.ones(40, dtype=torch.int32)]).to(DEV)
    nm, num, sh = batch_cell_list(P, 2.5, Cc, B, bi, max_neighbors=64)
    assert torch.equal(num[:40], num[40:])                                                # :123 same structure twice
    nl, ptr, lsh = batch_cell_list(P, 2.5, Cc, B, bi, return_neighbor_list=True)
    one = _oracle_pairs(p, c, b, 2.5)
    assert _pairs(nl, lsh) == sorted(one + [(i + 40, j + 40, s) for i, j, s in one])
    # two different structures, one of them non-periodic along y (:164, :318)
    p2, c2, _ = _random_system(25, 9.0, dtype, seed=6)
    b2 = torch.tensor([True, False, True], device=DEV)
    P2, C2, B2 = torch.cat([p, p2]), torch.cat([c, c2]), torch.stack([b, b2])
    bi2 = torch.cat([torch.zeros(40, dtype=torch.int32), torch.ones(25, dtype=torch.int32)]).to(DEV)
    nl, ptr, lsh = batch_cell_list(P2, 3.0, C2, B2, bi2, return_neighbor_list=True)
    second = [(i + 40, j + 40, s) for i, j, s in _oracle_pairs(p2, c2, b2, 3.0)]
    assert _pairs(nl, lsh) == sorted(_oracle_pairs(p, c, b, 3.0) + second)
    assert int(lsh[ptr[40].item():, 1].abs().max()) == 0                                   # no image along the open axis
    # fully open batch (:279)
    Bo = torch.zeros((2, 3), dtype=torch.bool, device=DEV)
    nl, ptr, lsh = batch_cell_list(P2, 3.0, C2, Bo, bi2, return_neighbor_list=True)
    assert int(lsh.abs().max()) == 0 if lsh.numel() else True
    # zero cutoff (:411)
    nm, num, sh = batch_cell_list(P2, 0.0, C2, B2, bi2)
    assert nm.shape[0] == 65 and int(num.sum()) == 0
    nl, ptr, lsh = batch_cell_list(P2, 0.0, C2, B2, bi2, return_neighbor_list=True)
    assert nl.shape == (2, 0) and ptr.shape == (66,) and lsh.shape == (0, 3)
    # empty batch (:550, :661)
    mc, rad = estimate_batch_cell_list_sizes(torch.zeros((0, 3, 3), dtype=dtype, device=DEV), torch.zeros((0, 3), dtype=torch.bool, device=DEV), 1.0)
    assert mc == 1 and rad.dtype == torch.int32
    out = batch_cell_list(torch.empty((0, 3), dtype=dtype, device=DEV), 1.0, torch.zeros((0, 3, 3), dtype=dtype, device=DEV),
                          torch.zeros((0, 3), dtype=torch.bool, device=DEV), torch.empty(0, dtype=torch.int32, device=DEV))
    assert len(out) == 3 and out[0].shape[0] == 0 and out[1].shape == (0,)
    # a single system through the batch entry equals the single-system entry (:70, :96)
    a = batch_cell_list(p, 2.5, c, b.reshape(1, 3), torch.zeros(40, dtype=torch.int32, device=DEV), max_neighbors=64)
    s = cell_list(p, 2.5, c, b, max_neighbors=64)
    assert all(torch.equal(x, y) for x, y in zip(a, s))


# ------------------------------------------------------------------------------------------------------ test_rebuild_detection.py
@pytest.mark.parametrize("dtype", DTYPES)
def test_rebuild_detection_scenarios(dtype):
    """TestRebuildDetection (test_rebuild_detection.py:79-440): no / small / large movement for both detectors, shape mismatch, empty
    input, the two convenience wrappers, mixed precision of reference and current positions."""
    from nvalchemiops.neighborlist import (allocate_cell_list, build_cell_list, cell_list_needs_rebuild, check_cell_list_rebuild_needed,
                                           check_neighbor_list_rebuild_needed, estimate_cell_list_sizes, neighbor_list_needs_rebuild)

    g = torch.Generator().manual_seed(3)
    cell = (torch.eye(3, dtype=dtype) * 10.0).reshape(1, 3, 3).to(DEV)
    pbc = torch.tensor([True, True, True], device=DEV)
    # atoms at cell centres of a 2.5 A grid, so that "small" and "large" moves are unambiguous
    base = (torch.randint(0, 4, (60, 3), generator=g).to(dtype) + 0.5) * 2.5
    pos = base.to(DEV)
    cutoff = 2.5
    mc, rad = estimate_cell_list_sizes(cell, pbc.reshape(1, 3), cutoff)
    cache = allocate_cell_list(60, mc, rad, torch.device(DEV))
    build_cell_list(pos, cutoff, cell, pbc, *cache)
    cells_per_dimension, atom_to_cell_mapping = cache[0], cache[3]
    same = cell_list_needs_rebuild(pos, atom_to_cell_mapping, cells_per_dimension, cell, pbc)
    assert same.dtype == torch.bool and not bool(same)                                    # :79
    assert not bool(cell_list_needs_rebuild(pos + 0.3, atom_to_cell_mapping, cells_per_dimension, cell, pbc))   # :113 stays inside its cell
    moved = pos.clone()
    moved[7, 0] += 2.6
    assert bool(cell_list_needs_rebuild(moved, atom_to_cell_mapping, cells_per_dimension, cell, pbc))           # :150 crosses a face
    assert not bool(cell_list_needs_rebuild(pos[:0], atom_to_cell_mapping[:0], cells_per_dimension, cell, pbc))  # :188 empty
    # skin criterion: displacement > skin / 2 (:212-276)
    skin = 1.0
    assert not bool(neighbor_list_needs_rebuild(pos, pos.clone(), skin / 2))
    assert not bool(neighbor_list_needs_rebuild(pos, pos + 0.2, skin / 2))
    far = pos.clone()
    far[11] += torch.tensor([0.4, 0.3, 0.2], dtype=dtype, device=DEV)                     # |d| = 0.539 > 0.5
    assert bool(neighbor_list_needs_rebuild(pos, far, skin / 2))
    assert bool(neighbor_list_needs_rebuild(pos, pos[:30], skin / 2))                     # :278 shape mismatch -> rebuild
    assert not bool(neighbor_list_needs_rebuild(pos[:0], pos[:0], skin / 2))              # :295
    # wrappers (:312, :368)
    assert check_cell_list_rebuild_needed(*cache, pos, cell, pbc, cutoff) is False
    assert check_cell_list_rebuild_needed(*cache, moved, cell, pbc, cutoff) is True
    assert check_neighbor_list_rebuild_needed(pos, pos + 0.1, skin / 2) is False
    assert check_neighbor_list_rebuild_needed(pos, far, skin / 2) is True
    # many atoms moved: still one boolean (:398 early termination is an implementation detail of the reference kernel)
    assert bool(neighbor_list_needs_rebuild(pos, pos + 5.0, skin / 2))
    # reference positions kept in the other precision (:424)
    other = torch.float64 if dtype == torch.float32 else torch.float32
    assert not bool(neighbor_list_needs_rebuild(pos.to(other), pos.to(other) + 0.1, skin / 2))


# ------------------------------------------------------------------------------------------------------------------ test_dftd3.py
D3 = dict(a1=0.3981, a2=4.4211, s8=1.9889)


def _d3_tables():
    from tests.test_d3_gpu import _params

    return _params()


def _d3_check(out, ref, virial=False):
    from tests.test_d3_gpu import _check

    _check(out, ref, virial)


def _d3_oracle(pos, z, t, **kw):
    from tests.test_d3_gpu import _wide

    return _wide(pos, z, t, k1=16.0, k3=-4.0, s6=1.0, **kw)


def test_dftd3_custom_op_branches_and_parameter_supply():
    """TestCustomOpBranches (test_dftd3.py:496-630) and TestParameterSupply (:1772-2148): float64 positions, S5 window, explicit
    fill_value, `device=` given or not; explicit tensors, D3Parameters, dict, dict / dataclass with single-tensor overrides, nothing."""
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3

    t, p = _d3_tables()
    pos = np.array([[0.0, 0.0, 0.0], [1.4, 0.0, 0.0]], np.float32)
    z = np.array([1, 1], np.int32)
    nm = np.array([[1, 2], [0, 2]], np.int32)
    tp, tz, tnm = torch.as_tensor(pos, device=DEV), torch.as_tensor(z, device=DEV), torch.as_tensor(nm, device=DEV)
    ref = _d3_oracle(pos, z, t, neighbor_matrix=nm, **D3)
    explicit = dict(covalent_radii=p.rcov, r4r2=p.r4r2, c6_reference=p.c6ab, coord_num_ref=p.cn_ref)
    base = dftd3(tp, tz, neighbor_matrix=tnm, **explicit, **D3)
    _d3_check(base, ref)
    assert base[0].shape == (1,) and base[1].shape == (2, 3) and base[2].shape == (2,)                     # :1205
    assert all(o.dtype == torch.float32 for o in base)
    # float64 positions: results stay float32 (dftd3.py:1912), numbers equal to the float32 run at this geometry
    out64 = dftd3(tp.double(), tz, neighbor_matrix=tnm, **explicit, **D3)
    assert all(o.dtype == torch.float32 for o in out64)
    _d3_check(out64, _d3_oracle(pos.astype(np.float64), z, t, neighbor_matrix=nm, **D3))
    # explicit fill_value, device argument present / absent
    for extra in (dict(fill_value=2), dict(device=DEV), dict()):
        _d3_check(dftd3(tp, tz, neighbor_matrix=tnm, **explicit, **D3, **extra), ref)
    # S5 window around the bond length (s5_off > s5_on enables it)
    sm = dftd3(tp, tz, neighbor_matrix=tnm, **explicit, **D3, s5_smoothing_on=5.0, s5_smoothing_off=10.0)
    _d3_check(sm, _d3_oracle(pos, z, t, neighbor_matrix=nm, s5_on=5.0, s5_off=10.0, **D3))
    # parameter supply: all of these are the same calculation
    as_dict = dict(rcov=p.rcov, r4r2=p.r4r2, c6ab=p.c6ab, cn_ref=p.cn_ref)
    for kw in (dict(d3_params=p), dict(d3_params=as_dict), dict(d3_params=as_dict, r4r2=p.r4r2.clone()),
               dict(d3_params=p, covalent_radii=p.rcov.clone()), dict(d3_params=p, c6_reference=p.c6ab.clone(), coord_num_ref=p.cn_ref.clone())):
        got = dftd3(tp, tz, neighbor_matrix=tnm, **kw, **D3)
        assert all(torch.equal(a, b) for a, b in zip(got, base))
    # an override really overrides (:1908, :1984): doubled r4r2 changes the energy, as in the oracle
    t2 = dict(t, r4r2=t["r4r2"] * 2.0)
    got = dftd3(tp, tz, neighbor_matrix=tnm, d3_params=as_dict, r4r2=p.r4r2 * 2.0, **D3)
    _d3_check(got, _d3_oracle(pos, z, t2, neighbor_matrix=nm, **D3))
    assert abs(float(got[0]) - float(base[0])) > 1e-6 * abs(float(base[0]))
    with pytest.raises(RuntimeError, match="DFT-D3 parameters must be explicitly provided"):              # :1961
        dftd3(tp, tz, neighbor_matrix=tnm, **D3)
    # validation messages of the neighbour formats (:2843)
    nl = torch.tensor([[0, 1], [1, 0]], dtype=torch.int32, device=DEV)
    ptr = torch.tensor([0, 1, 2], dtype=torch.int32, device=DEV)
    for kw, msg in ((dict(neighbor_matrix=tnm, neighbor_list=nl), "Cannot provide both neighbor_matrix and neighbor_list"),
                    (dict(), "Must provide either neighbor_matrix or neighbor_list"),
                    (dict(neighbor_matrix=tnm, unit_shifts=torch.zeros((2, 3), dtype=torch.int32, device=DEV)), "unit_shifts is for neighbor_list format"),
                    (dict(neighbor_list=nl, neighbor_matrix_shifts=torch.zeros((2, 2, 3), dtype=torch.int32, device=DEV)),
                     "neighbor_matrix_shifts is for neighbor_matrix format"),
                    (dict(neighbor_list=nl), "neighbor_ptr must be provided when using neighbor_list")):
        with pytest.raises(ValueError, match=msg):
            dftd3(tp, tz, d3_params=p, **kw, **D3)
    # list format == matrix format (:2753)
    lst = dftd3(tp, tz, d3_params=p, neighbor_list=nl, neighbor_ptr=ptr, **D3)
    _d3_check(lst, ref)
    # D3Parameters.to (:356-400)
    p64 = p.to(dtype=torch.float64)
    assert p64.rcov.dtype == torch.float64 and p64.c6ab.dtype == torch.float64 and p64.device == p.device
    assert isinstance(p.to(device="cpu"), D3Parameters) and p.to(device="cpu").device == torch.device("cpu")
    _d3_check(dftd3(tp, tz, d3_params=p64, neighbor_matrix=tnm, **D3), ref)                                 # :400 float64 tables


def test_dftd3_batching_scenarios():
    """TestBatchIndexHandling / TestBatching (test_dftd3.py:2150-2840): batch_idx alone fixes the system count, identical systems give
    identical results, batch == individual runs for systems of different sizes, per-system energies accumulate per system, a large batch."""
    from nvalchemiops.interactions.dispersion import dftd3

    t, p = _d3_tables()
    pos = np.array([[0, 0, 0], [1.4, 0, 0], [5.0, 0, 0], [6.4, 0, 0]], np.float32)
    z = np.array([1, 1, 1, 1], np.int32)
    nm = np.full((4, 5), 4, np.int32)
    nm[0, 0], nm[1, 0], nm[2, 0], nm[3, 0] = 1, 0, 3, 2
    bi = np.array([0, 0, 1, 1], np.int32)
    dev = lambda a: torch.as_tensor(a, device=DEV)  # noqa: E731
    out = dftd3(dev(pos), dev(z), d3_params=p, neighbor_matrix=dev(nm), batch_idx=dev(bi), **D3)
    assert out[0].shape == (2,) and out[1].shape == (4, 3) and out[2].shape == (4,)                        # :2153
    _d3_check(out, _d3_oracle(pos, z, t, neighbor_matrix=nm, batch_idx=bi, **D3))
    assert torch.allclose(out[0][0], out[0][1], rtol=1e-6) and torch.allclose(out[1][:2], out[1][2:], rtol=1e-5, atol=1e-9)   # :2221
    # systems of different sizes: the batch equals the runs of its members (:2293, :2393)
    g = np.random.default_rng(12)
    sizes = (3, 17, 64, 1, 30)
    parts, zs, mats = [], [], []
    off = 0
    total = sum(sizes)
    width = 70
    for n in sizes:
        pp = (g.random((n, 3)) * (2.0 + n ** (1 / 3) * 2.5)).astype(np.float32)
        parts.append(pp), zs.append(g.choice(np.array([1, 6, 7, 8], np.int32), n))
        m = np.full((n, width), total, np.int32)
        for i in range(n):
            others = [j for j in range(n) if j != i]
            m[i, :len(others)] = np.array(others, np.int32) + off
        mats.append(m)
        off += n
    P, Z, M = np.concatenate(parts), np.concatenate(zs), np.concatenate(mats)
    BI = np.concatenate([np.full(n, s, np.int32) for s, n in enumerate(sizes)])
    batch = dftd3(dev(P), dev(Z), d3_params=p, neighbor_matrix=dev(M), batch_idx=dev(BI), fill_value=total, **D3)
    assert batch[0].shape == (len(sizes),)
    _d3_check(batch, _d3_oracle(P, Z, t, neighbor_matrix=M, batch_idx=BI, fill_value=total, **D3))
    off = 0
    for s, n in enumerate(sizes):
        local = np.where(mats[s] == total, n, mats[s] - off).astype(np.int32)
        one = dftd3(dev(parts[s]), dev(zs[s]), d3_params=p, neighbor_matrix=dev(local), **D3)
        assert torch.allclose(one[0][0], batch[0][s], rtol=2e-6, atol=1e-9)                                 # :2545 accumulation per system
        assert torch.allclose(one[1], batch[1][off:off + n], rtol=1e-5, atol=1e-8) and torch.allclose(one[2], batch[2][off:off + n], rtol=1e-6)
        off += n
    # many small systems (:2676): 200 dimers
    nsys = 200
    P = np.zeros((2 * nsys, 3), np.float32)
    P[1::2, 0] = 1.2 + 0.01 * np.arange(nsys)
    P[:, 1] = np.repeat(np.arange(nsys), 2) * 30.0
    M = np.full((2 * nsys, 4), 2 * nsys, np.int32)
    M[0::2, 0] = np.arange(1, 2 * nsys, 2)
    M[1::2, 0] = np.arange(0, 2 * nsys, 2)
    BI = np.repeat(np.arange(nsys, dtype=np.int32), 2)
    Z = np.tile(np.array([6, 8], np.int32), nsys)
    out = dftd3(dev(P), dev(Z), d3_params=p, neighbor_matrix=dev(M), batch_idx=dev(BI), **D3)
    assert out[0].shape == (nsys,)
    _d3_check(out, _d3_oracle(P, Z, t, neighbor_matrix=M, batch_idx=BI, **D3))


@pytest.mark.parametrize("fmt", ["matrix", "list"])
def test_dftd3_periodic_and_virial_scenarios(fmt):
    """TestPBC (test_dftd3.py:768-1200) and the periodic batch cases (:2466, :2976): virial needs cell + shifts (messages), virial for
    matrix / list / batch / float64 input, periodic batch == members."""
    from nvalchemiops.interactions.dispersion import dftd3
    from nvalchemiops.neighborlist import batch_cell_list

    t, p = _d3_tables()
    g = np.random.default_rng(4)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=DEV)  # noqa: E731
    sizes, boxes = (24, 40), (9.0, 11.0)
    parts = []
    for n, b in zip(sizes, boxes):  # jittered lattice: no unphysical close contacts (their fp32 pair terms would dominate the force error)
        k = int(np.ceil(n ** (1 / 3)))
        grid = np.stack(np.meshgrid(*[np.arange(k)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
        parts.append(((grid + 0.5 + (g.random((n, 3)) - 0.5) * 0.4) * (b / k)).astype(np.float32))
    cells = np.stack([np.eye(3, dtype=np.float32) * b for b in boxes])
    cells[1, 1, 0] = 1.3
    P = np.concatenate(parts)
    Z = g.choice(np.array([1, 6, 8, 14], np.int32), P.shape[0])
    BI = np.concatenate([np.full(n, s, np.int32) for s, n in enumerate(sizes)])
    pbc = torch.ones((2, 3), dtype=torch.bool, device=DEV)
    if fmt == "matrix":
        nm, num, sh = batch_cell_list(dev(P), 7.0, dev(cells), pbc, dev(BI), max_neighbors=160)
        assert int(num.max()) <= 160
        kw = dict(neighbor_matrix=nm, neighbor_matrix_shifts=sh)
        okw = dict(neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy())
    else:
        nl, ptr, sh = batch_cell_list(dev(P), 7.0, dev(cells), pbc, dev(BI), return_neighbor_list=True)
        kw = dict(neighbor_list=nl, neighbor_ptr=ptr, unit_shifts=sh)
        okw = dict(idx_j=nl[1].cpu().numpy(), neighbor_ptr=ptr.cpu().numpy(), unit_shifts=sh.cpu().numpy())
    out = dftd3(dev(P), dev(Z), d3_params=p, cell=dev(cells), batch_idx=dev(BI), compute_virial=True, **kw, **D3)
    assert len(out) == 4 and out[3].shape == (2, 3, 3) and out[3].dtype == torch.float32
    _d3_check(out, _d3_oracle(P, Z, t, cell=cells, batch_idx=BI, compute_virial=True, **okw, **D3), virial=True)
    assert torch.allclose(out[3], out[3].transpose(1, 2), rtol=1e-4, atol=1e-7)                             # a symmetric tensor
    # float64 positions and cell (:1149): float32 results
    out64 = dftd3(dev(P).double(), dev(Z), d3_params=p, cell=dev(cells).double(), batch_idx=dev(BI), compute_virial=True, **kw, **D3)
    assert all(o.dtype == torch.float32 for o in out64)
    _d3_check(out64, _d3_oracle(P.astype(np.float64), Z, t, cell=cells.astype(np.float64), batch_idx=BI, compute_virial=True, **okw, **D3), virial=True)
    # without virial the first three outputs are unchanged
    plain = dftd3(dev(P), dev(Z), d3_params=p, cell=dev(cells), batch_idx=dev(BI), **kw, **D3)
    assert len(plain) == 3 and all(torch.equal(a, b) for a, b in zip(plain, out[:3]))
    # virial preconditions (:813): no cell, or a cell without shifts
    no_shift = {k: v for k, v in kw.items() if k not in ("neighbor_matrix_shifts", "unit_shifts")}
    with pytest.raises(ValueError, match="Virial computation requires periodic boundary conditions"):
        dftd3(dev(P), dev(Z), d3_params=p, batch_idx=dev(BI), compute_virial=True, **no_shift, **D3)
    with pytest.raises(ValueError, match="Virial computation requires periodic boundary conditions"):
        dftd3(dev(P), dev(Z), d3_params=p, cell=dev(cells), batch_idx=dev(BI), compute_virial=True, **no_shift, **D3)


# -------------------------------------------------------------------------------------------------------------------- test_pme.py
def _simple_system(n=5, dtype=torch.float64, box=10.0, seed=0):
    """The reference's `create_simple_system` shape: random positions in the middle 80 % of a cubic cell, neutral random charges."""
    g = torch.Generator().manual_seed(seed)
    pos = torch.rand((n, 3), generator=g, dtype=dtype) * box * 0.8 + box * 0.1
    q = torch.randn(n, generator=g, dtype=dtype)
    q[-1] = -q[:-1].sum()
    return pos.to(DEV), q.to(DEV), (torch.eye(3, dtype=dtype) * box).to(DEV)


def _dipole(dtype=torch.float64, sep=2.0, box=10.0):
    c = box / 2
    pos = torch.tensor([[c - sep / 2, c, c], [c + sep / 2, c, c]], dtype=dtype, device=DEV)
    return pos, torch.tensor([1.0, -1.0], dtype=dtype, device=DEV), (torch.eye(3, dtype=dtype) * box).to(DEV)


def test_pme_reciprocal_api_scenarios():
    """TestDtypeSupport, TestPMEReciprocalSpaceAPI, TestPMEConservationLaws, TestPMEConvergence, TestSingleAtomSystem, TestNonCubicCells,
    TestPrecomputedKVectors, TestAlphaSensitivity, TestZeroCharges, TestPrepareAlphaPME, TestPMEMeshDimensionErrors
    (test_pme.py:139-535, :1628-2095)."""
    from nvalchemiops.interactions.electrostatics import generate_k_vectors_pme, pme_reciprocal_space

    kw = dict(alpha=0.3, mesh_dimensions=(16, 16, 16))
    for dtype in DTYPES:                                                                                   # :144, :275, :296
        pos, q, cell = _simple_system(4, dtype)
        e = pme_reciprocal_space(pos, q, cell, **kw)
        e2, f = pme_reciprocal_space(pos, q, cell, compute_forces=True, **kw)
        assert e.shape == (4,) and f.shape == (4, 3) and e.dtype == dtype and f.dtype == dtype and torch.equal(e, e2)
    p64, q64, c64 = _simple_system(6, torch.float64)
    e64, f64 = pme_reciprocal_space(p64, q64, c64, compute_forces=True, **kw)
    e32, f32 = pme_reciprocal_space(p64.float(), q64.float(), c64.float(), compute_forces=True, **kw)     # :223
    assert torch.allclose(e32.double(), e64, rtol=1e-3, atol=1e-5) and torch.allclose(f32.double(), f64, rtol=1e-3, atol=1e-5)
    want = O.pme_reciprocal_space(p64.cpu().numpy(), q64.cpu().numpy(), c64.cpu().numpy(), 0.3, (16, 16, 16), compute_forces=True)
    assert np.allclose(e64.cpu().numpy(), want[0], rtol=1e-9, atol=1e-12) and np.allclose(f64.cpu().numpy(), want[1], rtol=1e-9, atol=1e-12)
    # batch entry: shapes, one system through the batch == the single call (:180, :317, :2275)
    bi = torch.zeros(6, dtype=torch.int32, device=DEV)
    eb, fb = pme_reciprocal_space(p64, q64, c64.unsqueeze(0), batch_idx=bi, compute_forces=True, **kw)
    # (not bit-equal: the reference's single-system spread keeps every weight > 0, its batch spread only weights > 1e-8, spline.py:548 / :820)
    assert eb.shape == (6,) and fb.shape == (6, 3) and torch.allclose(eb, e64, rtol=1e-6, atol=1e-8) and torch.allclose(fb, f64, rtol=1e-6, atol=1e-8)
    # empty system (:350)
    e0, f0 = pme_reciprocal_space(p64[:0], q64[:0], c64, compute_forces=True, **kw)
    assert e0.shape == (0,) and f0.shape == (0, 3)
    # spline orders the reference supports (:374)
    for order in (2, 3, 4):
        assert torch.isfinite(pme_reciprocal_space(p64, q64, c64, spline_order=order, **kw)).all()
    # conservation (:409-495): net force, translation invariance, the symmetric dipole
    _, f = pme_reciprocal_space(p64, q64, c64, alpha=0.3, mesh_dimensions=(20, 20, 20), compute_forces=True)
    assert float(f.sum(0).abs().max()) < 1e-4
    dp, dq, dc = _dipole()
    e1 = pme_reciprocal_space(dp, dq, dc, **kw)
    e2 = pme_reciprocal_space(dp + torch.tensor([1.5, 0.5, -0.3], dtype=torch.float64, device=DEV), dq, dc, **kw)
    assert torch.allclose(e1.sum(), e2.sum(), rtol=1e-4)
    _, fd = pme_reciprocal_space(dp, dq, dc, compute_forces=True, **kw)
    assert torch.allclose(fd[0], -fd[1], rtol=1e-6, atol=1e-12)
    # mesh convergence (:499): 4 -> 8 -> 16 -> 64
    es = [float(pme_reciprocal_space(dp, dq, dc, alpha=0.3, mesh_dimensions=(m, m, m)).sum()) for m in (4, 8, 16, 64)]
    assert abs(es[2] - es[1]) < abs(es[1] - es[0]) and abs(es[3] - es[2]) < abs(es[2] - es[1])
    # one atom (:1632): self + background terms only, zero force
    one = torch.tensor([[5.0, 5.0, 5.0]], dtype=torch.float64, device=DEV)
    e, f = pme_reciprocal_space(one, torch.tensor([1.0], dtype=torch.float64, device=DEV), dc, compute_forces=True, **kw)
    assert e.shape == (1,) and torch.isfinite(e).all() and float(f.abs().max()) < 1e-10
    # orthorhombic and triclinic cells (:1665, :1702) against the oracle
    for cell in (torch.diag(torch.tensor([8.0, 10.0, 12.0], dtype=torch.float64)),
                 torch.tensor([[10.0, 0, 0], [2.0, 9.0, 0], [1.0, 1.5, 11.0]], dtype=torch.float64)):
        cell = cell.to(DEV)
        frac = torch.rand((8, 3), generator=torch.Generator().manual_seed(3), dtype=torch.float64).to(DEV)
        pos = frac @ cell
        q = torch.tensor([1.0, -1.0] * 4, dtype=torch.float64, device=DEV)
        e, f = pme_reciprocal_space(pos, q, cell, alpha=0.3, mesh_dimensions=(16, 20, 24), compute_forces=True)
        w = O.pme_reciprocal_space(pos.cpu().numpy(), q.cpu().numpy(), cell.cpu().numpy(), 0.3, (16, 20, 24), compute_forces=True)
        assert np.allclose(e.cpu().numpy(), w[0], rtol=1e-9, atol=1e-12) and np.allclose(f.cpu().numpy(), w[1], rtol=1e-9, atol=1e-12)
    # precomputed k-vectors (:1738)
    kv, k2 = generate_k_vectors_pme(dc, (16, 16, 16))
    ek, fk = pme_reciprocal_space(dp, dq, dc, compute_forces=True, k_vectors=kv, k_squared=k2, **kw)
    assert torch.allclose(ek, e1, rtol=1e-6) and torch.allclose(fk, fd, rtol=1e-6, atol=1e-12)
    # alpha matters, zero charges give zero (:1881, :1915)
    assert abs(float(pme_reciprocal_space(dp, dq, dc, alpha=0.5, mesh_dimensions=(16, 16, 16)).sum()) - float(e1.sum())) > 1e-6
    z, fz = pme_reciprocal_space(p64, torch.zeros_like(q64), c64, compute_forces=True, **kw)
    assert float(z.abs().max()) == 0.0 and float(fz.abs().max()) == 0.0
    # alpha forms (:1984-2047): 0-d tensor, wrong length, wrong type
    a0 = pme_reciprocal_space(p64, q64, c64, alpha=torch.tensor(0.3, dtype=torch.float64, device=DEV), mesh_dimensions=(16, 16, 16))
    assert torch.allclose(a0, e64, rtol=1e-12)
    with pytest.raises(ValueError):
        pme_reciprocal_space(p64, q64, c64, alpha=torch.tensor([0.3, 0.5], dtype=torch.float64, device=DEV), mesh_dimensions=(16, 16, 16))
    with pytest.raises(TypeError):
        pme_reciprocal_space(p64, q64, c64, alpha="invalid", mesh_dimensions=(16, 16, 16))
    # mesh given by spacing, or not at all (:2052, :2074)
    with pytest.raises(ValueError, match="Either mesh_dimensions or mesh_spacing must be provided"):
        pme_reciprocal_space(p64, q64, c64, alpha=0.3, mesh_dimensions=None, mesh_spacing=None)
    assert torch.isfinite(pme_reciprocal_space(p64, q64, c64, alpha=0.3, mesh_spacing=0.5)).all()


def test_pme_batch_consistency_scenarios():
    """TestPMEBatchConsistency, TestBatchWithDifferentAlpha, TestBatchPMEShapePaths (test_pme.py:798-1360, :1942-1980, :2271-2343): a batch
    equals its members run one by one -- energies, explicit forces, and autograd gradients w.r.t. positions, charges and cells."""
    from nvalchemiops.interactions.electrostatics import pme_reciprocal_space

    sizes, boxes, alphas = (6, 9, 4), (10.0, 12.0, 9.0), (0.3, 0.35, 0.4)
    systems = [_simple_system(n, torch.float64, b, seed=10 + i) for i, (n, b) in enumerate(zip(sizes, boxes))]
    P = torch.cat([s[0] for s in systems])
    Q = torch.cat([s[1] for s in systems])
    Cc = torch.stack([s[2] for s in systems])
    bi = torch.cat([torch.full((n,), i, dtype=torch.int32) for i, n in enumerate(sizes)]).to(DEV)
    al = torch.tensor(alphas, dtype=torch.float64, device=DEV)
    mesh = (16, 16, 16)
    eb, fb = pme_reciprocal_space(P, Q, Cc, alpha=al, mesh_dimensions=mesh, batch_idx=bi, compute_forces=True)
    off = 0
    for (p, q, c), n, a in zip(systems, sizes, alphas):
        e, f = pme_reciprocal_space(p, q, c, alpha=a, mesh_dimensions=mesh, compute_forces=True)
        assert torch.allclose(eb[off:off + n], e, rtol=1e-6, atol=1e-8) and torch.allclose(fb[off:off + n], f, rtol=1e-6, atol=1e-8)   # :840, :915, :1281, :1946
        assert float(fb[off:off + n].sum(0).abs().max()) < 1e-4                                            # :976 momentum per system
        off += n
    # autograd through the batch == autograd through the members (:1030-1280)
    Pg, Qg, Cg = P.clone().requires_grad_(True), Q.clone().requires_grad_(True), Cc.clone().requires_grad_(True)
    pme_reciprocal_space(Pg, Qg, Cg, alpha=al, mesh_dimensions=mesh, batch_idx=bi).sum().backward()
    off = 0
    for i, ((p, q, c), n, a) in enumerate(zip(systems, sizes, alphas)):
        pg, qg, cg = p.clone().requires_grad_(True), q.clone().requires_grad_(True), c.clone().requires_grad_(True)
        pme_reciprocal_space(pg, qg, cg, alpha=a, mesh_dimensions=mesh).sum().backward()
        assert torch.allclose(Pg.grad[off:off + n], pg.grad, rtol=1e-6, atol=1e-8)
        assert torch.allclose(Qg.grad[off:off + n], qg.grad, rtol=1e-6, atol=1e-8)
        assert torch.allclose(Cg.grad[i], cg.grad, rtol=1e-5, atol=1e-8)
        off += n
    # forces == -dE/dr (:1417): the explicit forces differentiate in k-space (i k), autograd differentiates the spline weights; on this
    # coarse 16^3 mesh over 9-12 A boxes the two discretisations differ at the 1e-4 level (finer meshes: tests/test_autograd_gpu.py)
    assert torch.allclose(-Pg.grad, fb, rtol=1e-2, atol=5e-4)


@pytest.mark.parametrize("fmt", ["list", "matrix"])
def test_particle_mesh_ewald_scenarios(fmt):
    """TestParticleMeshEwald, TestFullPMENeighborList, TestParticleMeshEwaldAutoEstimation, TestPMEChargeGradients (test_pme.py:1585-1627,
    :1811-1850, :2095-2270, :2344-2640): output layouts, automatic alpha / mesh, mesh spacing, accuracy-driven mesh, charge gradients ==
    autograd for single systems and batches."""
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
    from nvalchemiops.neighborlist import batch_cell_list, cell_list

    pos, q, cell = _simple_system(5)
    pbc = torch.tensor([True, True, True], device=DEV)
    if fmt == "list":
        nl, ptr, sh = cell_list(pos, 5.0, cell, pbc, return_neighbor_list=True)
        nb = dict(neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=sh)
    else:
        nm, num, sh = cell_list(pos, 5.0, cell, pbc, max_neighbors=32)
        nb = dict(neighbor_matrix=nm, neighbor_matrix_shifts=sh)
    e = particle_mesh_ewald(pos, q, cell, alpha=0.3, mesh_dimensions=(16, 16, 16), **nb)
    assert e.shape == (5,) and e.dtype == torch.float64                                                    # :1589
    for kw in (dict(alpha=None), dict(alpha=0.3, mesh_spacing=0.5), dict(alpha=0.3, mesh_dimensions=None, mesh_spacing=None, accuracy=1e-4),
               dict(alpha=None, mesh_dimensions=None, mesh_spacing=None)):                                 # :2099, :2131, :2164, :2235
        ee, ff = particle_mesh_ewald(pos, q, cell, compute_forces=True, **kw, **nb)
        assert ee.shape == (5,) and ff.shape == (5, 3) and torch.isfinite(ee).all() and torch.isfinite(ff).all()
    # the four output layouts
    out = particle_mesh_ewald(pos, q, cell, alpha=0.3, mesh_dimensions=(16, 16, 16), compute_forces=True, compute_charge_gradients=True, **nb)
    assert len(out) == 3 and out[1].shape == (5, 3) and out[2].shape == (5,)
    ecg = particle_mesh_ewald(pos, q, cell, alpha=0.3, mesh_dimensions=(16, 16, 16), compute_charge_gradients=True, **nb)
    assert len(ecg) == 2 and torch.equal(ecg[0], out[0]) and torch.equal(ecg[1], out[2])
    # charge gradients == d(sum E)/dq by autograd (:2416), forces == -d(sum E)/dr
    qg, pg = q.clone().requires_grad_(True), pos.clone().requires_grad_(True)
    particle_mesh_ewald(pg, qg, cell, alpha=0.3, mesh_dimensions=(16, 16, 16), **nb).sum().backward()
    assert torch.allclose(qg.grad, out[2], rtol=1e-6, atol=1e-10)
    assert torch.allclose(-pg.grad, out[1], rtol=1e-3, atol=1e-4)
    # batch of two systems (:2570)
    p2, q2, c2 = _simple_system(7, box=11.0, seed=4)
    P, Q, Cc = torch.cat([pos, p2]), torch.cat([q, q2]), torch.stack([cell, c2])
    bi = torch.cat([torch.zeros(5, dtype=torch.int32), torch.ones(7, dtype=torch.int32)]).to(DEV)
    bp = torch.ones((2, 3), dtype=torch.bool, device=DEV)
    if fmt == "list":
        nl, ptr, sh = batch_cell_list(P, 5.0, Cc, bp, bi, return_neighbor_list=True)
        nbb = dict(neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=sh)
    else:
        nm, num, sh = batch_cell_list(P, 5.0, Cc, bp, bi, max_neighbors=32)
        nbb = dict(neighbor_matrix=nm, neighbor_matrix_shifts=sh)
    al = torch.tensor([0.3, 0.3], dtype=torch.float64, device=DEV)
    eb, fb, cgb = particle_mesh_ewald(P, Q, Cc, alpha=al, mesh_dimensions=(16, 16, 16), batch_idx=bi, compute_forces=True,
                                      compute_charge_gradients=True, **nbb)
    assert torch.allclose(eb[:5], out[0], rtol=1e-6, atol=1e-8) and torch.allclose(fb[:5], out[1], rtol=1e-6, atol=1e-8)
    assert torch.allclose(cgb[:5], out[2], rtol=1e-6, atol=1e-8)
    Qg = Q.clone().requires_grad_(True)
    particle_mesh_ewald(P, Qg, Cc, alpha=al, mesh_dimensions=(16, 16, 16), batch_idx=bi, **nbb).sum().backward()
    assert torch.allclose(Qg.grad, cgb, rtol=1e-6, atol=1e-10)


# ------------------------------------------------------------------------------------------------------------------ test_ewald.py
def _ewald_lists(pos, cell, cutoff, batch_idx=None, width=64):
    from nvalchemiops.neighborlist import batch_cell_list, cell_list

    if batch_idx is None:
        pbc = torch.tensor([True, True, True], device=DEV)
        nl, ptr, sh = cell_list(pos, cutoff, cell, pbc, return_neighbor_list=True)
        nm, num, msh = cell_list(pos, cutoff, cell, pbc, max_neighbors=width)
    else:
        pbc = torch.ones((cell.shape[0], 3), dtype=torch.bool, device=DEV)
        nl, ptr, sh = batch_cell_list(pos, cutoff, cell, pbc, batch_idx, return_neighbor_list=True)
        nm, num, msh = batch_cell_list(pos, cutoff, cell, pbc, batch_idx, max_neighbors=width)
    assert int(num.max()) <= width
    return dict(neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=sh), dict(neighbor_matrix=nm, neighbor_matrix_shifts=msh)


def test_ewald_api_shapes_dtypes_and_physics():
    """TestDtypeSupport, the three *API classes, TestPhysicalProperties, TestSingleAtomSystem, TestLikeCharges, TestNonCubicCells,
    TestNeighborMatrixFormat, TestAlphaSensitivity (test_ewald.py:157-860, :3042-3670)."""
    from nvalchemiops.interactions.electrostatics import (ewald_real_space, ewald_reciprocal_space, ewald_summation,
                                                          generate_k_vectors_ewald_summation)

    for dtype in DTYPES:
        pos, q, cell = _simple_system(8, dtype, seed=21)
        cell3 = cell.unsqueeze(0)
        al = torch.tensor([0.3], dtype=dtype, device=DEV)
        lst, mat = _ewald_lists(pos, cell3, 4.5)
        kv = generate_k_vectors_ewald_summation(cell3, 2.5)
        for nb in (lst, mat):
            e = ewald_real_space(pos, q, cell3, al, **nb)
            e2, f = ewald_real_space(pos, q, cell3, al, compute_forces=True, **nb)
            assert e.shape == (8,) and f.shape == (8, 3) and e.dtype == dtype and f.dtype == dtype and torch.equal(e, e2)   # :162, :397, :424
        e = ewald_reciprocal_space(pos, q, cell3, kv, al)
        e2, f = ewald_reciprocal_space(pos, q, cell3, kv, al, compute_forces=True)
        assert e.shape == (8,) and f.shape == (8, 3) and e.dtype == dtype and f.dtype == dtype                                # :202, :554
        e, f = ewald_summation(pos, q, cell3, alpha=0.3, k_cutoff=2.5, compute_forces=True, **lst)
        assert e.shape == (8,) and f.shape == (8, 3) and e.dtype == dtype                                                       # :237, :689
    # fp32 vs fp64 (:277)
    pos, q, cell = _simple_system(8, torch.float64, seed=21)
    cell3, al = cell.unsqueeze(0), torch.tensor([0.3], dtype=torch.float64, device=DEV)
    lst, mat = _ewald_lists(pos, cell3, 4.5)
    e64, f64 = ewald_summation(pos, q, cell3, alpha=0.3, k_cutoff=2.5, compute_forces=True, **lst)
    l32 = {k: v for k, v in lst.items()}
    e32, f32 = ewald_summation(pos.float(), q.float(), cell3.float(), alpha=0.3, k_cutoff=2.5, compute_forces=True, **l32)
    assert torch.allclose(e32.double(), e64, rtol=1e-3, atol=1e-4) and torch.allclose(f32.double(), f64, rtol=1e-3, atol=1e-4)
    # matrix == list (:3526); against the oracle
    er_l, fr_l = ewald_real_space(pos, q, cell3, al, compute_forces=True, **lst)
    er_m, fr_m = ewald_real_space(pos, q, cell3, al, compute_forces=True, **mat)
    assert torch.allclose(er_l, er_m, rtol=1e-12, atol=1e-14) and torch.allclose(fr_l, fr_m, rtol=1e-11, atol=1e-14)
    oe, of = O.ewald_real_space(pos.cpu().numpy(), q.cpu().numpy(), cell3.cpu().numpy(), np.array([0.3]),
                                neighbor_matrix=mat["neighbor_matrix"].cpu().numpy(), neighbor_matrix_shifts=mat["neighbor_matrix_shifts"].cpu().numpy(),
                                mask_value=8, compute_forces=True)
    assert np.allclose(er_m.cpu().numpy(), oe, rtol=1e-10, atol=1e-13) and np.allclose(fr_m.cpu().numpy(), of, rtol=1e-10, atol=1e-13)
    # physics (:3046-3155, :3416-3484): opposite charges attract / like charges repel, q -> 2q scales E by 4, translation invariance
    dp, dq, dc = _dipole(sep=2.0)
    dc3 = dc.unsqueeze(0)
    dl, dm = _ewald_lists(dp, dc3, 4.5)
    e, f = ewald_summation(dp, dq, dc3, alpha=0.3, k_cutoff=3.0, compute_forces=True, **dl)
    assert float(e.sum()) < 0 and float(f[0, 0]) > 0 and float(f[1, 0]) < 0
    same = torch.tensor([1.0, 1.0], dtype=torch.float64, device=DEV)
    er, fr = ewald_real_space(dp, same, dc3, al, compute_forces=True, **dl)
    assert float(er.sum()) > 0 and float(fr[0, 0]) < 0 and float(fr[1, 0]) > 0
    e2 = ewald_summation(dp, 2.0 * dq, dc3, alpha=0.3, k_cutoff=3.0, **dl)
    assert torch.allclose(e2.sum(), 4.0 * e.sum(), rtol=1e-10)
    shift = torch.tensor([1.3, -0.7, 0.4], dtype=torch.float64, device=DEV)
    dl2, _ = _ewald_lists(dp + shift, dc3, 4.5)
    assert torch.allclose(ewald_summation(dp + shift, dq, dc3, alpha=0.3, k_cutoff=3.0, **dl2).sum(), e.sum(), rtol=1e-9)
    # alpha changes the split, not the total, once both sums are converged (:3610)
    tot = [float(ewald_summation(dp, dq, dc3, alpha=a, k_cutoff=6.0, **_ewald_lists(dp, dc3, 4.99)[0]).sum()) for a in (0.8, 1.0)]
    assert abs(tot[0] - tot[1]) < 1e-5 * abs(tot[0])
    # one atom (:3230, :3266)
    one, oq = torch.tensor([[5.0, 5.0, 5.0]], dtype=torch.float64, device=DEV), torch.tensor([1.0], dtype=torch.float64, device=DEV)
    ol, om = _ewald_lists(one, dc3, 4.0)
    e, f = ewald_real_space(one, oq, dc3, al, compute_forces=True, **ol)
    assert e.shape == (1,) and float(e.abs().max()) == 0.0 and float(f.abs().max()) == 0.0
    e, f = ewald_reciprocal_space(one, oq, dc3, generate_k_vectors_ewald_summation(dc3, 2.0), al, compute_forces=True)
    assert torch.isfinite(e).all() and float(f.abs().max()) < 1e-12
    # orthorhombic / triclinic cells (:3299-3411) against the oracle
    for cell in (torch.diag(torch.tensor([8.0, 10.0, 12.0], dtype=torch.float64)),
                 torch.tensor([[10.0, 0, 0], [2.0, 9.0, 0], [1.0, 1.5, 11.0]], dtype=torch.float64)):
        cell3 = cell.to(DEV).unsqueeze(0)
        frac = torch.rand((10, 3), generator=torch.Generator().manual_seed(8), dtype=torch.float64).to(DEV)
        pos = frac @ cell3[0]
        q = torch.tensor([1.0, -1.0] * 5, dtype=torch.float64, device=DEV)
        kv = generate_k_vectors_ewald_summation(cell3, 2.5)
        e, f = ewald_reciprocal_space(pos, q, cell3, kv, al, compute_forces=True)
        oe, of = O.ewald_reciprocal_space(pos.cpu().numpy(), q.cpu().numpy(), cell3.cpu().numpy(), kv.cpu().numpy(), 0.3)[:2]
        assert np.allclose(e.cpu().numpy(), oe, rtol=1e-10, atol=1e-13) and np.allclose(f.cpu().numpy(), of, rtol=1e-10, atol=1e-13)


def test_ewald_batches_alpha_forms_and_validation():
    """Batch API and consistency (test_ewald.py:454-550, :601-860, :2748-2930, :3918-4020, :4068-4110), per-system alpha (:808), alpha / cell
    forms (:3669-3810), missing neighbour data (:3581), automatic parameters and default mask value (:4249-4290, :4626-4740)."""
    from nvalchemiops.interactions.electrostatics import (ewald_real_space, ewald_reciprocal_space, ewald_summation,
                                                          generate_k_vectors_ewald_summation)

    sizes, boxes = (6, 9), (10.0, 12.0)
    systems = [_simple_system(n, torch.float64, b, seed=31 + i) for i, (n, b) in enumerate(zip(sizes, boxes))]
    P, Q = torch.cat([s[0] for s in systems]), torch.cat([s[1] for s in systems])
    Cc = torch.stack([s[2] for s in systems])
    bi = torch.cat([torch.full((n,), i, dtype=torch.int32) for i, n in enumerate(sizes)]).to(DEV)
    al = torch.tensor([0.3, 0.35], dtype=torch.float64, device=DEV)
    lst, mat = _ewald_lists(P, Cc, 4.5, bi)
    kvs = [generate_k_vectors_ewald_summation(c.unsqueeze(0), 2.5) for _, _, c in systems]
    kmax = max(k.shape[0] for k in kvs)
    # the batch takes [B,K,3]: pad the shorter set with zero vectors (a k = 0 entry contributes nothing: ewald_kernels.py k^2 < 1e-10 guard)
    KV = torch.stack([torch.cat([k, torch.zeros((kmax - k.shape[0], 3), dtype=k.dtype, device=DEV)]) for k in kvs])
    for nb in (lst, mat):
        eb, fb = ewald_real_space(P, Q, Cc, al, batch_idx=bi, compute_forces=True, **nb)
        assert eb.shape == (15,) and fb.shape == (15, 3)
        off = 0
        for i, ((p, q, c), n) in enumerate(zip(systems, sizes)):
            l1, m1 = _ewald_lists(p, c.unsqueeze(0), 4.5)
            e, f = ewald_real_space(p, q, c.unsqueeze(0), al[i:i + 1], compute_forces=True, **(l1 if nb is lst else m1))
            assert torch.allclose(eb[off:off + n], e, rtol=1e-10, atol=1e-13) and torch.allclose(fb[off:off + n], f, rtol=1e-9, atol=1e-13)   # :2752
            off += n
    eb, fb = ewald_reciprocal_space(P, Q, Cc, KV, al, batch_idx=bi, compute_forces=True)
    es, fs = ewald_summation(P, Q, Cc, alpha=al, k_vectors=KV, batch_idx=bi, compute_forces=True, **lst)
    off = 0
    for i, ((p, q, c), n) in enumerate(zip(systems, sizes)):
        e, f = ewald_reciprocal_space(p, q, c.unsqueeze(0), kvs[i], al[i:i + 1], compute_forces=True)
        assert torch.allclose(eb[off:off + n], e, rtol=1e-10, atol=1e-13) and torch.allclose(fb[off:off + n], f, rtol=1e-9, atol=1e-13)       # :2811
        l1, _ = _ewald_lists(p, c.unsqueeze(0), 4.5)
        e, f = ewald_summation(p, q, c.unsqueeze(0), alpha=al[i:i + 1], k_vectors=kvs[i], compute_forces=True, **l1)
        assert torch.allclose(es[off:off + n], e, rtol=1e-10, atol=1e-13) and torch.allclose(fs[off:off + n], f, rtol=1e-9, atol=1e-13)       # :2866, :808
        off += n
    assert ewald_reciprocal_space(P, Q, Cc, KV, al, batch_idx=bi).shape == (15,)                           # :4072
    # alpha forms: 0-d tensor and float are the same; wrong length and wrong type are errors (:3673-3770); a [3,3] cell is accepted (:3777)
    p, q, c = systems[0]
    l1, _ = _ewald_lists(p, c.unsqueeze(0), 4.5)
    base = ewald_summation(p, q, c.unsqueeze(0), alpha=0.3, k_cutoff=2.5, **l1)
    assert torch.allclose(ewald_summation(p, q, c.unsqueeze(0), alpha=torch.tensor(0.3, dtype=torch.float64, device=DEV), k_cutoff=2.5, **l1), base, rtol=1e-12)
    assert torch.allclose(ewald_summation(p, q, c, alpha=0.3, k_cutoff=2.5, **l1), base, rtol=1e-12)
    with pytest.raises(ValueError):
        ewald_summation(p, q, c.unsqueeze(0), alpha=torch.tensor([0.3, 0.5], dtype=torch.float64, device=DEV), k_cutoff=2.5, **l1)
    with pytest.raises(TypeError):
        ewald_summation(p, q, c.unsqueeze(0), alpha="invalid", k_cutoff=2.5, **l1)
    with pytest.raises(ValueError):
        ewald_real_space(p, q, c.unsqueeze(0), al[:1])                                                      # :3581 no neighbour data
    # automatic alpha / k_cutoff / k-vectors (:4253, :4630, :4663) and the default mask value of the matrix format (:4698)
    for kw in (dict(), dict(alpha=0.3), dict(alpha=0.3, k_cutoff=2.5)):
        e, f = ewald_summation(p, q, c.unsqueeze(0), compute_forces=True, **kw, **l1)
        assert e.shape == (6,) and f.shape == (6, 3) and torch.isfinite(e).all() and torch.isfinite(f).all()
    nm = torch.tensor([[1, 6], [0, 2], [1, 3], [2, 4], [3, 5], [4, 6]], dtype=torch.int32, device=DEV)
    e, f = ewald_summation(p, q, c.unsqueeze(0), alpha=0.3, k_cutoff=2.5, neighbor_matrix=nm,
                           neighbor_matrix_shifts=torch.zeros((6, 2, 3), dtype=torch.int32, device=DEV), compute_forces=True)
    assert torch.isfinite(e).all() and torch.isfinite(f).all()


def test_ewald_empty_inputs_and_charge_gradients():
    """TestExplicitChargeGradients, TestExplicitReciprocalChargeGradients, TestEmptyNeighborListEarlyReturns, TestReciprocalSpaceEmptyReturns,
    TestBatchEmptyInputs, TestNumericalStability (test_ewald.py:1373-2062, :3156-3225, :3810-3917, :4112-4210, :4473-4625)."""
    from nvalchemiops.interactions.electrostatics import ewald_real_space, ewald_reciprocal_space, generate_k_vectors_ewald_summation

    f64 = dict(dtype=torch.float64, device=DEV)
    pos = torch.tensor([[2.0, 5.0, 5.0], [8.0, 5.0, 5.0]], **f64)
    q = torch.tensor([1.0, -1.0], **f64)
    cell = torch.eye(3, **f64).unsqueeze(0) * 10.0
    al = torch.tensor([0.3], **f64)
    i32 = dict(dtype=torch.int32, device=DEV)
    empty_list = dict(neighbor_list=torch.zeros((2, 0), **i32), neighbor_ptr=torch.zeros(3, **i32), neighbor_shifts=torch.zeros((0, 3), **i32))
    empty_mat = dict(neighbor_matrix=torch.full((2, 4), 2, **i32), neighbor_matrix_shifts=torch.zeros((2, 4, 3), **i32))
    for nb in (empty_list, empty_mat):                                                                     # :1588, :3160, :3814-3917
        e, f, cg = ewald_real_space(pos, q, cell, al, compute_forces=True, compute_charge_gradients=True, **nb)
        assert e.shape == (2,) and f.shape == (2, 3) and cg.shape == (2,)
        assert float(e.abs().max()) == 0 and float(f.abs().max()) == 0 and float(cg.abs().max()) == 0
        assert float(ewald_real_space(pos, q, cell, al, **nb).abs().max()) == 0
    # batch, empty (:4477-4625)
    P, Q, Cc = torch.cat([pos, pos]), torch.cat([q, q]), cell.expand(2, -1, -1).contiguous()
    bi = torch.tensor([0, 0, 1, 1], **i32)
    al2 = torch.tensor([0.3, 0.3], **f64)
    bl = dict(neighbor_list=torch.zeros((2, 0), **i32), neighbor_ptr=torch.zeros(5, **i32), neighbor_shifts=torch.zeros((0, 3), **i32))
    bm = dict(neighbor_matrix=torch.full((4, 3), 4, **i32), neighbor_matrix_shifts=torch.zeros((4, 3, 3), **i32))
    for nb in (bl, bm):
        e, f = ewald_real_space(P, Q, Cc, al2, batch_idx=bi, compute_forces=True, **nb)
        assert e.shape == (4,) and f.shape == (4, 3) and float(e.abs().max()) == 0 and float(f.abs().max()) == 0
        assert float(ewald_real_space(P, Q, Cc, al2, batch_idx=bi, **nb).abs().max()) == 0
    # no k-vectors (:2023, :4116, :4149): zeros from every op but the single-system energy-only one, which keeps the self term
    k0 = torch.zeros((0, 3), **f64)
    e, f, cg = ewald_reciprocal_space(pos, q, cell, k0, al, compute_forces=True, compute_charge_gradients=True)
    assert e.shape == (2,) and f.shape == (2, 3) and cg.shape == (2,) and float(e.abs().max()) == 0 and float(f.abs().max()) == 0 and float(cg.abs().max()) == 0
    e, f = ewald_reciprocal_space(pos, q, cell, k0, al, compute_forces=True)
    assert float(e.abs().max()) == 0 and float(f.abs().max()) == 0
    e = ewald_reciprocal_space(pos, q, cell, k0, al)
    assert torch.allclose(e, -0.3 * q * q / np.sqrt(np.pi), rtol=1e-12)                                    # ewald.py:1365-1490 has no early return
    kb = torch.zeros((2, 0, 3), **f64)
    assert float(ewald_reciprocal_space(P, Q, Cc, kb, al2, batch_idx=bi).abs().max()) == 0
    e, f = ewald_reciprocal_space(P, Q, Cc, kb, al2, batch_idx=bi, compute_forces=True)
    assert e.shape == (4,) and f.shape == (4, 3) and float(e.abs().max()) == 0 and float(f.abs().max()) == 0
    # and the zeros are differentiable constants
    qg = q.clone().requires_grad_(True)
    ewald_reciprocal_space(pos, qg, cell, k0, al, compute_forces=True)[0].sum().backward()
    assert float(qg.grad.abs().max()) == 0
    # explicit charge gradients == autograd of the summed energy, various systems, with and without forces, list and matrix, batch
    for n, seed in ((4, 1), (9, 2), (16, 3)):                                                               # :1377-1830
        p, qq, c = _simple_system(n, torch.float64, 10.0, seed=40 + seed)
        c3 = c.unsqueeze(0)
        lst, mat = _ewald_lists(p, c3, 4.5)
        for nb in (lst, mat):
            e, f, cg = ewald_real_space(p, qq, c3, al, compute_forces=True, compute_charge_gradients=True, **nb)
            e1, cg1 = ewald_real_space(p, qq, c3, al, compute_charge_gradients=True, **nb)
            assert torch.equal(e, e1) and torch.equal(cg, cg1)                                              # :1489 without forces
            qg = qq.clone().requires_grad_(True)
            ewald_real_space(p, qg, c3, al, **nb).sum().backward()
            assert torch.allclose(qg.grad, cg, rtol=1e-8, atol=1e-12)
        kv = generate_k_vectors_ewald_summation(c3, 2.5)
        e, f, cg = ewald_reciprocal_space(p, qq, c3, kv, al, compute_forces=True, compute_charge_gradients=True)
        e1, cg1 = ewald_reciprocal_space(p, qq, c3, kv, al, compute_charge_gradients=True)
        assert torch.allclose(e, e1, rtol=1e-13) and torch.allclose(cg, cg1, rtol=1e-13)                   # :1876
        qg = qq.clone().requires_grad_(True)
        ewald_reciprocal_space(p, qg, c3, kv, al).sum().backward()
        assert torch.allclose(qg.grad, cg, rtol=1e-8, atol=1e-12)                                          # :1834, :1905
    # explicit outputs next to a live autograd graph (:1634): the backward of the energies still works
    p, qq, c = _simple_system(6, torch.float64, 10.0, seed=50)
    lst, _ = _ewald_lists(p, c.unsqueeze(0), 4.5)
    pg = p.clone().requires_grad_(True)
    e, f, cg = ewald_real_space(pg, qq, c.unsqueeze(0), al, compute_forces=True, compute_charge_gradients=True, **lst)
    e.sum().backward()
    assert torch.isfinite(pg.grad).all() and torch.allclose(-pg.grad, f.detach(), rtol=1e-8, atol=1e-12)
    # reciprocal sum converges with the cutoff (:3193)
    es = [float(ewald_reciprocal_space(p, qq, c.unsqueeze(0), generate_k_vectors_ewald_summation(c.unsqueeze(0), kc), al).sum()) for kc in (1.0, 2.0, 4.0)]
    assert abs(es[2] - es[1]) < abs(es[1] - es[0])


@pytest.mark.parametrize("fmt", ["list", "matrix"])
def test_ewald_autograd_scenarios(fmt):
    """TestAutogradRealSpace / ReciprocalSpace / FullEwald, TestBatchAutograd, TestAutogradWithMatrixFormat, TestBatchMatrixAutograd
    (test_ewald.py:1108-1372, :2063-2750, :4019, :4207, :4290-4472): gradients exist, are finite, equal the explicit forces, and the batch
    gives its members' gradients -- positions, charges and cells."""
    from nvalchemiops.interactions.electrostatics import ewald_real_space, ewald_reciprocal_space, ewald_summation, generate_k_vectors_ewald_summation

    sizes, boxes = (6, 9), (10.0, 11.0)
    systems = [_simple_system(n, torch.float64, b, seed=61 + i) for i, (n, b) in enumerate(zip(sizes, boxes))]
    al = torch.tensor([0.3, 0.3], dtype=torch.float64, device=DEV)
    grads = []
    for i, (p, q, c) in enumerate(systems):
        lst, mat = _ewald_lists(p, c.unsqueeze(0), 4.5)
        nb = lst if fmt == "list" else mat
        pg, qg, cg = p.clone().requires_grad_(True), q.clone().requires_grad_(True), c.unsqueeze(0).clone().requires_grad_(True)
        kv = generate_k_vectors_ewald_summation(cg, 2.5)                                                    # k-vectors follow the cell (:2117)
        e_real = ewald_real_space(pg, qg, cg, al[:1], **nb)
        e_rec = ewald_reciprocal_space(pg, qg, cg, kv, al[:1])
        gr = torch.autograd.grad(e_real.sum(), (pg, qg, cg), retain_graph=True)
        gk = torch.autograd.grad(e_rec.sum(), (pg, qg, cg), retain_graph=True)
        assert all(torch.isfinite(g).all() for g in gr + gk)
        _, fr = ewald_real_space(p, q, c.unsqueeze(0), al[:1], compute_forces=True, **nb)
        _, fk = ewald_reciprocal_space(p, q, c.unsqueeze(0), kv.detach(), al[:1], compute_forces=True)
        assert torch.allclose(-gr[0], fr, rtol=1e-8, atol=1e-12) and torch.allclose(-gk[0], fk, rtol=1e-8, atol=1e-12)     # :1169, :2142
        # full Ewald through ewald_summation (:2308-2480): same gradients as the two parts
        pg2, qg2, cg2 = p.clone().requires_grad_(True), q.clone().requires_grad_(True), c.unsqueeze(0).clone().requires_grad_(True)
        tot = ewald_summation(pg2, qg2, cg2, alpha=0.3, k_cutoff=2.5, **nb)
        gt = torch.autograd.grad(tot.sum(), (pg2, qg2, cg2))
        for a, b, cgrad in zip(gt, gr, gk):
            assert torch.allclose(a, b + cgrad, rtol=1e-8, atol=1e-11)
        grads.append(gt)
        # cell gradient against a central difference of the total energy under a homogeneous strain of one component (:2394)
        h = 1e-5
        def total(cc):
            s = torch.linalg.solve(c, cc[0])  # positions follow the cell
            kv_ = generate_k_vectors_ewald_summation(cc, 2.5)
            return float((ewald_real_space(p, q, cc, al[:1], **nb) + ewald_reciprocal_space(p, q, cc, kv_, al[:1])).sum())
        d = torch.zeros_like(c.unsqueeze(0))
        d[0, 1, 2] = h
        fd = (total(c.unsqueeze(0) + d) - total(c.unsqueeze(0) - d)) / (2 * h)
        assert abs(fd - float(gt[2][0, 1, 2])) < 1e-6 * max(1.0, abs(fd))
    # batch == members (:2488-2750, :4019, :4207, :4370)
    P, Q = torch.cat([s[0] for s in systems]), torch.cat([s[1] for s in systems])
    Cc = torch.stack([s[2] for s in systems])
    bi = torch.cat([torch.full((n,), i, dtype=torch.int32) for i, n in enumerate(sizes)]).to(DEV)
    lst, mat = _ewald_lists(P, Cc, 4.5, bi)
    nb = lst if fmt == "list" else mat
    Pg, Qg, Cg = P.clone().requires_grad_(True), Q.clone().requires_grad_(True), Cc.clone().requires_grad_(True)
    kvs = [generate_k_vectors_ewald_summation(Cg[i:i + 1], 2.5) for i in range(2)]
    kmax = max(k.shape[0] for k in kvs)
    KV = torch.stack([torch.cat([k, torch.zeros((kmax - k.shape[0], 3), dtype=k.dtype, device=DEV)]) for k in kvs])
    tot = ewald_summation(Pg, Qg, Cg, alpha=al, k_vectors=KV, batch_idx=bi, **nb)
    gb = torch.autograd.grad(tot.sum(), (Pg, Qg, Cg))
    off = 0
    for i, n in enumerate(sizes):
        assert torch.allclose(gb[0][off:off + n], grads[i][0], rtol=1e-8, atol=1e-11)
        assert torch.allclose(gb[1][off:off + n], grads[i][1], rtol=1e-8, atol=1e-11)
        assert torch.allclose(gb[2][i], grads[i][2][0], rtol=1e-7, atol=1e-10)
        off += n


# ------------------------------------------------------------------------------------- test_naive.py / test_batch_naive.py / test_naive_dual.py
def _cubic8(dtype):
    g = torch.arange(2, dtype=dtype)
    pos = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).to(DEV)
    return pos, (torch.eye(3, dtype=dtype) * 2.0).reshape(1, 3, 3).to(DEV), torch.tensor([[True, True, True]], device=DEV)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("half_fill", [False, True])
def test_naive_family_edge_and_error_scenarios(dtype, half_fill):
    """TestNaiveMainAPI / Robustness / MemoryAndPerformance and their batch and dual-cutoff twins (test_naive.py:873-945, :1112-1166, :1400-1450;
    test_batch_naive.py:647-742; test_naive_dual.py:685-790): empty / single atom / zero cutoff, the cell-without-pbc errors, an extremely
    elongated cell, cutoffs beyond the box, rows that overflow max_neighbors, the max_neighbors2 default."""
    from nvalchemiops.neighborlist import (batch_naive_neighbor_list, batch_naive_neighbor_list_dual_cutoff, naive_neighbor_list,
                                           naive_neighbor_list_dual_cutoff)

    i32 = dict(dtype=torch.int32, device=DEV)
    empty = torch.empty((0, 3), dtype=dtype, device=DEV)
    one = torch.zeros((1, 3), dtype=dtype, device=DEV)
    pos, cell, pbc = _cubic8(dtype)
    # -- naive
    nm, num = naive_neighbor_list(positions=empty, cutoff=1.0, pbc=None, cell=None, max_neighbors=10, half_fill=half_fill)
    assert nm.shape == (0, 10) and num.shape == (0,)
    nm, num = naive_neighbor_list(positions=one, cutoff=1.0, pbc=None, cell=None, max_neighbors=10, half_fill=half_fill)
    assert int(num[0]) == 0
    nm, num = naive_neighbor_list(positions=pos[:4], cutoff=0.0, pbc=None, cell=None, max_neighbors=10, half_fill=half_fill)
    assert int(num.abs().sum()) == 0
    with pytest.raises(ValueError, match="If cell is provided, pbc must also be provided"):
        naive_neighbor_list(pos, 1.0, pbc=None, cell=cell, max_neighbors=10)
    with pytest.raises(ValueError, match="If pbc is provided, cell must also be provided"):
        naive_neighbor_list(pos, 1.0, pbc=pbc, cell=None, max_neighbors=10)
    # elongated cell 10 x 0.1 x 0.1 with cutoff 0.2: two image shells along the short axes (:1112)
    long_cell = torch.tensor([[[10.0, 0, 0], [0, 0.1, 0], [0, 0, 0.1]]], dtype=dtype, device=DEV)
    lp = torch.rand((10, 3), generator=torch.Generator().manual_seed(1), dtype=dtype).to(DEV) * torch.tensor([10.0, 0.1, 0.1], dtype=dtype, device=DEV)
    nm, num, sh = naive_neighbor_list(positions=lp, cutoff=0.2, pbc=pbc, cell=long_cell, max_neighbors=64, half_fill=half_fill)
    ref = O.naive(lp.cpu().numpy(), 0.2, cell=long_cell.cpu().numpy(), pbc=pbc.cpu().numpy(), max_neighbors=64, half_fill=half_fill)
    assert np.array_equal(num.cpu().numpy(), ref[1]) and int(num.min()) >= (1 if half_fill else 2)   # at least the atom's own images along y and z
    # cutoff 5 in a box of 2 (:1140): every atom sees many images
    nm, num, sh = naive_neighbor_list(positions=pos, cutoff=5.0, pbc=pbc, cell=cell, max_neighbors=512, half_fill=half_fill)
    ref = O.naive(pos.cpu().numpy(), 5.0, cell=cell.cpu().numpy(), pbc=pbc.cpu().numpy(), max_neighbors=512, half_fill=half_fill)
    assert int(num.min()) > 0 and np.array_equal(num.cpu().numpy(), ref[1])
    assert int(num.sum()) == (8 * 484 if not half_fill else 4 * 484)                                  # 485 lattice vectors with |v|^2 < 25, minus the atom itself
    # rows longer than max_neighbors = 3 (:1400): shapes stay, counts keep counting
    nm, num, sh = naive_neighbor_list(positions=pos, cutoff=2.0, pbc=pbc, cell=cell, max_neighbors=3, half_fill=half_fill)
    assert nm.shape == (8, 3) and sh.shape == (8, 3, 3) and num.shape == (8,) and int(num.max()) > 3
    # -- batch naive
    nm, num = batch_naive_neighbor_list(positions=empty, cutoff=1.0, batch_idx=torch.empty(0, **i32), batch_ptr=torch.tensor([0], **i32),
                                        max_neighbors=10, pbc=None, cell=None, half_fill=half_fill)
    assert nm.shape == (0, 10) and num.shape == (0,)
    nm, num = batch_naive_neighbor_list(positions=one, cutoff=1.0, batch_idx=torch.tensor([0], **i32), batch_ptr=torch.tensor([0, 1], **i32),
                                        max_neighbors=10, pbc=None, cell=None, half_fill=half_fill)
    assert int(num[0]) == 0
    bi, bp = torch.tensor([0, 0, 0, 1, 1, 1, 1], **i32), torch.tensor([0, 3, 7], **i32)
    nm, num = batch_naive_neighbor_list(positions=pos[:7], cutoff=0.0, batch_idx=bi, batch_ptr=bp, max_neighbors=10, pbc=None, cell=None,
                                        half_fill=half_fill)
    assert int(num.abs().sum()) == 0
    cells2, pbc2 = cell.expand(2, -1, -1).contiguous(), pbc.expand(2, -1).contiguous()
    with pytest.raises(ValueError, match="If cell is provided, pbc must also be provided"):
        batch_naive_neighbor_list(pos[:7], 1.0, batch_idx=bi, batch_ptr=bp, max_neighbors=10, pbc=None, cell=cells2)
    with pytest.raises(ValueError, match="If pbc is provided, cell must also be provided"):
        batch_naive_neighbor_list(pos[:7], 1.0, batch_idx=bi, batch_ptr=bp, max_neighbors=10, pbc=pbc2, cell=None)
    nm, num, sh = batch_naive_neighbor_list(pos[:7], 2.0, batch_idx=bi, batch_ptr=bp, max_neighbors=3, pbc=pbc2, cell=cells2, half_fill=half_fill)
    assert nm.shape == (7, 3) and sh.shape == (7, 3, 3) and int(num.max()) > 3
    # -- dual cutoff
    with pytest.raises(ValueError, match="If cell is provided, pbc must also be provided"):
        naive_neighbor_list_dual_cutoff(pos, 1.0, 1.5, pbc=None, cell=cell, max_neighbors1=10)
    with pytest.raises(ValueError, match="If pbc is provided, cell must also be provided"):
        naive_neighbor_list_dual_cutoff(pos, 1.0, 1.5, pbc=pbc, cell=None, max_neighbors1=10)
    two = torch.tensor([[0.0, 0, 0], [1.0, 0, 0]], dtype=dtype, device=DEV)
    out = naive_neighbor_list_dual_cutoff(positions=two, cutoff1=0.5, cutoff2=1.5, max_neighbors1=10, half_fill=half_fill)   # :767
    assert len(out) == 4 and out[0].shape == (2, 10) and out[2].shape == (2, 10)
    assert int(out[1].sum()) == 0 and int(out[3].sum()) == (1 if half_fill else 2)
    out = naive_neighbor_list_dual_cutoff(positions=empty, cutoff1=0.5, cutoff2=1.5, max_neighbors1=10, half_fill=half_fill)
    assert out[0].shape == (0, 10) and out[1].shape == (0,)
    out = batch_naive_neighbor_list_dual_cutoff(pos[:7], 1.1, 1.5, batch_idx=bi, batch_ptr=bp, pbc=pbc2, cell=cells2, max_neighbors1=30, max_neighbors2=40,
                                                half_fill=half_fill)
    assert len(out) == 6 and out[0].shape == (7, 30) and out[3].shape == (7, 40) and int(out[4].sum()) >= int(out[1].sum()) > 0


# ----------------------------------------------------------------------------------------------------------------- test_spline.py
@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_spline_property_scenarios(order):
    """TestSplineSpread / Gather / GatherGradient / GatherVec3 / NonCubicCell / SpreadGatherConsistency and their batch twins
    (test_spline.py:174-1215): charge conservation, shapes, locality, centre of mass of the spread, uniform meshes, adjointness,
    batch == members, one 2-D cell shared by a batch (:1814-1980)."""
    from nvalchemiops.spline import spline_gather, spline_gather_gradient, spline_gather_vec3, spline_spread

    f64 = dict(dtype=torch.float64, device=DEV)
    cell = torch.eye(3, **f64) * 10.0
    g = torch.Generator().manual_seed(5)
    pos = (torch.rand((12, 3), generator=g, dtype=torch.float64) * 10.0).to(DEV)
    q = torch.randn(12, generator=g, dtype=torch.float64).to(DEV)
    mesh = spline_spread(pos, q, cell, [16, 16, 16], order)                                                 # a list of dims is accepted
    assert mesh.shape == (16, 16, 16) and mesh.dtype == torch.float64
    assert torch.allclose(mesh.sum(), q.sum(), rtol=1e-10, atol=1e-12)                                        # :179
    one = torch.tensor([[5.0, 5.0, 5.0]], **f64)
    m1 = spline_spread(one, torch.ones(1, **f64), cell, (16, 16, 16), order)
    assert int((m1 != 0).sum()) <= order ** 3                                                                 # :227 locality
    if order >= 2:
        for p in ([5.0, 5.0, 5.0], [2.5, 7.5, 4.0], [3.3, 6.1, 8.2]):                                          # :252 centre of mass == position
            m = spline_spread(torch.tensor([p], **f64), torch.ones(1, **f64), cell, (16, 16, 16), order)
            coords = torch.arange(16, **f64) * (10.0 / 16)
            com = torch.stack([(m * coords.view(-1, 1, 1)).sum(), (m * coords.view(1, -1, 1)).sum(), (m * coords.view(1, 1, -1)).sum()]) / m.sum()
            assert torch.allclose(com, torch.tensor(p, **f64), rtol=1e-10)
    # uniform mesh: gather returns the constant, gather_gradient zero, gather_vec3 q * field (:337, :385, :440)
    # (to 1e-6 like the reference's test: weights <= 1e-8 are dropped by the gather kernels, spline.py:608)
    assert torch.allclose(spline_gather(pos, torch.full((16, 16, 16), 2.5, **f64), cell, order), torch.full((12,), 2.5, **f64), rtol=1e-6)
    gg = spline_gather_gradient(pos, q, torch.full((16, 16, 16), 2.5, **f64), cell, order)
    assert gg.shape == (12, 3) and float(gg.abs().max()) < 1e-9
    field = torch.tensor([1.0, -2.0, 0.5], **f64).expand(16, 16, 16, 3).contiguous()
    v3 = spline_gather_vec3(pos, q, field, cell, order)
    assert v3.shape == (12, 3) and torch.allclose(v3, q.unsqueeze(1) * field[0, 0, 0], rtol=1e-6, atol=1e-9)
    # <spread(q), phi> == <q, gather(phi)> (:637) up to the gather's 1e-8 weight cut
    phi = torch.randn((16, 16, 16), generator=g, dtype=torch.float64).to(DEV)
    assert abs(float((mesh * phi).sum() - (q * spline_gather(pos, phi, cell, order)).sum())) < 1e-6
    # triclinic cell (:601): conservation still holds
    tri = torch.tensor([[10.0, 0, 0], [2.0, 9.0, 0], [1.0, 1.5, 11.0]], **f64)
    assert torch.allclose(spline_spread(pos @ torch.linalg.inv(cell) @ tri, q, tri, (12, 16, 20), order).sum(), q.sum(), rtol=1e-10, atol=1e-12)
    # batch: shapes, conservation per system, members (:710-1215); a single 2-D cell for the whole batch (:1818-1900)
    bi = torch.cat([torch.zeros(5, dtype=torch.int32), torch.ones(7, dtype=torch.int32)]).to(DEV)
    cells = torch.stack([cell, cell * 1.2])
    mb = spline_spread(pos, q, cells, (16, 16, 16), order, batch_idx=bi)
    assert mb.shape == (2, 16, 16, 16)
    assert torch.allclose(mb[0].sum(), q[:5].sum(), rtol=1e-6, atol=1e-6) and torch.allclose(mb[1].sum(), q[5:].sum(), rtol=1e-6, atol=1e-6)   # batch spread drops weights <= 1e-8 (spline.py:820)
    assert torch.allclose(mb[1], spline_spread(pos[5:], q[5:], cells[1], (16, 16, 16), order), rtol=1e-7, atol=1e-7)   # batch cut: weights > 1e-8 only
    phis = torch.randn((2, 16, 16, 16), generator=g, dtype=torch.float64).to(DEV)
    gb = spline_gather(pos, phis, cells, order, batch_idx=bi)
    assert torch.allclose(gb[:5], spline_gather(pos[:5], phis[0], cells[0], order), rtol=1e-12) and torch.allclose(gb[5:], spline_gather(pos[5:], phis[1], cells[1], order), rtol=1e-12)
    fb = spline_gather_gradient(pos, q, phis, cells, order, batch_idx=bi)
    assert torch.allclose(fb[5:], spline_gather_gradient(pos[5:], q[5:], phis[1], cells[1], order), rtol=1e-11, atol=1e-13)
    vb = spline_gather_vec3(pos, q, torch.stack([field, 2 * field]), cells, order, batch_idx=bi)
    assert vb.shape == (12, 3) and torch.allclose(vb[5:], 2 * q[5:].unsqueeze(1) * field[0, 0, 0], rtol=1e-6, atol=1e-9)
    shared = spline_spread(pos, q, cell, (16, 16, 16), order, batch_idx=bi)                                   # one [3,3] cell, two systems
    assert shared.shape == (2, 16, 16, 16) and torch.allclose(shared.sum((1, 2, 3)), torch.stack([q[:5].sum(), q[5:].sum()]), rtol=1e-6, atol=1e-6)
    assert spline_gather(pos, phis, cell, order, batch_idx=bi).shape == (12,)
    assert spline_gather_vec3(pos, q, torch.stack([field, field]), cell, order, batch_idx=bi).shape == (12, 3)
    assert spline_gather_gradient(pos, q, phis, cell, order, batch_idx=bi).shape == (12, 3)


def test_spline_channels_and_deconvolution_scenarios():
    """TestMultiChannel* (test_spline.py:1216-1530, :1764-1980) and TestBSplineDeconvolution / RoundTrip / DeconvolutionCoverage
    (:1533-1663, :1982-2030)."""
    from nvalchemiops.spline import (compute_bspline_deconvolution, compute_bspline_deconvolution_1d, spline_gather, spline_gather_channels,
                                     spline_spread, spline_spread_channels)

    f64 = dict(dtype=torch.float64, device=DEV)
    cell = torch.eye(3, **f64) * 10.0
    g = torch.Generator().manual_seed(6)
    pos = (torch.rand((10, 3), generator=g, dtype=torch.float64) * 10.0).to(DEV)
    vals = torch.randn((10, 4), generator=g, dtype=torch.float64).to(DEV)
    m = spline_spread_channels(pos, vals, cell, (8, 8, 8), 4)
    assert m.shape == (4, 8, 8, 8) and torch.allclose(m.sum((1, 2, 3)), vals.sum(0), rtol=1e-10, atol=1e-12)     # :1221, :1244
    assert torch.allclose(m[2], spline_spread(pos, vals[:, 2].contiguous(), cell, (8, 8, 8), 4), rtol=1e-12, atol=1e-15)
    out = spline_gather_channels(pos, m, cell, 4)
    assert out.shape == (10, 4) and torch.allclose(out[:, 1], spline_gather(pos, m[1], cell, 4), rtol=1e-12)        # :1279, :1329
    assert torch.allclose(spline_gather_channels(pos, torch.ones((3, 8, 8, 8), **f64), cell, 4), torch.ones((10, 3), **f64), rtol=1e-6)   # :1299
    bi = torch.cat([torch.zeros(4, dtype=torch.int32), torch.ones(6, dtype=torch.int32)]).to(DEV)
    cells = torch.stack([cell, cell])
    mb = spline_spread_channels(pos, vals, cells, (8, 8, 8), 4, batch_idx=bi)
    assert mb.shape == (2, 4, 8, 8, 8) and spline_gather_channels(pos, mb, cells, 4, batch_idx=bi).shape == (10, 4)  # :1358, :1399
    assert spline_spread_channels(pos, vals, cell, (8, 8, 8), 4, batch_idx=bi).shape == (2, 4, 8, 8, 8)               # :1927 one 2-D cell
    vg, pg = vals.clone().requires_grad_(True), pos.clone().requires_grad_(True)                                     # :1439-1530
    (spline_spread_channels(pg, vg, cell, (8, 8, 8), 4) * torch.randn((4, 8, 8, 8), generator=g, dtype=torch.float64).to(DEV)).sum().backward()
    assert torch.isfinite(vg.grad).all() and torch.isfinite(pg.grad).all() and float(pg.grad.abs().max()) > 0
    mg = m.clone().requires_grad_(True)
    spline_gather_channels(pos, mg, cell, 4).sum().backward()
    assert torch.isfinite(mg.grad).all() and torch.allclose(mg.grad.sum((1, 2, 3)), torch.full((4,), 10.0, **f64), rtol=1e-6)
    # deconvolution
    for order in (2, 3, 4, 5, 6):
        d = compute_bspline_deconvolution((8, 12, 16), spline_order=order)
        assert d.shape == (8, 12, 16) and abs(float(d[0, 0, 0]) - 1.0) < 1e-6 and bool((d > 0).all())
        c = compute_bspline_deconvolution((8, 8, 8), spline_order=order)
        for i in range(1, 4):
            assert torch.allclose(c[i, 0, 0], c[-i, 0, 0], rtol=1e-6) and torch.allclose(c[0, i, 0], c[0, -i, 0], rtol=1e-6) and torch.allclose(c[0, 0, i], c[0, 0, -i], rtol=1e-6)
    d1 = compute_bspline_deconvolution_1d(16, spline_order=4)
    assert d1.shape == (16,) and abs(float(d1[0]) - 1.0) < 1e-6 and bool((d1 > 0).all()) and d1.device.type == "cpu"
    assert compute_bspline_deconvolution((8, 8, 8), spline_order=4, device=torch.device(DEV)).device.type == "cuda"
    assert compute_bspline_deconvolution_1d(16, spline_order=6, device=torch.device(DEV)).device.type == "cuda"
    p3 = torch.tensor([[2.0, 2.0, 2.0], [5.0, 5.0, 5.0], [7.0, 3.0, 4.0]], **f64)                                   # :1618 round trip
    q3 = torch.tensor([1.0, -0.5, 0.3], **f64)
    mesh = spline_spread(p3, q3, cell, (32, 32, 32), spline_order=4)
    corr = torch.fft.ifftn(torch.fft.fftn(mesh) * compute_bspline_deconvolution((32, 32, 32), spline_order=4, device=torch.device(DEV))).real
    assert torch.isfinite(spline_gather(p3, corr, cell, spline_order=4)).all() and torch.isfinite(spline_gather(p3, mesh, cell, spline_order=4)).all()

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

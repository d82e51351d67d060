"""Coordination numbers as a by-product of the neighbour search (round 6; `mi_nl_neighbors_packed_cn` -> `mi_d3_packed_cn`)
and the device-side guards that decide whether `dftd3` may trust what the search left behind.

Reference pass being replaced: `_cn_kernel_nm` (interactions/dispersion/dftd3.py:833-941) with `_cn_counting` (:608-645): CN_i = sum over
the stored row of 1 / (1 + exp(-k1 ((rcov_i + rcov_j) / r - 1))).  Checked here, all through the public API and the C ABI behind it:
  * the search's numbers against the oracle's CN pass on the same list (rtol = atol = 1e-6, the reference's own CPU-vs-GPU bar) for the
    tiled and the wave-per-atom search kernel, fp32 / fp64, triclinic, batches, atoms outside the tables (Z = 0);
  * D3 outputs with the adopted numbers against the oracle and against the ordinary pass (same tolerances as tests/test_d3_gpu.py);
  * the fingerprint: other positions (through torch AND through a raw-pointer write torch cannot see), other species, another cell,
    another k1 -> the ordinary pass runs, the answer is the oracle's for the inputs as they ARE;
  * overflowed rows -> the search raises its flag, the ordinary pass runs;
  * the companion guard: matrix / shifts edited through `tensor.data`, through a raw-pointer kernel and through `untyped_storage().copy_`
    (none of which moves a torch version counter) are caught by the sampled / full device-side comparison, and `invalidate` is the
    documented way out for a single-entry edit."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import systems as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FP = dict(a1=0.4289, a2=4.4407, s8=0.7875)


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def _params(zmax=17):
    tables = S.d3_test_tables(zmax)
    from nvalchemiops.interactions.dispersion import D3Parameters

    return tables, D3Parameters(rcov=_t(tables["rcov"]), r4r2=_t(tables["r4r2"]), c6ab=_t(tables["c6ab"]), cn_ref=_t(tables["cn_ref"]))


@pytest.fixture
def engine(monkeypatch):
    from nvalchemiops.neighborlist import _engine as E

    monkeypatch.setattr(E, "_PACKED_POLICY", "1")
    monkeypatch.setattr(E, "_PACKED_WANTED", set())
    return E


def _cn_block(E, nm):
    """(flag, cutoff, cn[N]) of the coordination-number block riding with the companion of `nm`."""
    rec = getattr(nm, E._PACKED_ATTR)
    assert rec.cn is not None
    hdr = rec.cn[:16].view(torch.int32)
    return int(hdr[0]), float(rec.cn[8:12].view(torch.float32)[0]), rec.cn[1024:].view(torch.float32)


def _search(E, pos, cutoff, cell, numbers, rcov, m, k1=16.0, bi=None, pbc=None):
    from nvalchemiops.neighborlist import batch_cell_list, cell_list

    n = pos.shape[0]
    nm = torch.empty((n, m), dtype=torch.int32, device=DEV)
    sh = torch.empty((n, m, 3), dtype=torch.int32, device=DEV)
    num = torch.empty(n, dtype=torch.int32, device=DEV)
    E.attach_dftd3_context(nm, numbers, rcov, k1)
    pbc = torch.tensor([True] * 3, device=DEV) if pbc is None else pbc
    if bi is None:
        cell_list(pos, cutoff, cell, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    else:
        batch_cell_list(pos, cutoff, cell, pbc, bi, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    return nm, num, sh


def _d3(pos, z, p, nm, sh, cell, bi=None, nsys=None, **kw):
    from nvalchemiops.interactions.dispersion import dftd3

    args = dict(d3_params=p, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=cell, compute_virial=True, **FP)
    args.update(kw)
    if bi is not None:
        args.update(batch_idx=bi, num_systems=nsys)
    return dftd3(pos, z, **args)


def _oracle(pos, numbers, tables, nm, sh, cell, bi=None, **kw):
    with O.d3_wide_sums():
        return O.dftd3(pos, numbers, tables, neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy(), cell=cell,
                       compute_virial=True, batch_idx=bi, **{**FP, **kw})


def _close(a, b, rtol=1e-6, atol=1e-6, scale=None):
    """|a - b| <= atol + rtol |b| (+ rtol * scale: the extra absolute room tests/test_d3_gpu.py gives the cancelling sums -- forces
    5e-6 max|F| (scale = 5 max|F|), virial 2e-7 max|V| (scale = 0.2 max|V|))."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    tol = atol + rtol * np.abs(b) + (rtol * scale if scale is not None else 0.0)
    return bool((np.abs(a - b) <= tol).all())


def _fs(ref):
    return 5 * np.abs(ref[1]).max()


def _vs(ref):
    return 0.2 * np.abs(ref[3]).max()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,cutoff,m", [(4000, 11.0, 384), (300, 6.0, 96)])  # dense cells (tiled kernel) and sparse cells (wave-per-atom kernel)
def test_search_cn_equals_the_oracle_pass(engine, dtype, n, cutoff, m):
    tables, p = _params()
    pos, cell, _, numbers = S.fcc_box(n, dtype=dtype)
    numbers = numbers.copy()
    numbers[7] = 0  # an atom outside the tables: searched like any other, no coordination number, no term in anybody else's
    tp, tz, tc = _t(pos), _t(numbers), _t(cell)
    nm, num, sh = _search(engine, tp, cutoff, tc, tz, p.rcov, m)
    assert int(num.max()) <= m
    flag, rc, cn = _cn_block(engine, nm)
    assert flag == 0 and abs(rc - cutoff) < 1e-6
    ref = _oracle(pos.astype(dtype), numbers, tables, nm, sh, cell.astype(dtype))
    assert _close(cn, ref[2]), np.abs(cn.cpu().numpy() - ref[2]).max()
    assert float(cn[7]) == 0.0
    # dftd3 adopts them: CN output = the search's numbers exactly, everything else within the usual bars of the oracle
    e, f, c, v = _d3(tp, tz, p, nm, sh, tc[None])
    live = numbers != 0
    assert torch.equal(c[torch.as_tensor(live, device=DEV)], cn[torch.as_tensor(live, device=DEV)])
    assert _close(e, ref[0]) and _close(c, ref[2])
    assert _close(f, ref[1], scale=_fs(ref)) and _close(v, ref[3], scale=_vs(ref))
    # and the ordinary pass on the same list (switch: the search's numbers ignored) agrees to fp32 rounding of the two summation orders
    import os
    os.environ["NVALCHEMIOPS_D3_SEARCH_CN"] = "0"
    try:
        e0, f0, c0, v0 = _d3(tp, tz, p, nm, sh, tc[None])
    finally:
        del os.environ["NVALCHEMIOPS_D3_SEARCH_CN"]
    assert _close(c, c0) and _close(e, e0)
    assert _close(f, f0, scale=_fs(ref))


def test_search_cn_triclinic_batch(engine):
    """A ragged batch of triclinic periodic systems through `batch_cell_list`: per-system cells enter the shifted distances."""
    tables, p = _params()
    rng = np.random.default_rng(11)
    parts, cells, bi = [], [], []
    for s, n in enumerate((700, 450, 900)):
        pos, cell, _, numbers = S.fcc_box(n, seed=40 + s, dtype=np.float32)
        tilt = np.eye(3, dtype=np.float32) + 0.08 * rng.standard_normal((3, 3)).astype(np.float32)
        parts.append((pos @ tilt, numbers))
        cells.append(cell @ tilt)
        bi.append(np.full(n, s, dtype=np.int32))
    pos = np.concatenate([q[0] for q in parts]).astype(np.float32)
    numbers = np.concatenate([q[1] for q in parts])
    cell = np.stack(cells).astype(np.float32)
    bi = np.concatenate(bi)
    tp, tz, tc, tb = _t(pos), _t(numbers), _t(cell), _t(bi)
    pbc = torch.ones((3, 3), dtype=torch.bool, device=DEV)
    nm, num, sh = _search(engine, tp, 9.0, tc, tz, p.rcov, 320, bi=tb, pbc=pbc)
    assert int(num.max()) <= 320
    flag, _, cn = _cn_block(engine, nm)
    assert flag == 0
    ref = _oracle(pos, numbers, tables, nm, sh, cell, bi=bi)
    assert _close(cn, ref[2])
    e, f, c, v = _d3(tp, tz, p, nm, sh, tc, bi=tb, nsys=3)
    assert torch.equal(c, cn)
    assert _close(e, ref[0]) and _close(f, ref[1], scale=_fs(ref))


def _setup(engine, n=2048, cutoff=9.0, m=256, seed=3):
    tables, p = _params()
    pos, cell, _, numbers = S.fcc_box(n, seed=seed, dtype=np.float32)
    tp, tz, tc = _t(pos), _t(numbers), _t(cell)
    nm, num, sh = _search(engine, tp, cutoff, tc, tz, p.rcov, m)
    assert int(num.max()) <= m
    return tables, p, pos, cell, numbers, tp, tz, tc, nm, num, sh


def test_moved_positions_do_not_get_the_stale_numbers(engine):
    """An MD step that reuses its list: positions change between the search and dftd3.  Whether torch sees the write (in-place op) or not
    (raw-pointer kernel of this library on `data_ptr()`), the fingerprint differs and the answer is the oracle's for the NEW positions."""
    from nvalchemiops import _capi as C

    tables, p, pos, cell, numbers, tp, tz, tc, nm, num, sh = _setup(engine)
    _, _, cn_search = _cn_block(engine, nm)
    cn_search = cn_search.clone()
    e0, f0, c0, _ = _d3(tp, tz, p, nm, sh, tc[None])
    assert torch.equal(c0, cn_search)
    # (a) through torch
    delta = 0.02 * torch.randn(tp.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    tp.add_(delta)
    e1, f1, c1, _ = _d3(tp, tz, p, nm, sh, tc[None])
    ref1 = _oracle(tp.cpu().numpy(), numbers, tables, nm, sh, cell)
    assert _close(c1.cpu().numpy(), ref1[2]) and _close(e1.cpu().numpy(), ref1[0])
    assert not torch.equal(c1, cn_search)
    # (b) behind torch's back: the library's calibration fill writes the first 4 KiB of the positions through the raw pointer
    v = tp._version
    C.lib().mi_calibrate_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_float, ctypes.c_void_p]
    rc = C.lib().mi_calibrate_fill(ctypes.c_void_p(tp.data_ptr()), ctypes.c_size_t(16384), ctypes.c_float(1.25), C.stream_of(tp))
    assert rc == 0 and tp._version == v
    torch.cuda.synchronize()
    e2, f2, c2, _ = _d3(tp, tz, p, nm, sh, tc[None])
    ref2 = _oracle(tp.cpu().numpy(), numbers, tables, nm, sh, cell)
    assert _close(c2.cpu().numpy(), ref2[2]) and _close(e2.cpu().numpy(), ref2[0])
    assert not torch.equal(c2, c1)


def test_other_species_cell_or_k1_do_not_get_the_stale_numbers(engine):
    tables, p, pos, cell, numbers, tp, tz, tc, nm, num, sh = _setup(engine, seed=4)
    _, _, cn_search = _cn_block(engine, nm)
    # other species (the context attached to the buffer still holds the old ones: it is the fingerprint that notices)
    z2 = numbers.copy()
    z2[::3] = 1
    out = _d3(tp, _t(z2), p, nm, sh, tc[None])
    ref = _oracle(pos, z2, tables, nm, sh, cell)
    assert _close(out[2], ref[2]) and _close(out[0], ref[0]) and not torch.equal(out[2], cn_search)
    # another cell (shifted pairs change their distances)
    cell2 = cell * np.float32(1.01)
    out = _d3(tp, tz, p, nm, sh, _t(cell2)[None])
    ref = _oracle(pos, numbers, tables, nm, sh, cell2)
    assert _close(out[2], ref[2]) and _close(out[0], ref[0])
    # another k1
    out = _d3(tp, tz, p, nm, sh, tc[None], k1=12.0)
    ref = _oracle(pos, numbers, tables, nm, sh, cell, k1=12.0)
    assert _close(out[2], ref[2]) and _close(out[0], ref[0])
    # other radii
    from nvalchemiops.interactions.dispersion import D3Parameters
    t2 = dict(tables)
    t2["rcov"] = tables["rcov"] * np.float32(1.05)
    p2 = D3Parameters(rcov=_t(t2["rcov"]), r4r2=p.r4r2, c6ab=p.c6ab, cn_ref=p.cn_ref)
    out = _d3(tp, tz, p2, nm, sh, tc[None])
    ref = _oracle(pos, numbers, t2, nm, sh, cell)
    assert _close(out[2], ref[2]) and _close(out[0], ref[0])
    # and the original call still adopts
    out = _d3(tp, tz, p, nm, sh, tc[None])
    assert torch.equal(out[2], cn_search)


def test_overflowed_rows_raise_the_flag_and_the_ordinary_pass_runs(engine):
    tables, p = _params()
    pos, cell, _, numbers = S.fcc_box(2048, seed=5, dtype=np.float32)
    tp, tz, tc = _t(pos), _t(numbers), _t(cell)
    nm, num, sh = _search(engine, tp, 9.0, tc, tz, p.rcov, 64)  # rows hold 64 of ~190 neighbours
    assert int(num.max()) > 64
    flag, _, cn = _cn_block(engine, nm)
    assert flag != 0
    out = _d3(tp, tz, p, nm, sh, tc[None])
    ref = _oracle(pos, numbers, tables, nm, sh, cell)  # the oracle sums the 64 stored entries, as the reference does
    assert _close(out[2], ref[2]) and _close(out[0], ref[0])
    assert not _close(out[2], cn)  # (the search counted every hit)


@pytest.mark.parametrize("how", ["data", "raw_pointer", "storage_copy"])
def test_bulk_edits_behind_torchs_back_are_caught_on_the_device(engine, how, monkeypatch):
    """Writers that move no torch version counter.  The host-side record still says "valid"; the sampled device-side comparison (here at
    stride 1 = every row, and at the default stride for the bulk edit) sends every pass to the caller's arrays, and the answer is the
    oracle's for the list as it IS."""
    from nvalchemiops import _capi as C

    tables, p, pos, cell, numbers, tp, tz, tc, nm, num, sh = _setup(engine, seed=6)
    n, m = nm.shape
    v_nm, v_sh = nm._version, sh._version
    if how == "data":  # drop the second half of every row
        nm.data[:, m // 2:] = n
        sh.data[:, m // 2:, :] = 0
    elif how == "raw_pointer":  # the library's streaming fill over the first quarter of the matrix: every entry becomes `n` (= padding: rows emptied)
        import struct
        C.lib().mi_calibrate_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_float, ctypes.c_void_p]
        nbytes = (n // 4) * m * 4 // 16384 * 16384
        as_float = struct.unpack("f", struct.pack("i", n))[0]  # the float whose bit pattern is the integer n (a denormal: stored, never computed with)
        assert C.lib().mi_calibrate_fill(ctypes.c_void_p(nm.data_ptr()), ctypes.c_size_t(nbytes), ctypes.c_float(as_float), C.stream_of(nm)) == 0
        torch.cuda.synchronize()
        assert int(nm[0, 0]) == n and int(nm[n // 4 - 8, 5]) == n
    else:
        other = nm.clone()
        other[::2] = n  # every other row emptied
        nm.untyped_storage().copy_(other.untyped_storage())
    assert nm._version == v_nm and sh._version == v_sh
    assert engine.packed_companion(nm, sh, n) is not None  # the host cannot know
    for stride in (1, 64):
        monkeypatch.setattr(engine, "_VERIFY_STRIDE", stride)
        out = _d3(tp, tz, p, nm, sh, tc[None])
        ref = _oracle(pos, numbers, tables, nm, sh, cell)
        assert _close(out[2], ref[2]), (how, stride)
        assert _close(out[0], ref[0]) and _close(out[1], ref[1], scale=_fs(ref))


def test_single_entry_edit_needs_invalidate(engine, monkeypatch):
    """The documented limit of the sampled check: one entry edited through `tensor.data` is only seen by the call whose sample contains
    that row -- one call in `stride` (the sampled rows rotate: consecutive calls visit every row once) -- and the other calls answer for
    the unedited list.  `invalidate` is the contract for such writers; NVALCHEMIOPS_NL_PACKED_VERIFY=1 compares every row on every call."""
    tables, p, pos, cell, numbers, tp, tz, tc, nm, num, sh = _setup(engine, seed=7)
    n, m = nm.shape
    before = _d3(tp, tz, p, nm, sh, tc[None])
    nm.data[1000, 0] = n  # one neighbour of one atom removed behind torch's back
    ref = _oracle(pos, numbers, tables, nm, sh, cell)
    assert not _close(before[2], ref[2])
    monkeypatch.setattr(engine, "_VERIFY_STRIDE", 64)
    monkeypatch.setattr(engine, "_verify_calls", 0)
    seen = stale = 0
    for _ in range(64):  # phases 0 .. 63: exactly one of them samples row 1000
        out = _d3(tp, tz, p, nm, sh, tc[None])
        if torch.equal(out[2], before[2]):
            stale += 1
        else:
            assert _close(out[2], ref[2]) and _close(out[0], ref[0])
            seen += 1
    assert (seen, stale) == (1, 63)
    monkeypatch.setattr(engine, "_VERIFY_STRIDE", 1)  # every row, every call
    out = _d3(tp, tz, p, nm, sh, tc[None])
    assert _close(out[2], ref[2]) and _close(out[0], ref[0])
    monkeypatch.setattr(engine, "_VERIFY_STRIDE", 0)  # check off: the companion is believed
    out = _d3(tp, tz, p, nm, sh, tc[None])
    assert torch.equal(out[2], before[2])
    from nvalchemiops.neighborlist import invalidate
    invalidate(nm)
    out = _d3(tp, tz, p, nm, sh, tc[None])
    assert _close(out[2], ref[2]) and _close(out[0], ref[0])


def test_search_without_context_is_unchanged(engine):
    """No context attached: the companion carries no coordination numbers and the list is bit-identical to the context-carrying search's."""
    from nvalchemiops.neighborlist import cell_list

    tables, p, pos, cell, numbers, tp, tz, tc, nm, num, sh = _setup(engine, seed=8)
    nm0, num0, sh0 = cell_list(tp, 9.0, tc, torch.tensor([True] * 3, device=DEV), max_neighbors=nm.shape[1])
    assert getattr(nm0, engine._PACKED_ATTR).cn is None
    assert torch.equal(nm, nm0) and torch.equal(num, num0) and torch.equal(sh, sh0)
    a, b = _d3(tp, tz, p, nm, sh, tc[None]), _d3(tp, tz, p, nm0, sh0, tc[None])
    assert _close(a[2], b[2]) and _close(a[0], b[0])


def test_auto_policy_learns_the_species_from_dftd3(monkeypatch):
    """Policy "auto", untouched reference call sequence (no `attach_dftd3_context`, no `tuned_neighbor_buffers`): the first `dftd3` call on a
    matrix this package built registers shape and species; the next search into the same buffers -- or into freshly allocated outputs of
    the same shape -- carries companion AND coordination numbers, and `dftd3` adopts them (CN output = the search's block, bit for bit).
    Results stay the oracle's throughout."""
    from nvalchemiops.neighborlist import _engine as E
    from nvalchemiops.neighborlist import cell_list

    monkeypatch.setattr(E, "_PACKED_POLICY", "auto")
    monkeypatch.setattr(E, "_PACKED_WANTED", set())
    monkeypatch.setattr(E, "_D3CTX_BY_SHAPE", {})
    tables, p = _params()
    pos, cell, _, numbers = S.fcc_box(2048, seed=9, dtype=np.float32)
    tp, tz, tc = _t(pos), _t(numbers), _t(cell)
    pbc = torch.tensor([True] * 3, device=DEV)
    n, m = 2048, 256
    nm = torch.empty((n, m), dtype=torch.int32, device=DEV)
    sh = torch.empty((n, m, 3), dtype=torch.int32, device=DEV)
    num = torch.empty(n, dtype=torch.int32, device=DEV)
    cell_list(tp, 9.0, tc, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    assert not hasattr(nm, E._PACKED_ATTR)  # nobody asked yet
    ref = _oracle(pos, numbers, tables, nm, sh, cell)
    out0 = _d3(tp, tz, p, nm, sh, tc[None])  # learns shape + species
    assert _close(out0[2], ref[2]) and _close(out0[0], ref[0])
    cell_list(tp, 9.0, tc, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    rec = getattr(nm, E._PACKED_ATTR)
    assert rec.cn is not None
    out1 = _d3(tp, tz, p, nm, sh, tc[None])
    assert torch.equal(out1[2], rec.cn[1024:].view(torch.float32)) and _close(out1[2], ref[2]) and _close(out1[0], ref[0])
    # outputs allocated by the search itself: the shape remembers
    nm2, num2, sh2 = cell_list(tp, 9.0, tc, pbc, max_neighbors=m)
    rec2 = getattr(nm2, E._PACKED_ATTR)
    assert rec2.cn is not None
    out2 = _d3(tp, tz, p, nm2, sh2, tc[None])
    assert torch.equal(out2[2], rec2.cn[1024:].view(torch.float32)) and _close(out2[0], ref[0])
    # other species on the next call: the stale context's numbers are not adopted, the context is replaced
    z2 = numbers.copy()
    z2[::2] = 1
    out3 = _d3(tp, _t(z2), p, nm2, sh2, tc[None])
    ref3 = _oracle(pos, z2, tables, nm2, sh2, cell)
    assert _close(out3[2], ref3[2]) and _close(out3[0], ref3[0])

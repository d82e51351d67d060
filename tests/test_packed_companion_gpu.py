"""The packed companion of a padded neighbour matrix (round 5; `mi_nl_neighbors_packed` -> `mi_d3_packed`).

The neighbour search can leave, next to the API-format matrix, the 4 B/slot word list the D3 passes stream; `dftd3` uses it only while
matrix and shifts are provably the tensors the search wrote.  What is checked here, all through the public API and the C ABI behind it:
the word format against the matrix it describes, bit-identical D3 outputs with and without the companion (plain, spatial order, batch,
out-of-range shifts -> device-side fallback), that an edited matrix is never served from a stale companion (answers = the oracle's on the
EDITED list), and the "auto" policy (learned from a first dftd3 call, never paid by a caller who only builds lists)."""
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


def _expected_words(nm, num, sh):
    """The companion's contract, restated with torch: index | (shift + 1) fields for the stored entries, all ones for padding."""
    n, m = nm.shape
    slot = torch.arange(m, device=nm.device)[None, :]
    valid = slot < num.clamp(max=m)[:, None]
    s1 = sh.long() + 1
    code = (s1[..., 0] << 26) | (s1[..., 1] << 28) | (s1[..., 2] << 30)
    w = (nm.long() | code) & 0xFFFFFFFF
    return torch.where(valid, w, torch.full_like(w, 0xFFFFFFFF))


def _words_of(E, nm):
    rec = getattr(nm, E._PACKED_ATTR)
    raw = rec.words.view(torch.int32)
    flag = int(raw[0])
    words = raw[64:64 + nm.numel()].view(nm.shape).long() & 0xFFFFFFFF
    return flag, words


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,cutoff,m", [(4000, 11.0, 384), (300, 6.0, 96)])  # dense cells (tiled kernel) and sparse cells (wave-per-atom kernel)
def test_companion_words_describe_the_matrix(engine, dtype, n, cutoff, m):
    from nvalchemiops.neighborlist import cell_list

    pos, cell, _, _ = S.fcc_box(n, dtype=dtype)
    nm, num, sh = cell_list(_t(pos), cutoff, _t(cell), torch.tensor([True] * 3, device=DEV), max_neighbors=m)
    assert int(num.max()) <= m
    flag, words = _words_of(engine, nm)
    assert flag == 0
    assert torch.equal(words, _expected_words(nm, num, sh))
    # and the list itself is what the search writes without a companion
    engine._PACKED_POLICY = "0"
    nm0, num0, sh0 = cell_list(_t(pos), cutoff, _t(cell), torch.tensor([True] * 3, device=DEV), max_neighbors=m)
    assert not hasattr(nm0, engine._PACKED_ATTR)
    assert torch.equal(nm, nm0) and torch.equal(num, num0) and torch.equal(sh, sh0)


def _d3(pos, z, p, nm, sh, cell, bi=None, nsys=None):
    from nvalchemiops.interactions.dispersion import dftd3

    kw = dict(d3_params=p, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=cell, compute_virial=True, **FP)
    if bi is not None:
        kw.update(batch_idx=bi, num_systems=nsys)
    return dftd3(pos, z, **kw)


@pytest.mark.parametrize("sort", ["0", "1"])
def test_d3_with_and_without_the_companion_is_bit_identical(engine, sort, monkeypatch):
    from nvalchemiops.neighborlist import cell_list

    tables, p = _params()
    n = 4000
    pos, cell, _, numbers = S.fcc_box(n, dtype=np.float32)
    perm = np.random.default_rng(5).permutation(n)  # incoherent numbering: the spatial order has something to do
    pos, numbers = pos[perm], numbers[perm]
    tp, tz, tc = _t(pos), _t(numbers), _t(cell)
    nm, num, sh = cell_list(tp, 11.0, tc, torch.tensor([True] * 3, device=DEV), max_neighbors=384)
    assert int(num.max()) <= 384
    assert hasattr(nm, engine._PACKED_ATTR) and engine.packed_companion(nm, sh, n) is not None
    monkeypatch.setenv("NVALCHEMIOPS_D3_SORT", sort)
    with_pk = _d3(tp, tz, p, nm, sh, tc[None])
    with_pk2 = _d3(tp, tz, p, nm, sh, tc[None])  # the companion is read-only: a second call sees the same words
    plain = _d3(tp, tz, p, nm.clone(), sh.clone(), tc[None])  # clones carry no companion
    for a, b, c in zip(with_pk, plain, with_pk2):
        assert torch.equal(a, b) and torch.equal(a, c)
    with O.d3_wide_sums():
        ref = O.dftd3(pos, numbers, tables, neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy(), cell=cell, compute_virial=True, **FP)
    assert abs(float(with_pk[0][0]) - float(ref[0][0])) <= 2e-6 * abs(float(ref[0][0])) + 1e-6
    assert np.abs(with_pk[1].cpu().numpy() - ref[1]).max() <= 1e-5 * np.abs(ref[1]).max() + 1e-6


def test_an_edited_matrix_is_never_served_from_a_stale_companion(engine):
    """Between the search and dftd3 the caller removes a neighbour (in-place torch ops on matrix and shifts): the version counters move, the
    companion is dropped, and the answer is the oracle's for the EDITED list -- not the one the companion still describes."""
    from nvalchemiops.neighborlist import cell_list

    tables, p = _params()
    pos, cell, _, numbers = S.fcc_box(500, dtype=np.float32)
    tp, tz, tc = _t(pos), _t(numbers), _t(cell)
    nm, num, sh = cell_list(tp, 10.0, tc, torch.tensor([True] * 3, device=DEV), max_neighbors=256)
    before = _d3(tp, tz, p, nm, sh, tc[None])
    assert engine.packed_companion(nm, sh, 500) is not None
    nm[:, 0] = 500  # every atom loses its first neighbour (fill_value = n = padding)
    assert engine.packed_companion(nm, sh, 500) is None
    after = _d3(tp, tz, p, nm, sh, tc[None])
    assert not torch.equal(before[0], after[0])
    fresh = _d3(tp, tz, p, nm.clone(), sh.clone(), tc[None])
    for a, b in zip(after, fresh):
        assert torch.equal(a, b)
    with O.d3_wide_sums():
        ref = O.dftd3(pos, numbers, tables, neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy(), cell=cell, compute_virial=True, **FP)
    assert abs(float(after[0][0]) - float(ref[0][0])) <= 2e-6 * abs(float(ref[0][0])) + 1e-6
    # editing only the shifts invalidates it as well
    nm2, num2, sh2 = cell_list(tp, 10.0, tc, torch.tensor([True] * 3, device=DEV), max_neighbors=256)
    assert engine.packed_companion(nm2, sh2, 500) is not None
    sh2.add_(0)
    assert engine.packed_companion(nm2, sh2, 500) is None
    # a different shifts tensor (even an equal one) is not the tensor the companion was built with
    nm3, num3, sh3 = cell_list(tp, 10.0, tc, torch.tensor([True] * 3, device=DEV), max_neighbors=256)
    assert engine.packed_companion(nm3, sh3.clone(), 500) is None
    # ... nor is another fill_value the index limit it was built for
    assert engine.packed_companion(nm3, sh3, 400) is None


def test_shifts_outside_the_packed_range_fall_back_on_the_device(engine):
    """A cell much smaller than the cutoff stores unit shifts of +-2: the search raises the companion's flag, `mi_d3_packed`'s plain launch
    does the work, outputs bit-identical to the no-companion call."""
    from nvalchemiops.neighborlist import cell_list

    _, p = _params()
    pos, cell = S.random_box(12, 7.0, seed=9, dtype=np.float32, triclinic=True)
    z = np.random.default_rng(2).choice(np.array([1, 6, 8, 17], np.int32), 12)
    tp, tz, tc = _t(pos), _t(z), _t(cell)
    nm, num, sh = cell_list(tp, 14.0, tc, torch.tensor([True] * 3, device=DEV), max_neighbors=448)
    assert int(sh.abs().max()) >= 2 and int(num.max()) <= 448
    flag, _ = _words_of(engine, nm)
    assert flag != 0
    a = _d3(tp, tz, p, nm, sh, tc[None])
    b = _d3(tp, tz, p, nm.clone(), sh.clone(), tc[None])
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_batch_companion(engine):
    from nvalchemiops.neighborlist import batch_cell_list

    _, p = _params()
    parts = [S.fcc_box(864, seed=3 + b, dtype=np.float32) for b in range(3)]
    pos = np.concatenate([q[0] for q in parts])
    cell = np.stack([q[1] for q in parts])
    z = np.concatenate([q[3] for q in parts])
    bi = np.repeat(np.arange(3, dtype=np.int32), 864)
    tp, tz, tc, tb = _t(pos), _t(z), _t(cell), _t(bi)
    pbc = torch.ones((3, 3), dtype=torch.bool, device=DEV)
    nm, num, sh = batch_cell_list(tp, 12.0, tc, pbc, tb, max_neighbors=512)
    assert int(num.max()) <= 512 and engine.packed_companion(nm, sh, len(pos)) is not None
    flag, words = _words_of(engine, nm)
    assert flag == 0 and torch.equal(words, _expected_words(nm, num, sh))
    a = _d3(tp, tz, p, nm, sh, tc, tb, 3)
    b = _d3(tp, tz, p, nm.clone(), sh.clone(), tc, tb, 3)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_auto_policy_learns_from_dftd3_and_reuses_the_storage(monkeypatch):
    """Default policy: a list-only caller never gets (or pays for) a companion; once dftd3 has been handed a matrix of some shape that this
    package built, the next search of that shape emits one -- into the same storage when the caller searches into the same buffers."""
    from nvalchemiops.neighborlist import _engine as E
    from nvalchemiops.neighborlist import cell_list

    monkeypatch.setattr(E, "_PACKED_POLICY", "auto")
    monkeypatch.setattr(E, "_PACKED_WANTED", set())
    monkeypatch.setattr(E, "_D3CTX_BY_SHAPE", {})
    _, p = _params()
    pos, cell, _, numbers = S.fcc_box(2048, dtype=np.float32)
    tp, tz, tc = _t(pos), _t(numbers), _t(cell)
    pbc = torch.tensor([True] * 3, device=DEV)
    nm = torch.empty((2048, 384), dtype=torch.int32, device=DEV)
    sh = torch.empty((2048, 384, 3), dtype=torch.int32, device=DEV)
    num = torch.empty(2048, dtype=torch.int32, device=DEV)
    for _ in range(2):
        cell_list(tp, 11.0, tc, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
        assert not hasattr(nm, E._PACKED_ATTR)  # nobody asked
    first = _d3(tp, tz, p, nm, sh, tc[None])  # step 1 of an MD loop: ordinary path, and the shape is noted
    cell_list(tp, 11.0, tc, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    rec = getattr(nm, E._PACKED_ATTR)
    second = _d3(tp, tz, p, nm, sh, tc[None])
    # round 6: the first dftd3 call also taught the policy the SPECIES, so this search summed the coordination numbers and dftd3 adopted
    # them -- another summation order for CN (<= 1e-6 relative, tests/test_search_cn_gpu.py), everything else follows within its own bar
    assert rec.cn is not None and torch.equal(second[2], rec.cn[1024:].view(torch.float32))
    for a, b, scale in zip(first, second, (1.0, 5.0 * float(first[1].abs().max()), 1.0, 0.2 * float(first[3].abs().max()))):
        assert bool(((a - b).abs() <= 1e-6 + 1e-6 * b.abs() + 1e-6 * scale).all())
    cell_list(tp, 11.0, tc, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    assert getattr(nm, E._PACKED_ATTR).words.data_ptr() == rec.words.data_ptr() and getattr(nm, E._PACKED_ATTR).cn.data_ptr() == rec.cn.data_ptr()
    third = _d3(tp, tz, p, nm, sh, tc[None])
    for a, b in zip(second, third):  # the fused path is deterministic: same buffers, same inputs, same bits
        assert torch.equal(a, b)
    # a half-filled or shift-less search into the same buffers drops it
    cell_list(tp, 11.0, tc, pbc, half_fill=True, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    assert not hasattr(nm, E._PACKED_ATTR)


def test_tuned_neighbor_buffers(monkeypatch):
    """`tuned_neighbor_buffers`: candidate buffer sets timed by a trial search, the fastest kept.  Whatever set wins, a search into it gives
    the list a search into plain `torch.empty` buffers gives; with `for_dftd3` the searches emit the companion from the first one on."""
    from nvalchemiops.neighborlist import _engine as E
    from nvalchemiops.neighborlist import cell_list, tuned_neighbor_buffers

    monkeypatch.setattr(E, "_PACKED_POLICY", "auto")
    monkeypatch.setattr(E, "_PACKED_WANTED", set())
    n, m, rc = 24000, 768, 13.0  # 16 n m = 295 MB: above the size where the choice is made by measurement
    pos, cell, _, _ = S.fcc_box(n, dtype=np.float32)
    tp, tc = _t(pos), _t(cell)
    pbc = torch.tensor([True] * 3, device=DEV)
    rep = {}
    nm, sh, num = tuned_neighbor_buffers(tp, rc, tc, pbc, m, candidates=3, for_dftd3=True, report=rep)
    assert nm.shape == (n, m) and sh.shape == (n, m, 3) and num.shape == (n,)
    assert rep["candidates"] == 3 and len(rep["trial_ms"]) == 3 and 0 <= rep["chosen"] < 3 and rep["trial_ms"][rep["chosen"]] == min(rep["trial_ms"])
    cell_list(tp, rc, tc, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    assert int(num.max()) <= m and E.packed_companion(nm, sh, n) is not None
    nm0, num0, sh0 = cell_list(tp, rc, tc, pbc, max_neighbors=m)
    assert torch.equal(nm, nm0) and torch.equal(sh, sh0) and torch.equal(num, num0)
    # a small list: nothing to choose
    rep = {}
    a, b, c = tuned_neighbor_buffers(tp[:500], 6.0, tc, pbc, 64, report=rep)
    assert a.shape == (500, 64) and rep["candidates"] == 1


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_companion_on_awkward_systems(engine, seed):
    """Triclinic cells, atoms far outside the cell (per-atom wraps: the searches' mixed-shift path), rows that overflow the matrix, a batch
    with a non-periodic axis: wherever the stored shifts fit the packed word the companion equals the matrix word for word and its flag is
    clear, otherwise the flag is raised; either way D3 with the companion equals D3 without it bit for bit."""
    from nvalchemiops.neighborlist import batch_cell_list, cell_list

    _, p = _params()
    g = np.random.default_rng(100 + seed)
    n = int(g.integers(200, 900))
    box = float(g.uniform(9.0, 16.0))
    pos, cell = S.random_box(n, box, seed=seed, dtype=np.float32, triclinic=bool(seed % 2), outside=seed >= 3)
    z = g.choice(np.array([1, 6, 8, 17], np.int32), n)
    rc = float(g.uniform(4.0, 7.0))
    m = 48 if seed == 4 else 160  # seed 4: rows overflow (the search keeps counting past the row width)
    tp, tz, tc = _t(pos), _t(z), _t(cell)
    pbc = torch.tensor([True, True, seed != 2], device=DEV)
    nm, num, sh = cell_list(tp, rc, tc, pbc, max_neighbors=m)
    flag, words = _words_of(engine, nm)
    fits = int(sh.abs().max()) <= 1
    overflow = int(num.max()) > m  # (a hit past the row width is not stored, but a shift it carries may still raise the flag: conservative)
    assert (flag != 0) if not fits else (flag == 0 or overflow), (flag, int(sh.abs().max()), overflow)
    if fits and flag == 0:
        assert torch.equal(words, _expected_words(nm, num, sh))
    if int(num.max()) <= m:  # D3 needs a full list
        a = _d3(tp, tz, p, nm, sh, tc[None])
        b = _d3(tp, tz, p, nm.clone(), sh.clone(), tc[None])
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    # the same atoms as two systems of a batch
    half = n // 2
    bi = _t(np.repeat(np.arange(2, dtype=np.int32), [half, n - half]))
    cells = torch.stack([tc, tc * 1.1])
    pbc2 = torch.tensor([[True, True, True], [True, seed != 2, True]], device=DEV)
    nm2, num2, sh2 = batch_cell_list(tp, rc, cells, pbc2, bi, max_neighbors=m)
    flag2, words2 = _words_of(engine, nm2)
    if int(sh2.abs().max()) > 1:
        assert flag2 != 0
    elif flag2 == 0:
        assert torch.equal(words2, _expected_words(nm2, num2, sh2))
    else:
        assert int(num2.max()) > m
    if int(num2.max()) <= m:
        a = _d3(tp, tz, p, nm2, sh2, cells, bi, 2)
        b = _d3(tp, tz, p, nm2.clone(), sh2.clone(), cells, bi, 2)
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_inference_mode_tensors_simply_take_the_ordinary_path(engine):
    """Tensors created under `torch.inference_mode()` carry no version counter, so nothing could vouch for a companion: the search attaches
    none, and dftd3 gives the result it gives for ordinary tensors."""
    from nvalchemiops.neighborlist import cell_list

    _, p = _params()
    pos, cell, _, numbers = S.fcc_box(500, dtype=np.float32)
    pbc = torch.tensor([True] * 3, device=DEV)
    tp, tz, tc = _t(pos), _t(numbers), _t(cell)
    with torch.inference_mode():
        nm, num, sh = cell_list(tp, 10.0, tc, pbc, max_neighbors=256)
        assert not hasattr(nm, engine._PACKED_ATTR)
        a = _d3(tp, tz, p, nm, sh, tc[None])
    nm2, num2, sh2 = cell_list(tp, 10.0, tc, pbc, max_neighbors=256)
    b = _d3(tp, tz, p, nm2, sh2, tc[None])
    assert torch.equal(nm, nm2) and torch.equal(sh, sh2)
    for x, y in zip(a, b):
        assert torch.equal(x, y)

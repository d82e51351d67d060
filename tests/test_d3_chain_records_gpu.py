"""Chain records of the D3 chain-rule pass (round 6, csrc/d3.hip `D3CRec`): the energy pass leaves one 16-byte record per atom -- position +
dE/dCN with the compact species id in spare bits of that float -- and the chain pass gathers ONE record per neighbour instead of a record
and a dE/dCN value.  What must hold:

* <= 7 species: the id sits in exponent bits only; the chain pass adds the numbers it added before, bit for bit (on = off);
* 8 - 15 species: dE/dCN_j loses its lowest mantissa bit (half an fp32 ulp): inside the reference bar against the oracle, next to off;
* >= 16 species, |dE/dCN| >= 256 Ha, the packed list found unusable: the two-gather walk (on = off, bit for bit);
* padding atoms (Z = 0), CSR lists, batches, the spatial order: all through the same records.
`NVALCHEMIOPS_D3_CHAIN_RECORDS=0` is the A/B switch (read on every call)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import systems as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FP = dict(a1=0.4, a2=4.0, s8=0.8, k1=16.0, k3=-4.0, s6=1.0)


@pytest.fixture(autouse=True)
def _fixed_companion_policy(monkeypatch):
    """on / off runs must see the same list machinery: under the "auto" policy the second search into a shape `dftd3` has seen writes the packed
    companion and sums the coordination numbers itself (another summation order: CN equal to 1e-6, not bit for bit)."""
    from nvalchemiops.neighborlist import _engine as E

    monkeypatch.setattr(E, "_PACKED_POLICY", "0")


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def _params(zmax=24, seed=3):
    from nvalchemiops.interactions.dispersion import D3Parameters

    t = O.d3_test_tables(zmax, seed)
    return t, D3Parameters(rcov=_t(t["rcov"]), r4r2=_t(t["r4r2"]), c6ab=_t(t["c6ab"]), cn_ref=_t(t["cn_ref"]))


def _wide(*args, **kw):
    with O.d3_wide_sums():
        return O.dftd3(*args, **kw)


def _on_off(monkeypatch, fn):
    monkeypatch.setenv("NVALCHEMIOPS_D3_CHAIN_RECORDS", "0")
    off = fn()
    monkeypatch.delenv("NVALCHEMIOPS_D3_CHAIN_RECORDS")
    on = fn()
    return on, off


def _box(n_species, n=600, box=34.0, seed=5, with_padding=True):
    pos, cell = S.random_box(n, box, seed=seed, dtype=np.float32, triclinic=True)
    z = (np.random.default_rng(seed + 1).integers(0, n_species, n) + 1).astype(np.int32)
    if with_padding:
        z[::37] = 0  # padding atoms: no energy, no force, nobody's neighbour
    return pos, cell, z


@pytest.mark.parametrize("n_species", [1, 2, 3, 4, 7])
@pytest.mark.parametrize("fmt", ["matrix", "csr"])
def test_up_to_seven_species_bit_identical(n_species, fmt, monkeypatch):
    from nvalchemiops.interactions.dispersion import dftd3
    from nvalchemiops.neighborlist import cell_list

    t, p = _params()
    pos, cell, z = _box(n_species)
    pbc = torch.tensor([True] * 3, device=DEV)
    if fmt == "matrix":
        nm, num, sh = cell_list(_t(pos), 14.0, _t(cell), pbc, max_neighbors=384)
        assert int(num.max()) <= 384
        kw = dict(neighbor_matrix=nm, neighbor_matrix_shifts=sh)
        okw = dict(neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy())
    else:
        lst, nptr, lsh = cell_list(_t(pos), 14.0, _t(cell), pbc, return_neighbor_list=True)
        kw = dict(neighbor_list=lst, neighbor_ptr=nptr, unit_shifts=lsh)
        okw = dict(idx_j=lst[1].cpu().numpy(), neighbor_ptr=nptr.cpu().numpy(), unit_shifts=lsh.cpu().numpy())
    on, off = _on_off(monkeypatch, lambda: dftd3(_t(pos), _t(z), d3_params=p, cell=_t(cell)[None], compute_virial=True, **kw, **FP))
    for a, b in zip(on, off):
        assert torch.equal(a, b)
    assert float(on[1][::37].abs().max()) == 0.0  # padding atoms keep zero forces
    ref = _wide(pos, z, t, cell=cell, compute_virial=True, **okw, **FP)
    got = on[1].cpu().numpy()
    assert np.abs(got - ref[1]).max() <= 1e-6 + 1e-6 * np.abs(ref[1]).max() + 5e-6 * np.abs(ref[1]).max()


@pytest.mark.parametrize("n_species", [8, 12, 15])
def test_eight_to_fifteen_species_within_the_reference_bar(n_species, monkeypatch):
    from nvalchemiops.interactions.dispersion import dftd3
    from nvalchemiops.neighborlist import cell_list

    t, p = _params()
    pos, cell, z = _box(n_species, seed=11)
    pbc = torch.tensor([True] * 3, device=DEV)
    nm, num, sh = cell_list(_t(pos), 14.0, _t(cell), pbc, max_neighbors=384)
    assert int(num.max()) <= 384
    on, off = _on_off(monkeypatch, lambda: dftd3(_t(pos), _t(z), d3_params=p, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=_t(cell)[None],
                                                  compute_virial=True, **FP))
    assert torch.equal(on[0], off[0]) and torch.equal(on[2], off[2])  # energy and CN do not pass through the records
    f_on, f_off = on[1].cpu().numpy().astype(np.float64), off[1].cpu().numpy().astype(np.float64)
    scale = np.abs(f_off).max()
    assert np.abs(f_on - f_off).max() <= 2e-7 * scale  # half an ulp of dE/dCN_j per term
    ref = _wide(pos, z, t, neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy(), cell=cell, compute_virial=True, **FP)
    for got in (f_on, f_off):
        assert np.abs(got - ref[1]).max() <= 1e-6 + 6e-6 * np.abs(ref[1]).max()
    v_on, v_off = on[3].cpu().numpy().astype(np.float64), off[3].cpu().numpy().astype(np.float64)
    assert np.abs(v_on - v_off).max() <= 1e-6 * np.abs(v_off).max()


@pytest.mark.parametrize("n_species", [16, 22])
def test_sixteen_or_more_species_take_the_two_gather_walk(n_species, monkeypatch):
    from nvalchemiops.interactions.dispersion import dftd3
    from nvalchemiops.neighborlist import cell_list

    t, p = _params()
    pos, cell, z = _box(n_species, seed=13)
    pbc = torch.tensor([True] * 3, device=DEV)
    nm, num, sh = cell_list(_t(pos), 14.0, _t(cell), pbc, max_neighbors=384)
    on, off = _on_off(monkeypatch, lambda: dftd3(_t(pos), _t(z), d3_params=p, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=_t(cell)[None],
                                                  compute_virial=True, **FP))
    for a, b in zip(on, off):
        assert torch.equal(a, b)


def test_huge_dE_dCN_raises_the_flag(monkeypatch):
    """s6 = s8 = 1e7: |dE/dCN| leaves the window of the records (>= 256 Ha); the energy pass raises the flag and the chain pass gathers
    dE/dCN from its own array as before -- bit-identical, finite."""
    from nvalchemiops.interactions.dispersion import dftd3
    from nvalchemiops.neighborlist import cell_list

    t, p = _params()
    pos, cell, z = _box(3, seed=17)
    pbc = torch.tensor([True] * 3, device=DEV)
    nm, num, sh = cell_list(_t(pos), 14.0, _t(cell), pbc, max_neighbors=384)
    fp = dict(FP, s6=1.0e7, s8=1.0e7)
    on, off = _on_off(monkeypatch, lambda: dftd3(_t(pos), _t(z), d3_params=p, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=_t(cell)[None],
                                                  compute_virial=True, **fp))
    # the window really is left (otherwise this test tests nothing): the energy is ~1e7 x an ordinary one
    assert abs(float(on[0][0])) > 1.0e4
    for a, b in zip(on, off):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    # ... and a later ordinary call on the same stream is not affected by the flag of this one (cleared per call)
    on2, off2 = _on_off(monkeypatch, lambda: dftd3(_t(pos), _t(z), d3_params=p, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=_t(cell)[None],
                                                    compute_virial=True, **FP))
    for a, b in zip(on2, off2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("policy", ["0", "1"])
@pytest.mark.parametrize("batched", [False, True])
def test_spatial_order_and_batches(batched, policy, monkeypatch):
    """Randomly numbered atoms with the spatial order forced on (records indexed by PLACE) and off (by atom): on = off with and without
    chain records; a batch of two systems as well."""
    from nvalchemiops.interactions.dispersion import dftd3
    from nvalchemiops.neighborlist import batch_cell_list, cell_list

    from nvalchemiops.neighborlist import _engine as E

    monkeypatch.setattr(E, "_PACKED_POLICY", policy)  # "1": every search writes the packed companion the D3 passes then walk
    t, p = _params()
    n = 4000
    pos, cell, _, numbers = S.fcc_box(n, dtype=np.float32)
    z = np.where(numbers == 6, 6, 8).astype(np.int32)
    z[::53] = 1
    perm = np.random.default_rng(5).permutation(n)
    pos, z = pos[perm], z[perm]
    pbc = torch.tensor([True] * 3, device=DEV)

    def run():
        if batched:
            P = np.concatenate([pos, pos]); Z = np.concatenate([z, z])
            bi = _t(np.repeat(np.arange(2, dtype=np.int32), n)); C = _t(np.stack([cell, cell]))
            nm, num, sh = batch_cell_list(_t(P), 12.0, C, pbc[None].expand(2, 3).contiguous(), bi, max_neighbors=512)
            return dftd3(_t(P), _t(Z), d3_params=p, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=C, batch_idx=bi, compute_virial=True, **FP)
        nm, num, sh = cell_list(_t(pos), 12.0, _t(cell), pbc, max_neighbors=512)
        return dftd3(_t(pos), _t(z), d3_params=p, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=_t(cell)[None], compute_virial=True, **FP)

    res = {}
    for sort in ("0", "1"):
        monkeypatch.setenv("NVALCHEMIOPS_D3_SORT", sort)
        res[sort] = _on_off(monkeypatch, run)
        res[sort + "again"] = _on_off(monkeypatch, run)  # (the second sorted call builds its grid from the first one's measured cutoff)
    base = res["0"][1]
    for key, (on, off) in res.items():
        for a, b, c in zip(on, off, base):
            assert torch.equal(a, c) and torch.equal(b, c), key


def test_packed_list_fallback_keeps_the_two_gather_walk(monkeypatch):
    """A box smaller than the cutoff: unit shifts of +-2 make the packed list unusable (device-side flag); the energy pass's fallback launch
    walks the caller's arrays, the chain pass must not take records indexed for the packed walk.  With the spatial order on, too."""
    from nvalchemiops.interactions.dispersion import dftd3
    from nvalchemiops.neighborlist import cell_list

    t, p = _params()
    n = 2304
    pos, cell = S.random_box(n, 30.0, seed=3, dtype=np.float32)
    z = np.random.default_rng(4).choice(np.array([1, 6, 8], np.int32), n)
    pbc = torch.tensor([True] * 3, device=DEV)
    nm, num, sh = cell_list(_t(pos), 32.0, _t(cell), pbc, max_neighbors=12800)
    assert int(num.max()) <= 12800 and int(sh.abs().max()) == 2
    args = dict(d3_params=p, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=_t(cell)[None], compute_virial=True, **FP)
    outs = []
    for sort in ("0", "1"):
        monkeypatch.setenv("NVALCHEMIOPS_D3_SORT", sort)
        outs += list(_on_off(monkeypatch, lambda: dftd3(_t(pos), _t(z), **args)))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.isfinite(a).all() and torch.equal(a, b)


def test_molecules_without_a_cell(monkeypatch):
    """Non-periodic padded matrix and CSR (config 3's shape: no packed list at all): records indexed by atom."""
    from nvalchemiops.interactions.dispersion import dftd3

    t, p = _params(17, None)
    mol, numbers, _ = S.molecule(96, seed=21)
    n = 96
    nm = np.array([[j for j in range(n) if j != i] + [n] * 3 for i in range(n)], np.int32)
    kw = dict(a1=FP["a1"], a2=FP["a2"], s8=FP["s8"], d3_params=p)
    on, off = _on_off(monkeypatch, lambda: dftd3(_t(mol), _t(numbers), neighbor_matrix=_t(nm), **kw))
    for a, b in zip(on, off):
        assert torch.equal(a, b)
    ref = _wide(mol, numbers, t, neighbor_matrix=nm, **{k: v for k, v in kw.items() if k != "d3_params"})
    assert np.abs(on[1].cpu().numpy() - ref[1]).max() <= 1e-6 + 6e-6 * np.abs(ref[1]).max()

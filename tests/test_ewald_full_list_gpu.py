"""`mi_ewald_real_listed` (round 6): the real-space sum on a padded matrix that is provably the unmodified output of a full search skips its
per-entry symmetry checksums -- the search's own counts, the tensors' version counters and a sampled mirror look-up stand in for them.
What must hold: identical results on symmetric lists; the general (scatter) result whenever the list is NOT symmetric -- overflowed rows,
edits through torch, bulk edits behind torch's back -- each checked against the checksum path on the same arrays and against the oracle
(reference semantics: every stored entry scatters to both ends, ewald_kernels.py:518-544, :864-873)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def _system(n, dtype, seed=0, box=12.0):
    rng = np.random.default_rng(seed)
    cell = (np.eye(3) * box + 0.8 * rng.standard_normal((3, 3))).astype(dtype)
    pos = (rng.random((n, 3)) @ cell).astype(dtype)
    q = rng.standard_normal(n).astype(dtype)
    q -= q.mean()
    return pos, cell, q


def _run(pos, q, cell, nm, sh, trust, monkeypatch, n):
    from nvalchemiops.interactions.electrostatics import ewald as EW
    from nvalchemiops.interactions.electrostatics import ewald_real_space

    monkeypatch.setattr(EW, "_TRUST_FULL_LISTS", trust)
    alpha = torch.tensor([0.4], dtype=_t(pos).dtype, device=DEV)
    return ewald_real_space(_t(pos), _t(q), _t(cell)[None], alpha, neighbor_matrix=nm, neighbor_matrix_shifts=sh, mask_value=n,
                            compute_forces=True, compute_charge_gradients=True)


def _oracle(pos, q, cell, nm, sh, n):
    return O.ewald_real_space(pos, q, cell, 0.4, neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy(), mask_value=n,
                              compute_forces=True, compute_charge_gradients=True)


def _same(a, b, dtype, what, ref_scale):
    tol = 1e-11 if dtype == np.float64 else 2e-4
    for x, y, w in zip(a, b, ("energies", "forces", "charge gradients")):
        y = y if isinstance(y, torch.Tensor) else _t(np.asarray(y))
        err = float((x.double() - y.double().to(x.device)).abs().max())
        assert err <= tol * max(ref_scale, 1.0), (what, w, err)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_full_list_takes_the_trusted_form_and_gives_the_same_numbers(dtype, monkeypatch):
    from nvalchemiops.neighborlist import _engine as E
    from nvalchemiops.neighborlist import cell_list

    n = 250
    pos, cell, q = _system(n, dtype, seed=3)
    pbc = torch.tensor([True] * 3, device=DEV)
    nm, num, sh = cell_list(_t(pos), 7.0, _t(cell), pbc, max_neighbors=320)
    assert int(num.max()) <= 320 and E.full_list_counts(nm, sh) is num
    trusted = _run(pos, q, cell, nm, sh, True, monkeypatch, n)
    hashed = _run(pos, q, cell, nm, sh, False, monkeypatch, n)
    for a, b in zip(trusted, hashed):
        assert torch.equal(a, b)  # same pairs, same order per lane: bit-identical
    ref = _oracle(pos, q, cell, nm, sh, n)
    _same(trusted, ref, dtype, "symmetric list vs oracle", float(np.abs(ref[1]).max()))
    # a half-filled search leaves no record: nothing to trust
    hm, hnum, hsh = cell_list(_t(pos), 7.0, _t(cell), pbc, max_neighbors=320, half_fill=True)
    assert E.full_list_counts(hm, hsh) is None


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_overflowed_rows_raise_the_mark(dtype, monkeypatch):
    """max_neighbors below the fullest row: the search truncates, its counts say so, the trusted kernel marks the call and the scatter path
    answers -- the reference's result for the truncated arrays."""
    from nvalchemiops.neighborlist import _engine as E
    from nvalchemiops.neighborlist import cell_list

    n = 250
    pos, cell, q = _system(n, dtype, seed=5)
    pbc = torch.tensor([True] * 3, device=DEV)
    nm, num, sh = cell_list(_t(pos), 7.0, _t(cell), pbc, max_neighbors=40)
    assert int(num.max()) > 40 and E.full_list_counts(nm, sh) is num
    trusted = _run(pos, q, cell, nm, sh, True, monkeypatch, n)
    ref = _oracle(pos, q, cell, nm, sh, n)
    scale = float(np.abs(ref[1]).max()) * (10 if dtype == np.float32 else 1)
    _same(trusted, ref, dtype, "truncated rows vs oracle", scale)
    _same(trusted, _run(pos, q, cell, nm, sh, False, monkeypatch, n), dtype, "truncated rows vs checksum path", scale)


def test_an_edit_through_torch_kills_the_record(monkeypatch):
    from nvalchemiops.neighborlist import _engine as E
    from nvalchemiops.neighborlist import cell_list

    n, dtype = 250, np.float64
    pos, cell, q = _system(n, dtype, seed=7)
    pbc = torch.tensor([True] * 3, device=DEV)
    nm, num, sh = cell_list(_t(pos), 7.0, _t(cell), pbc, max_neighbors=320)
    assert E.full_list_counts(nm, sh) is num
    nm[3, 0] = (int(nm[3, 0]) + 1) % n  # in-place torch op: the version counter moves, the list is no longer symmetric
    assert E.full_list_counts(nm, sh) is None
    out = _run(pos, q, cell, nm, sh, True, monkeypatch, n)
    ref = _oracle(pos, q, cell, nm, sh, n)
    _same(out, ref, dtype, "edited entry vs oracle", float(np.abs(ref[1]).max()))
    # ... and so does an edit of the shifts or of the counts, or invalidate()
    nm2, num2, sh2 = cell_list(_t(pos), 7.0, _t(cell), pbc, max_neighbors=320)
    sh2[0, 0, 0] += 0
    assert E.full_list_counts(nm2, sh2) is None
    nm3, num3, sh3 = cell_list(_t(pos), 7.0, _t(cell), pbc, max_neighbors=320)
    num3.add_(0)
    assert E.full_list_counts(nm3, sh3) is None
    nm4, num4, sh4 = cell_list(_t(pos), 7.0, _t(cell), pbc, max_neighbors=320)
    E.invalidate(sh4)
    assert E.full_list_counts(nm4, sh4) is None


@pytest.mark.parametrize("how", ["data", "storage_copy"])
def test_bulk_edits_behind_torchs_back_are_caught_by_the_sampled_look_up(how, monkeypatch):
    """Rows of every second atom emptied without moving a version counter: the record still looks valid, the kernel's sampled rows look
    their entry up in the partner's row, miss it, and the call takes the scatter path -- the result of the arrays as they ARE."""
    from nvalchemiops.neighborlist import _engine as E
    from nvalchemiops.neighborlist import cell_list

    n, dtype = 1500, np.float64
    pos, cell, q = _system(n, dtype, seed=9, box=26.0)
    pbc = torch.tensor([True] * 3, device=DEV)
    nm, num, sh = cell_list(_t(pos), 7.0, _t(cell), pbc, max_neighbors=192)
    assert int(num.max()) <= 192
    edited = nm.clone()
    edited[::2] = n
    if how == "data":
        nm.data[::2] = n
    else:
        nm.untyped_storage().copy_(edited.untyped_storage())
    torch.cuda.synchronize()
    assert E.full_list_counts(nm, sh) is num  # torch saw nothing
    out = _run(pos, q, cell, nm, sh, True, monkeypatch, n)
    ref = _oracle(pos, q, cell, nm, sh, n)
    _same(out, ref, dtype, f"bulk edit ({how}) vs oracle", float(np.abs(ref[1]).max()))
    _same(out, _run(pos, q, cell, nm, sh, False, monkeypatch, n), dtype, f"bulk edit ({how}) vs checksum path", float(np.abs(ref[1]).max()))


def test_particle_mesh_ewald_on_a_search_output_is_unchanged(monkeypatch):
    """The public entry point the headline step calls: same energies and forces with and without the trusted real-space form."""
    from nvalchemiops.interactions.electrostatics import ewald as EW
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
    from nvalchemiops.neighborlist import cell_list

    n, dtype = 800, np.float64
    pos, cell, q = _system(n, dtype, seed=11, box=24.0)
    pbc = torch.tensor([True] * 3, device=DEV)
    nm = torch.empty((n, 256), dtype=torch.int32, device=DEV)
    sh = torch.empty((n, 256, 3), dtype=torch.int32, device=DEV)
    num = torch.empty(n, dtype=torch.int32, device=DEV)
    res = {}
    for trust in (True, False):
        monkeypatch.setattr(EW, "_TRUST_FULL_LISTS", trust)
        cell_list(_t(pos), 9.0, _t(cell), pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
        res[trust] = particle_mesh_ewald(_t(pos), _t(q), _t(cell), alpha=0.35, mesh_dimensions=(32, 32, 32), spline_order=5, neighbor_matrix=nm,
                                         neighbor_matrix_shifts=sh, compute_forces=True)
    for a, b in zip(res[True], res[False]):
        assert float((a - b).abs().max()) <= 1e-11 * max(1.0, float(b.abs().max()))


def test_a_mask_value_that_is_an_atom_index_keeps_the_checksum_path(monkeypatch):
    """mask_value = 5 makes every stored entry j == 5 padding: the list the sum sees is no longer the search's symmetric list, so the trusted
    form is not taken -- both settings run the checksum kernel and its scatter fix-up and agree to rounding.  (No oracle leg: with a mask_value other than the
    fill value the reference -- and its restatement -- index out of bounds on the real padding.)"""
    from nvalchemiops.interactions.electrostatics import ewald as EW
    from nvalchemiops.interactions.electrostatics import ewald_real_space
    from nvalchemiops.neighborlist import _engine as E
    from nvalchemiops.neighborlist import cell_list

    n, dtype = 250, np.float64
    pos, cell, q = _system(n, dtype, seed=13)
    pbc = torch.tensor([True] * 3, device=DEV)
    nm, num, sh = cell_list(_t(pos), 7.0, _t(cell), pbc, max_neighbors=320)
    assert E.full_list_counts(nm, sh) is num
    alpha = torch.tensor([0.4], dtype=torch.float64, device=DEV)
    out = {}
    for trust in (True, False):
        monkeypatch.setattr(EW, "_TRUST_FULL_LISTS", trust)
        out[trust] = ewald_real_space(_t(pos), _t(q), _t(cell)[None], alpha, neighbor_matrix=nm, neighbor_matrix_shifts=sh, mask_value=5,
                                      compute_forces=True, compute_charge_gradients=True)
    for a, b in zip(out[True], out[False]):  # (the scatter path adds with atomics: equal to rounding, not bit for bit)
        assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 1e-11 * max(1.0, float(b.abs().max()))
    # the entries to atom 5 really are dropped: other rows' energies differ from the ordinary call's
    monkeypatch.setattr(EW, "_TRUST_FULL_LISTS", True)
    plain = ewald_real_space(_t(pos), _t(q), _t(cell)[None], alpha, neighbor_matrix=nm, neighbor_matrix_shifts=sh, mask_value=n,
                             compute_forces=True, compute_charge_gradients=True)
    assert not torch.equal(plain[0], out[True][0])

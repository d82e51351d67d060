"""CPU-side tests (`-m "not gpu"`): the C ABI library loads and exports every symbol the header declares, host-side API
logic (validation, empty inputs, parameter estimation, dispatcher rules), the no-CPU-fallback guarantee, and the
world_size-2 gloo test of the batch-sharding path."""
import math
import os
import socket

import numpy as np
import pytest
import torch

from tests import systems as S


def test_library_exports_every_declared_symbol():
    from nvalchemiops import _capi
    from tools.abi_symbols import declared_symbols

    lib = _capi.lib()
    names = declared_symbols()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/nvalchemiops_hip.h but not exported"
    assert lib.mi_version() == 1
    assert lib.mi_nl_workspace_bytes(100000, 1, 0) > 0


def test_comm_entry_points_validate_before_touching_rccl():
    """`mi_comm_*` (csrc/comm.cpp): argument errors are MI_EINVAL with a message, a NULL communicator destroys to MI_OK -- none of this needs
    RCCL or a GPU (RCCL is bound on the first call that really needs it)."""
    import ctypes

    from nvalchemiops import _capi

    lib = _capi.lib()
    comm = ctypes.c_void_p()
    ident = ctypes.create_string_buffer(128)
    assert lib.mi_comm_unique_id(None, ctypes.c_size_t(128)) == -1
    assert lib.mi_comm_unique_id(ident, ctypes.c_size_t(64)) == -1 and b"128" in lib.mi_last_error()
    assert lib.mi_comm_init(None, ctypes.c_size_t(128), 2, 0, ctypes.byref(comm)) == -1
    assert lib.mi_comm_init(ident, ctypes.c_size_t(128), 2, 2, ctypes.byref(comm)) == -1 and b"rank 2 of 2" in lib.mi_last_error()
    assert lib.mi_comm_init(ident, ctypes.c_size_t(128), 0, 0, ctypes.byref(comm)) == -1 and comm.value is None
    assert lib.mi_comm_init(ident, ctypes.c_size_t(128), 1, 0, None) == -1
    assert lib.mi_comm_size(None, None, None) == -1
    assert lib.mi_comm_allgather_f32(None, None, None, ctypes.c_size_t(4), None) == -1
    assert lib.mi_comm_allgather_f64(None, None, None, ctypes.c_size_t(4), None) == -1
    assert lib.mi_comm_library_version(None) == -1
    assert lib.mi_comm_destroy(None) == 0


def test_no_cpu_fallback():
    from nvalchemiops._capi import NativeLibraryError
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
    from nvalchemiops.interactions.electrostatics import ewald_real_space, pme_reciprocal_space
    from nvalchemiops.neighborlist import cell_list, neighbor_list

    pos = torch.rand(10, 3)
    cell, pbc = torch.eye(3) * 5, torch.tensor([True] * 3)
    with pytest.raises(NativeLibraryError):
        cell_list(pos, 1.0, cell, pbc)
    with pytest.raises(NativeLibraryError):
        neighbor_list(pos, 1.0)
    p = D3Parameters(rcov=torch.rand(5), r4r2=torch.rand(5), c6ab=torch.rand(5, 5, 5, 5), cn_ref=torch.rand(5, 5, 5, 5))
    with pytest.raises(NativeLibraryError):
        dftd3(pos, torch.ones(10, dtype=torch.int32), 0.4, 4.0, 0.8, d3_params=p, neighbor_matrix=torch.zeros((10, 4), dtype=torch.int32))
    with pytest.raises(NativeLibraryError):
        pme_reciprocal_space(pos, torch.rand(10), cell, 0.3, mesh_dimensions=(8, 8, 8))
    with pytest.raises(NativeLibraryError):
        ewald_real_space(pos, torch.rand(10), cell, torch.tensor([0.3]), neighbor_matrix=torch.zeros((10, 4), dtype=torch.int32))


def test_empty_inputs_and_shapes():
    from nvalchemiops.neighborlist import batch_cell_list, batch_naive_neighbor_list, cell_list, naive_neighbor_list

    cell, pbc = torch.eye(3), torch.tensor([True] * 3)
    nm, num, sh = cell_list(torch.zeros(0, 3), 2.0, cell, pbc)
    assert nm.shape == (0, 0) and num.shape == (0,) and sh.shape == (0, 0, 3)
    lst, ptr, lsh = cell_list(torch.zeros(0, 3), 2.0, cell, pbc, return_neighbor_list=True)
    assert lst.shape == (2, 0) and ptr.shape == (1,) and lsh.shape == (0, 3)
    nm, num, sh = cell_list(torch.zeros(4, 3), 0.0, cell, pbc)  # cutoff <= 0
    assert nm.shape == (4, 0) and num.tolist() == [0] * 4 and (nm == 4).all()
    nm, num, sh = batch_cell_list(torch.zeros(0, 3), 2.0, cell[None], pbc[None], torch.zeros(0, dtype=torch.int32))
    assert nm.shape == (0, 0)
    nm, num = naive_neighbor_list(torch.zeros(3, 3), 0.0, max_neighbors=4)
    assert nm.shape == (3, 4) and (nm == 3).all() and num.tolist() == [0, 0, 0]
    with pytest.raises(ValueError):
        naive_neighbor_list(torch.zeros(3, 3), 1.0, cell=cell)
    from nvalchemiops.neighborlist import neighbor_list

    with pytest.raises(ValueError):
        neighbor_list(torch.zeros(3, 3), 1.0, method="nope")
    out = neighbor_list(torch.zeros(0, 3), 1.0, cutoff2=2.0)  # dual cutoff: interleaved (matrix1, num1, matrix2, num2)
    assert len(out) == 4 and out[0].shape[0] == 0 and out[2].shape[0] == 0
    # no atoms / one atom through the dispatcher, COO output (test_neighborlist.py:905-950): (list, ptr) with ptr = [0] / [0, 0]
    lst, ptr = neighbor_list(torch.zeros(0, 3), 2.0, method="naive", return_neighbor_list=True)
    assert lst.shape == (2, 0) and ptr.tolist() == [0]
    # cutoff <= 0 keeps the reference's own tuple for that case (naive.py:627-657): (list, zeros[n], zeros[n+1])
    out = neighbor_list(torch.zeros(1, 3), 0.0, method="naive", return_neighbor_list=True)
    assert len(out) == 3 and out[0].shape == (2, 0) and out[1].shape == (1,) and out[2].shape == (2,)
    lst, ptr = batch_naive_neighbor_list(torch.zeros(0, 3), 2.0, batch_idx=torch.zeros(0, dtype=torch.int32),
                                         batch_ptr=torch.zeros(1, dtype=torch.int32), return_neighbor_list=True)
    assert lst.shape == (2, 0) and ptr.tolist() == [0]
    lst, ptr, lsh = cell_list(torch.zeros(8, 3), 0.0, cell, pbc, return_neighbor_list=True)  # test_cell_list.py:292-313
    assert lst.shape == (2, 0) and ptr.shape == (9,) and lsh.shape == (0, 3)
    from nvalchemiops.neighborlist import estimate_cell_list_sizes

    for c_, p_, rc in ((torch.zeros((0, 3, 3)), torch.zeros((0, 3), dtype=torch.bool), 1.0), (torch.eye(3).reshape(1, 3, 3), pbc[None], -1.0)):
        ncell, radius = estimate_cell_list_sizes(c_, p_, rc)  # test_cell_list.py:427-446
        assert ncell == 1 and radius.shape == (3,) and radius.dtype == torch.int32


def test_estimate_max_neighbors_and_overflow_error():
    from nvalchemiops.neighborlist import NeighborOverflowError, estimate_max_neighbors

    assert estimate_max_neighbors(5.0) == 928 and estimate_max_neighbors(6.0) == 1584 and estimate_max_neighbors(0.0) == 0
    assert estimate_max_neighbors(21.2) == 69856
    assert "12 > 8" in str(NeighborOverflowError(8, 12))


def test_pme_parameter_estimation_matches_formulas():
    from nvalchemiops.interactions.electrostatics import (estimate_ewald_parameters, estimate_pme_mesh_dimensions, estimate_pme_parameters,
                                                          mesh_spacing_to_dimensions)

    pos = torch.randn(100, 3, dtype=torch.float64)
    cell = torch.eye(3, dtype=torch.float64) * 20.0
    p = estimate_ewald_parameters(pos, cell, accuracy=1e-6)
    eta = (8000.0**2 / 100) ** (1 / 6) / math.sqrt(2 * math.pi)
    assert abs(p.alpha.item() - 1 / (math.sqrt(2) * eta)) < 1e-12
    assert abs(p.real_space_cutoff.item() - math.sqrt(-2 * math.log(1e-6)) * eta) < 1e-12
    assert abs(p.reciprocal_space_cutoff.item() - math.sqrt(-2 * math.log(1e-6)) / eta) < 1e-12
    q = estimate_pme_parameters(pos, cell)
    want = 2 ** math.ceil(math.log2(2 * q.alpha.item() * 20.0 / (3 * 1e-6**0.2)))
    assert q.mesh_dimensions == (want,) * 3
    assert estimate_pme_mesh_dimensions(cell, torch.tensor([0.3], dtype=torch.float64)) == (64, 64, 64)  # docstring example of the reference
    assert mesh_spacing_to_dimensions(cell, 0.9) == (32, 32, 32)
    # batch: per-system atom counts through batch_idx
    cells = torch.stack([cell, cell * 2])
    bi = torch.tensor([0] * 40 + [1] * 60, dtype=torch.int32)
    pb = estimate_ewald_parameters(pos, cells, bi)
    eta1 = ((8000.0 * 8) ** 2 / 60) ** (1 / 6) / math.sqrt(2 * math.pi)
    assert abs(pb.alpha[1].item() - 1 / (math.sqrt(2) * eta1)) < 1e-12


def test_k_vectors_match_numpy():
    from nvalchemiops.interactions.electrostatics import generate_k_vectors_pme
    from oracle import oracle as O

    cell = np.array([[10.0, 0, 0], [2, 9, 0], [1, -1, 11]])
    kv, k2 = generate_k_vectors_pme(torch.as_tensor(cell), (8, 6, 10))
    okv, ok2 = O.generate_k_vectors_pme(cell, (8, 6, 10))
    assert kv.shape == (8, 6, 6, 3)
    np.testing.assert_allclose(kv.numpy(), okv, rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(k2.numpy(), ok2, rtol=1e-13, atol=1e-14)


def test_d3_parameter_validation_and_dispatch_rules():
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3

    with pytest.raises(ValueError):
        D3Parameters(rcov=torch.rand(1), r4r2=torch.rand(1), c6ab=torch.rand(1, 1, 5, 5), cn_ref=torch.rand(1, 1, 5, 5))
    with pytest.raises(TypeError):
        D3Parameters(rcov=torch.rand(5).int(), r4r2=torch.rand(5), c6ab=torch.rand(5, 5, 5, 5), cn_ref=torch.rand(5, 5, 5, 5))
    p = D3Parameters(rcov=torch.rand(5), r4r2=torch.rand(5), c6ab=torch.rand(5, 5, 5, 5), cn_ref=torch.rand(5, 5, 5, 5))
    assert p.max_z == 4 and p.to(dtype=torch.float64).c6ab.dtype == torch.float64
    pos, z = torch.zeros(2, 3), torch.ones(2, dtype=torch.int32)
    nm = torch.zeros((2, 2), dtype=torch.int32)
    with pytest.raises(ValueError):
        dftd3(pos, z, 0.4, 4.0, 0.8, d3_params=p, neighbor_matrix=nm, neighbor_list=nm)
    with pytest.raises(ValueError):
        dftd3(pos, z, 0.4, 4.0, 0.8, d3_params=p, neighbor_list=nm)  # neighbor_ptr missing
    with pytest.raises(RuntimeError):
        dftd3(pos, z, 0.4, 4.0, 0.8, neighbor_matrix=nm)
    e, f, cn, v = dftd3(torch.zeros(0, 3), torch.zeros(0, dtype=torch.int32), 0.4, 4.0, 0.8, d3_params=p, neighbor_matrix=torch.zeros((0, 2), dtype=torch.int32),
                        cell=torch.eye(3)[None], neighbor_matrix_shifts=torch.zeros((0, 2, 3), dtype=torch.int32), compute_virial=True)
    assert e.shape == (1,) and f.shape == (0, 3) and v.shape == (0, 3, 3)


def test_partition_systems_balances_atoms():
    from nvalchemiops.distributed import partition_systems

    parts = partition_systems([2000] * 1024, 8)
    assert parts == [(128 * r, 128 * (r + 1)) for r in range(8)]
    ragged = [10, 500, 20, 30, 400, 40, 5, 5]
    p2 = partition_systems(ragged, 2)
    assert p2[0][0] == 0 and p2[-1][1] == len(ragged) and p2[0][1] == p2[1][0]
    assert abs(sum(ragged[: p2[0][1]]) - sum(ragged) / 2) <= 250


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_worker(rank, world, port, out):
    import torch.distributed as dist

    from nvalchemiops.distributed import all_gather_system_values, segment_energy, shard_batch

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counts = [3, 5, 2, 4, 6]
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32)
    n = int(ptr[-1])
    per_atom_e = torch.arange(n, dtype=torch.float64)
    s0, s1, a0, a1, local_idx, (e_loc,), (cells,) = shard_batch(ptr, rank, world, per_atom_e, per_system=(torch.arange(5.0)[:, None],))
    assert cells.shape[0] == s1 - s0 and local_idx.shape[0] == a1 - a0 and (local_idx.max().item() == s1 - s0 - 1)
    local_sys = segment_energy(e_loc, local_idx, s1 - s0)
    from nvalchemiops.distributed import partition_systems

    sizes = [b - a for a, b in partition_systems(counts, world)]
    full = all_gather_system_values(local_sys, sizes)
    expect = torch.tensor([per_atom_e[ptr[i]:ptr[i + 1]].sum() for i in range(5)], dtype=torch.float64)
    ok = torch.allclose(full, expect)
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_energy_gather():
    """N > 1 path on CPU: shard a ragged batch over 2 ranks, reduce per-system energies locally, one all_gather."""
    import torch.multiprocessing as mp

    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gloo_worker, args=(2, port, out), nprocs=2, join=True)
    assert out.get(0) is True and out.get(1) is True


def test_custom_ops_registered_under_reference_names():
    import nvalchemiops  # noqa: F401

    for name in ("build_cell_list", "query_cell_list", "batch_build_cell_list", "batch_query_cell_list", "dftd3_nm", "dftd3_nl"):
        op = getattr(torch.ops.nvalchemiops, name)
        schema = str(op.default._schema)
        assert "-> ()" in schema and "Tensor(a" in schema, schema  # mutating ops returning None, as in the reference


def test_public_signatures_match_reference():
    """Drop-in boundary: every public function on the path has the reference's argument names, order and defaults
    (tests/golden/reference_signatures.json, written by tests/golden/make_signatures.py from the reference sources)."""
    import importlib.util
    import json

    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_signatures", os.path.join(here, "golden", "make_signatures.py"))
    ms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ms)
    want = json.load(open(os.path.join(here, "golden", "reference_signatures.json")))
    mine = os.path.join(os.path.dirname(here), "nvalchemi-toolkit-ops_amd", "nvalchemiops") + os.sep
    checked = 0
    for rel, funcs in want.items():
        for name, sig in funcs.items():
            assert ms.signature(mine + rel, name) == sig, f"{rel}:{name} differs from the reference signature"
            checked += 1
    assert checked == 47


def test_bspline_deconvolution_factors():
    """Properties the reference's tests check (test_spline.py:1537-1600) plus the closed forms of its coefficient table
    (spline.py:2999-3022: the "order n" table is the order-(n+1) cardinal B-spline at the integers 1..n)."""
    import math

    from nvalchemiops.spline import compute_bspline_deconvolution, compute_bspline_deconvolution_1d

    for order in (1, 2, 3, 4, 5, 6):
        d = compute_bspline_deconvolution((8, 12, 16), order)
        assert d.shape == (8, 12, 16) and d.dtype == torch.float64
        assert abs(float(d[0, 0, 0]) - 1.0) < 1e-12 and bool((d > 0).all())
        assert torch.allclose(d[1:], d[1:].flip(0)) and torch.allclose(d[:, 1:], d[:, 1:].flip(1)) and torch.allclose(d[..., 1:], d[..., 1:].flip(2))
        d1 = [compute_bspline_deconvolution_1d(n, order) for n in (8, 12, 16)]
        sep = d1[0].view(8, 1, 1) * d1[1].view(1, 12, 1) * d1[2].view(1, 1, 16)
        ok = sep < 1e14  # even orders vanish at the Nyquist frequency: |b|^2 is clamped to 1e-15 there (spline.py:3108)
        assert torch.allclose(d[ok], sep[ok], rtol=1e-12)
    table = {2: [0.5, 0.5], 3: [1 / 6, 4 / 6, 1 / 6], 4: [1 / 24, 11 / 24, 11 / 24, 1 / 24]}
    for order, c in table.items():
        n = 10
        for k in (1, 3, 4):  # not the Nyquist index (clamped for even orders)
            b = sum(cj * complex(math.cos(2 * math.pi * k * j / n), math.sin(2 * math.pi * k * j / n)) for j, cj in enumerate(c))
            assert abs(float(compute_bspline_deconvolution_1d(n, order)[k]) - 1.0 / abs(b) ** 2) < 1e-9 / abs(b) ** 2


def test_rebuild_detection_and_batch_size_edge_cases():
    """Edge cases the reference tests on every device (test_rebuild_detection.py:188-310, test_batch_cell_list.py:550-640): empty
    systems never need a rebuild, mismatched atom counts always do, empty / negative-cutoff size estimates, empty batch build."""
    from nvalchemiops.neighborlist import (batch_build_cell_list, cell_list_needs_rebuild, estimate_batch_cell_list_sizes,
                                           neighbor_list_needs_rebuild)

    i32 = lambda x: torch.tensor(x, dtype=torch.int32)  # noqa: E731
    r = cell_list_needs_rebuild(current_positions=torch.empty((0, 3)), atom_to_cell_mapping=torch.empty((0, 3), dtype=torch.int32),
                                cells_per_dimension=i32([1, 1, 1]), cell=torch.eye(3).unsqueeze(0), pbc=torch.tensor([True, True, True]))
    assert r.shape == (1,) and r.dtype == torch.bool and not r.item()
    r = neighbor_list_needs_rebuild(reference_positions=torch.randn(5, 3), current_positions=torch.randn(7, 3), skin_distance_threshold=0.5)
    assert r.shape == (1,) and r.dtype == torch.bool and r.item()
    r = neighbor_list_needs_rebuild(reference_positions=torch.empty((0, 3)), current_positions=torch.empty((0, 3)), skin_distance_threshold=0.5)
    assert r.shape == (1,) and not r.item()
    ncell, radius = estimate_batch_cell_list_sizes(torch.zeros((0, 3, 3)), torch.zeros((0, 3), dtype=torch.bool), 1.0)
    assert ncell == 1 and radius.shape == (0, 3) and radius.dtype == torch.int32
    ncell, radius = estimate_batch_cell_list_sizes(torch.eye(3).reshape(1, 3, 3), torch.tensor([[True, True, True]]), -1.0)
    assert ncell == 1 and radius.shape == (1, 3) and radius.dtype == torch.int32
    batch_build_cell_list(torch.empty(0, 3), 1.0, torch.eye(3).reshape(1, 3, 3), torch.tensor([[True, True, True]]), torch.empty(0, dtype=torch.int32),
                          i32([1, 1, 1]), i32([1, 1, 1]), i32([0, 0, 0]), i32([0, 0, 0]), i32([0]), i32([0]), i32([]))  # returns without touching a device


def test_openmp_oracle_equals_serial_oracle():
    """The -fopenmp build of the oracle (bench.py's all-core cpu_baseline leg) computes what the serial parity oracle computes:
    identical neighbour sets / counts, D3 and PME equal up to the order of the cross-row additions."""
    from oracle import oracle as O
    from tests import systems as S

    pos, cell, q, numbers = S.fcc_box(500, dtype=np.float64)
    nm, num, sh = O.cell_list(pos, 7.0, cell, [True] * 3, max_neighbors=160)
    p32, c32 = (pos * 1.8897261246).astype(np.float32), (cell * 1.8897261246).astype(np.float32)
    dm, dnum, dsh = O.cell_list(p32, 20.0, c32, [True] * 3, max_neighbors=512)
    tab = O.d3_test_tables(17)
    kw = dict(neighbor_matrix=dm, neighbor_matrix_shifts=dsh, cell=c32, compute_virial=True)
    d3 = O.dftd3(p32, numbers, tab, 0.4289, 4.4407, 0.7875, **kw)
    with O.extended_splines():
        pme = O.particle_mesh_ewald(pos, q, cell, 0.35, (24, 24, 24), 5, neighbor_matrix=nm, neighbor_matrix_shifts=sh, compute_forces=True)
        with O.openmp(2) as om:
            assert om.threads == 2
            nm2, num2, sh2 = O.cell_list(pos, 7.0, cell, [True] * 3, max_neighbors=160)
            d3b = O.dftd3(p32, numbers, tab, 0.4289, 4.4407, 0.7875, **kw)
            pmeb = O.particle_mesh_ewald(pos, q, cell, 0.35, (24, 24, 24), 5, neighbor_matrix=nm, neighbor_matrix_shifts=sh, compute_forces=True)
    assert np.array_equal(num, num2) and np.array_equal(O.canonical_pairs(nm, num, sh), O.canonical_pairs(nm2, num2, sh2))
    for a, b in zip(d3, d3b):
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(pme[0], pmeb[0], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(pme[1], pmeb[1], rtol=1e-11, atol=1e-13)
    # the serial library is untouched by the switch
    assert np.array_equal(O.dftd3(p32, numbers, tab, 0.4289, 4.4407, 0.7875, **kw)[1], d3[1])


def test_every_reference_custom_op_name_is_registered():
    """The reference registers 36 `alchemiops::*` ops (spline.py, interactions/electrostatics/pme.py, ewald.py) and 8 `nvalchemiops::*`
    Coulomb ops (coulomb.py:716-1330) with torch.library; code that calls `torch.ops.<ns>.<name>` directly must find them here.  The four
    multi-channel spline ops (multipole path, out of scope: SURVEY section 8) are served by their Python composition only."""
    import torch

    import nvalchemiops._eops  # noqa: F401

    have = {n.split(".")[0] for n in torch._C._dispatch_get_all_op_names()}
    variants = {"spline": ["_spline_spread", "_spline_gather", "_spline_gather_vec3", "_spline_gather_gradient"],
                "pme": ["_pme_green_structure_factor", "_pme_energy_corrections", "_pme_energy_corrections_with_charge_grad"],
                "recip": ["_ewald_reciprocal_space_" + s for s in ("energy", "energy_forces", "energy_forces_charge_grad")],
                "real": ["_ewald_real_space_" + s + m for s in ("energy", "energy_forces", "energy_forces_charge_grad") for m in ("", "_matrix")]}
    names = [f"alchemiops::{b}{n}" for group in variants.values() for n in group for b in ("", "_batch")]
    names += [f"nvalchemiops::{b}_coulomb_energy{f}_{fmt}" for b in ("", "_batch") for f in ("", "_forces") for fmt in ("list", "matrix")]
    assert len(names) == 32 + 8
    missing = [n for n in names if n not in have]
    assert not missing, missing
    # argument names and order are the reference's (a keyword call written against the reference must bind)
    s = torch.ops.alchemiops._batch_ewald_reciprocal_space_energy_forces.default._schema
    assert [a.name for a in s.arguments] == ["positions", "charges", "cell", "k_vectors", "alpha", "batch_idx"]
    s = torch.ops.nvalchemiops._batch_coulomb_energy_matrix.default._schema
    assert [a.name for a in s.arguments] == ["positions", "charges", "cell", "neighbor_matrix", "neighbor_matrix_shifts", "batch_idx", "cutoff",
                                             "alpha", "fill_value"]


def test_spread_path_policy():
    """Host policy of the PME mesh path (no compute): the tile pipeline only where it is possible (`mi_spline_spread_is_tiled`) AND measured faster
    (>= 12 000 atoms, >= 128 mesh tiles over all systems: profiles/r04_ab_spread_path.log); reference-mode orders 5 / 6 never build a tile list."""
    from nvalchemiops import _capi as C

    L = C.lib()
    prefers, possible = L.mi_spline_spread_prefers_tiles, L.mi_spline_spread_is_tiled
    assert possible(1, 128, 128, 128, 5) == 1 and prefers(100_000, 1, 128, 128, 128, 5) == 1
    assert prefers(11_999, 1, 128, 128, 128, 5) == 0 and prefers(12_000, 1, 128, 128, 128, 5) == 1
    assert possible(1, 32, 32, 32, 4) == 1 and prefers(100_000, 1, 32, 32, 32, 4) == 0      # 64 tiles: too few blocks for 256 CUs
    assert prefers(256_000, 128, 32, 32, 32, 4) == 1                                         # ... but 128 systems of them are 8192 tiles
    assert possible(1, 31, 31, 31, 4) == 0 and prefers(100_000, 1, 31, 31, 31, 4) == 0       # prime mesh: atomic kernel
    assert prefers(100_000, 1, 128, 128, 128, 5 | C.SPLINE_REFERENCE_ORDERS) == 0


def test_mesh_solve_policy():
    """Host logic of the fused PME mesh solve (no compute): which meshes it supports (powers of two whose (ny, nz/2+1) complex plane and tables
    fit 160 KB of LDS), where it is preferred (since round 5: wherever it is supported, profiles/r05_ab_solve_size.log), and that its scratch covers
    the half spectrum, the channel spectra and the per-call tables."""
    import ctypes

    from nvalchemiops import _capi as C

    L = C.lib()
    ok, pref = L.mi_pme_solve_supported, L.mi_pme_solve_preferred
    f32, f64 = 0, 1
    assert ok(1, 128, 128, 128, f64) == 1 and ok(1, 256, 64, 256, f64) == 1 and ok(1, 128, 256, 128, f32) == 1
    assert ok(1, 128, 256, 128, f64) == 0 and ok(1, 256, 256, 256, f32) == 0        # plane larger than the LDS
    # round 6: mixed radix -- every axis length a product of 2, 3 and 5 (nz even): what `mesh_spacing=` callers get
    assert ok(1, 48, 48, 48, f64) == 1 and ok(1, 96, 100, 120, f64) == 1 and ok(3, 120, 96, 100, f32) == 1
    assert ok(1, 42, 48, 48, f64) == 0 and ok(1, 48, 48, 45, f64) == 0 and ok(1, 48, 48, 14, f64) == 0  # a factor 7; odd nz; nz / 2 = 7
    assert ok(1, 31, 9, 6, f64) == 0 and ok(1, 4, 8, 8, f64) == 0 and ok(0, 32, 32, 32, f64) == 0
    assert pref(1, 32, 32, 32, f64) == 1 and pref(128, 32, 32, 32, f64) == 1 and pref(8, 64, 64, 64, f64) == 1 and pref(2, 128, 128, 128, f64) == 1
    assert pref(1, 96, 100, 120, f64) == 1
    assert pref(1, 42, 48, 48, f64) == 0 and pref(4, 128, 256, 128, f64) == 0  # unsupported meshes keep the (self-tested) plans
    # the transforms on their own (`mi_fft_lds`): one half spectrum of scratch + the tables; 0 for a mesh they do not take
    assert 128 * 128 * 65 * 16 <= int(L.mi_fft_lds_scratch_bytes(1, 128, 128, 128, f64)) <= 128 * 128 * 65 * 16 + 65536
    assert int(L.mi_fft_lds_scratch_bytes(1, 42, 48, 48, f64)) == 0
    L.mi_pme_solve_scratch_bytes.restype = ctypes.c_size_t
    for nch in (1, 4):
        half = 128 * 128 * 65 * 16
        need = int(L.mi_pme_solve_scratch_bytes(1, 128, 128, 128, nch, f64))
        assert half * (1 + nch) <= need <= half * (1 + nch) + 3 * 256 + 3 * 128 * 16 + 6 * 128 * 8 + 1024


def test_fft_plan_self_test_logic():
    """`_FftPlan.self_test` (the known-answer check a new hipFFT plan gets at creation) on CPU tensors with stand-in plans that
    run torch.fft: a correct transform passes, a transform that is off fails, a check that cannot run reports None -- both directions, both
    precisions, a batch of three."""
    from nvalchemiops import _capi as C
    from nvalchemiops.interactions.electrostatics import pme as P

    class StandIn(P._FftPlan):
        def __init__(self, inverse, dims, scale):
            self.inverse, self.dims, self.scale = inverse, dims, scale

        def __call__(self, src, dst):
            if self.scale is None:
                raise RuntimeError("exec failed")
            r = torch.fft.irfftn(src, s=self.dims, dim=(1, 2, 3), norm="forward") if self.inverse else torch.fft.rfftn(src, dim=(1, 2, 3))
            dst.copy_(r * self.scale)

    dims = (6, 5, 8)
    for dt in (torch.float32, torch.float64):
        for inverse in (False, True):
            for scale, expect in ((1.0, True), (1.6, False), (float("nan"), False), (None, None)):
                ok, detail = StandIn(inverse, dims, scale).self_test(torch.device("cpu"), dims, 3, C.dtype_code(dt), inverse)
                assert ok is expect, (dt, inverse, scale, ok, detail)


def test_fft_plan_cache_is_fail_safe_and_bounded(monkeypatch):
    """`_fft_plan` (pme.py) with stand-in plans on CPU tensors: a plan that fails -- or cannot run -- its self-test is DESTROYED and the key is
    served by the dense-DFT stand-in (exact), with one warning; the cache is an LRU that destroys what it evicts and re-tests what comes back."""
    import warnings

    import numpy as np

    from nvalchemiops import _capi as C
    from nvalchemiops.interactions.electrostatics import pme as P

    log = []

    class Fake(P._FftPlan):
        bad_shapes = {(6, 5, 8): 1.6, (4, 4, 6): None}

        def __init__(self, dims, batch, code, inverse):
            self.dims, self.batch, self.inverse = tuple(dims), batch, inverse
            self.scale = self.bad_shapes.get(self.dims, 1.0)
            log.append(("create", self.dims, inverse))

        def __call__(self, src, dst):
            if self.scale is None:
                raise RuntimeError("exec failed")
            r = torch.fft.irfftn(src, s=self.dims, dim=(1, 2, 3), norm="forward") if self.inverse else torch.fft.rfftn(src, dim=(1, 2, 3))
            dst.copy_(r * self.scale)

        def destroy(self):
            log.append(("destroy", self.dims, self.inverse))

    class CpuDft:  # stand-in for `_DenseDft` (whose kernels need the GPU): the same contract on CPU tensors
        def __init__(self, dims, batch, code, inverse):
            self.dims, self.batch, self.inverse = tuple(dims), batch, inverse

        def __call__(self, src, dst):
            nx, ny, nz = self.dims
            if self.inverse:
                out = torch.fft.irfftn(src.reshape(self.batch, nx, ny, nz // 2 + 1), s=self.dims, dim=(1, 2, 3), norm="forward")
            else:
                out = torch.fft.rfftn(src.reshape(self.batch, nx, ny, nz), dim=(1, 2, 3), norm="backward")
            dst.view(out.shape).copy_(out)

        def destroy(self):
            pass

    monkeypatch.setattr(P, "_FftPlan", Fake)
    monkeypatch.setattr(P, "_DenseDft", CpuDft)
    monkeypatch.setattr(P, "_FFT_PLANS", type(P._FFT_PLANS)())
    monkeypatch.setattr(P, "_FFT_FALLBACKS", [])
    monkeypatch.setattr(P, "_FFT_PLAN_CAP", 3)
    dev, code = torch.device("cpu"), C.dtype_code(torch.float64)
    g = torch.Generator().manual_seed(0)
    for dims in ((6, 5, 8), (4, 4, 6)):  # wrong transform; exec raises
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            plan = P._fft_plan(dev, dims, 2, code, False)
            again = P._fft_plan(dev, dims, 2, code, False)
        assert isinstance(plan, CpuDft) and again is plan and len(caught) == 1, [str(w.message) for w in caught]
        assert ("destroy", dims, False) in log
        mesh = torch.randn((2,) + dims, generator=g, dtype=torch.float64)
        out = torch.empty((2, dims[0], dims[1], dims[2] // 2 + 1), dtype=torch.complex128)
        plan(mesh, out)
        assert np.abs(out.numpy() - np.fft.rfftn(mesh.numpy(), axes=(1, 2, 3))).max() < 1e-12
        back = torch.empty_like(mesh)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            P._fft_plan(dev, dims, 2, code, True)(out.clone(), back)
        assert (back / float(np.prod(dims)) - mesh).abs().max() < 1e-12
    assert len(P._FFT_FALLBACKS) == 4
    # LRU: capacity 3 -> creating more good plans evicts (and destroys) the oldest; a shape that returns is created and tested again
    P._FFT_PLANS.clear()
    log.clear()
    shapes = [(8, 8, 8), (8, 8, 10), (8, 10, 8), (10, 8, 8)]
    for d in shapes:
        assert isinstance(P._fft_plan(dev, d, 1, code, False), Fake)
    assert len(P._FFT_PLANS) == 3 and ("destroy", (8, 8, 8), False) in log
    P._fft_plan(dev, (8, 8, 10), 1, code, False)  # touch: now the most recent
    P._fft_plan(dev, (8, 8, 8), 1, code, False)   # back again: re-created, (8, 10, 8) is the one that goes
    assert log.count(("create", (8, 8, 8), False)) == 2 and ("destroy", (8, 10, 8), False) in log and ("destroy", (8, 8, 10), False) not in log
    # a plan that a captured HIP graph replays into (`pinned`: executed during stream capture) is never evicted: the graph holds its work area
    # by address; the eviction takes the oldest plan that is not pinned instead
    oldest = next(iter(P._FFT_PLANS))
    P._FFT_PLANS[oldest].pinned = True
    log.clear()
    for d in ((12, 8, 8), (8, 12, 8)):
        P._fft_plan(dev, d, 1, code, False)
    assert oldest in P._FFT_PLANS and ("destroy", oldest[1], False) not in log and len([e for e in log if e[0] == "destroy"]) == 2
    assert len(P._FFT_PLANS) == 3


def test_packed_companion_validity_rules():
    """Host logic of the packed companion (neighborlist/_engine.py), no GPU: it is served only for the very tensors it was built with, at
    the versions it was built at; every write -- by torch or by this package (`_written`) -- retires it; the "auto" policy learns a shape
    only from a matrix this package built and a consumer asked about."""
    import torch

    from nvalchemiops.neighborlist import _engine as E

    nm = torch.zeros((6, 4), dtype=torch.int32)
    sh = torch.zeros((6, 4, 3), dtype=torch.int32)
    E._written(nm, sh)
    words = torch.zeros(8, dtype=torch.uint8)
    setattr(nm, E._PACKED_ATTR, E.PackedCompanion(words, nm, sh, 6))
    assert E.packed_companion(nm, sh, 6).words is words
    assert E.packed_companion(nm, sh.clone(), 6) is None      # another tensor, equal contents
    assert E.packed_companion(nm, sh, 5) is None              # another index limit
    assert E.packed_companion(nm, None, 6) is None
    assert E.packed_companion(nm.clone(), sh, 6) is None      # attributes do not travel with clones
    sh.add_(0)                                                # any in-place op moves the version counter
    assert E.packed_companion(nm, sh, 6) is None
    setattr(nm, E._PACKED_ATTR, E.PackedCompanion(words, nm, sh, 6))
    assert E.packed_companion(nm, sh, 6).words is words
    nm[0, 0] = 3
    assert E.packed_companion(nm, sh, 6) is None
    setattr(nm, E._PACKED_ATTR, E.PackedCompanion(words, nm, sh, 6))
    E._written(nm)                                            # a raw-pointer write of this package: attribute gone, version bumped
    assert not hasattr(nm, E._PACKED_ATTR)
    v = nm._version
    E._written(nm)
    assert nm._version == v + 1
    # a shifts tensor that died: the weak reference is dead, never a match (even if a new tensor reuses the address)
    setattr(nm, E._PACKED_ATTR, E.PackedCompanion(words, nm, sh, 6))
    del sh
    assert E.packed_companion(nm, torch.zeros((6, 4, 3), dtype=torch.int32), 6) is None
    # the public way out for writers torch's version counters do not see (`tensor.data[...] = ...`, raw-pointer kernels, DLPack):
    # `invalidate` on either tensor drops the companion
    sh2 = torch.zeros((6, 4, 3), dtype=torch.int32)
    import weakref
    setattr(nm, E._PACKED_ATTR, E.PackedCompanion(words, nm, sh2, 6))
    setattr(sh2, "_nvalchemiops_owner", weakref.ref(nm))
    nm.data[0, 0] = 5                                         # does NOT move nm._version: the host-side record cannot see it
    assert E.packed_companion(nm, sh2, 6) is not None
    E.invalidate(sh2)
    assert E.packed_companion(nm, sh2, 6) is None and not hasattr(nm, E._PACKED_ATTR)
    setattr(nm, E._PACKED_ATTR, E.PackedCompanion(words, nm, sh2, 6))
    E.invalidate(nm)
    assert E.packed_companion(nm, sh2, 6) is None
    # sampled device-side check: (stride, phase) rotate from call to call
    a, b = E.verify_args(), E.verify_args()
    assert a[0] == b[0] and b[1] == a[1] + 1


def test_spread_workspace_is_sized_for_the_order_and_dtype_it_serves():
    """`mi_spline_spread_workspace_bytes_for`: the tile-box scratch scales with (e + order - 1)^3 and the mesh dtype; the
    any-order fp64 bound of `mi_spline_spread_workspace_bytes` is its maximum (host arithmetic, no GPU)."""
    from nvalchemiops import _capi as C

    L = C.lib()
    n, b, dims = 100000, 1, (256, 256, 256)
    full = int(L.mi_spline_spread_workspace_bytes(n, b, *dims))
    sizes = {(o, dt): int(L.mi_spline_spread_workspace_bytes_for(n, b, *dims, o, dt)) for o in (3, 4, 5, 6) for dt in (C.MI_F32, C.MI_F64)}
    assert sizes[(6, C.MI_F64)] == full
    assert all(v <= full for v in sizes.values())
    assert sizes[(4, C.MI_F32)] < 0.4 * full and sizes[(5, C.MI_F64)] < 0.85 * full
    assert sizes[(4, C.MI_F32)] < sizes[(4, C.MI_F64)] < sizes[(5, C.MI_F64)] < sizes[(6, C.MI_F64)]
    assert int(L.mi_spline_spread_workspace_bytes_for(n, b, *dims, 4 | C.SPLINE_REFERENCE_ORDERS, C.MI_F32)) == sizes[(4, C.MI_F32)]
    assert int(L.mi_spline_spread_workspace_bytes_for(n, b, 31, 29, 37, 4, C.MI_F64)) == 256  # prime dimensions: the atomic kernel, no scratch


def test_full_list_record_validity_rules():
    """`FullListRecord` (what lets `ewald_real_space` skip its symmetry checksums, round 6): alive only while matrix, shifts and counts are
    the tensors the search wrote, at the versions it left; `invalidate` (on either tensor) and the next search's `_written` drop it."""
    from nvalchemiops.neighborlist import _engine as E

    nm = torch.zeros((6, 4), dtype=torch.int32)
    sh = torch.zeros((6, 4, 3), dtype=torch.int32)
    num = torch.zeros(6, dtype=torch.int32)
    assert E.full_list_counts(nm, sh) is None and E.full_list_counts(None, None) is None
    E._record_full_list(nm, sh, num, 6, qualifies=False)
    assert E.full_list_counts(nm, sh) is None
    E._record_full_list(nm, sh, num, 6, qualifies=True)
    assert E.full_list_counts(nm, sh) is num
    assert E.full_list_counts(nm, sh.clone()) is None and E.full_list_counts(nm, None) is None   # other shifts tensor
    nm[0, 0] = 1
    assert E.full_list_counts(nm, sh) is None                                                      # version moved
    E._record_full_list(nm, sh, num, 6, qualifies=True)
    num += 0
    assert E.full_list_counts(nm, sh) is None
    E._record_full_list(nm, sh, num, 6, qualifies=True)
    E.invalidate(sh)                                                                                # through the shifts' owner link
    assert E.full_list_counts(nm, sh) is None
    E._record_full_list(nm, sh, num, 6, qualifies=True)
    E._written(nm, sh, num)                                                                         # what the next search does first
    assert E.full_list_counts(nm, sh) is None
    E._record_full_list(nm, sh, num, 6, qualifies=True)
    del num
    import gc
    gc.collect()
    assert E.full_list_counts(nm, sh) is None                                                      # the counts tensor is gone

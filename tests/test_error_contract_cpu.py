"""The Python-level error contract of the reference (SURVEY.md section 8b "Errors"): the exceptions its own tests expect with
`pytest.raises` -- type and message fragment -- are raised by this package before any device work, so they are checked on CPU tensors.
Sources: test/neighborlist/test_naive.py:922-940, test_batch_naive.py:715-735, test_naive_dual.py:753-765,
test_batch_naive_dual_cutoff.py:826-850, test_neighborlist.py:727/:867/:968, test/interactions/dispersion/test_dftd3.py:260-350,
:836-875, :1970, :2885-2960, test/interactions/electrostatics/test_pme.py:2018-2070, test_ewald.py:3596/:3729/:3760,
test_parameters.py:599, test_coulomb.py:1001-1057 (tests/test_oracle_golden.py covers the Coulomb messages)."""
import pytest
import torch


def test_neighbor_list_errors():
    from nvalchemiops.neighborlist import (batch_naive_neighbor_list, batch_naive_neighbor_list_dual_cutoff, naive_neighbor_list,
                                           naive_neighbor_list_dual_cutoff, neighbor_list)
    from nvalchemiops.neighborlist.neighbor_utils import _prepare_batch_idx_ptr

    pos, cell, pbc = torch.randn(50, 3), torch.eye(3).reshape(1, 3, 3) * 10, torch.ones(1, 3, dtype=torch.bool)
    bi = torch.zeros(50, dtype=torch.int32)
    for fn, args, kw in ((naive_neighbor_list, (pos, 2.0), {}), (batch_naive_neighbor_list, (pos, 2.0), dict(batch_idx=bi)),
                         (naive_neighbor_list_dual_cutoff, (pos, 2.0, 3.0), {}),
                         (batch_naive_neighbor_list_dual_cutoff, (pos, 2.0, 3.0), dict(batch_idx=bi))):
        with pytest.raises(ValueError, match="If cell is provided, pbc must also be provided"):
            fn(*args, cell=cell, **kw)
        with pytest.raises(ValueError, match="If pbc is provided, cell must also be provided"):
            fn(*args, pbc=pbc, **kw)
    # caller-supplied dual-cutoff outputs reach the kernel as raw pointers: wrong dtype / row count / width / layout is an error, not an
    # out-of-bounds device write
    i32 = dict(dtype=torch.int32)
    good = dict(neighbor_matrix1=torch.zeros((50, 8), **i32), neighbor_matrix2=torch.zeros((50, 16), **i32), num_neighbors1=torch.zeros(50, **i32),
                num_neighbors2=torch.zeros(50, **i32), neighbor_matrix_shifts1=torch.zeros((50, 8, 3), **i32),
                neighbor_matrix_shifts2=torch.zeros((50, 16, 3), **i32))
    for name, bad in (("neighbor_matrix1", torch.zeros((50, 8), dtype=torch.int64)), ("neighbor_matrix2", torch.zeros((49, 16), **i32)),
                      ("num_neighbors1", torch.zeros(51, **i32)), ("neighbor_matrix_shifts2", torch.zeros((50, 8, 3), **i32)),
                      ("neighbor_matrix_shifts1", torch.zeros((50, 3, 8), **i32).transpose(1, 2))):
        with pytest.raises(ValueError, match=name + " must be a contiguous int32 tensor"):
            naive_neighbor_list_dual_cutoff(pos, 2.0, 3.0, pbc=pbc, cell=cell, **{**good, name: bad})
    with pytest.raises(ValueError, match="Invalid method"):
        neighbor_list(pos, 2.0, method="invalid_method")
    with pytest.raises(TypeError):
        neighbor_list(pos, 2.0, method="naive", invalid_parameter_name=123)
    with pytest.raises(ValueError, match="Either batch_idx or batch_ptr must be provided."):
        _prepare_batch_idx_ptr(None, None, 50, torch.device("cpu"))


def test_d3_errors():
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3

    r = torch.rand
    ok = dict(rcov=r(10), r4r2=r(10), c6ab=r(10, 10, 5, 5), cn_ref=r(10, 10, 5, 5))
    for exc, match, bad in ((TypeError, "must be a torch.Tensor", dict(rcov=[1.0, 2.0])),
                            (TypeError, "must be float32 or float64", dict(rcov=torch.tensor([1, 2, 3], dtype=torch.int32))),
                            (ValueError, "rcov must be 1D tensor", dict(rcov=r(3, 3))),
                            (ValueError, "must have at least 2 elements", dict(rcov=r(1), r4r2=r(1), c6ab=r(1, 1, 5, 5), cn_ref=r(1, 1, 5, 5))),
                            (ValueError, "r4r2 must have shape", dict(r4r2=r(5))),
                            (ValueError, "c6ab must have shape", dict(c6ab=r(5, 5, 5, 5))),
                            (ValueError, "cn_ref must have shape", dict(cn_ref=r(5, 5, 5, 5)))):
        with pytest.raises(exc, match=match):
            D3Parameters(**{**ok, **bad})
    p = D3Parameters(**ok)
    pos, z = torch.rand(4, 3), torch.tensor([1, 2, 3, 4], dtype=torch.int32)
    nm = torch.zeros((4, 2), dtype=torch.int32)
    nl, ptr = torch.zeros((2, 3), dtype=torch.int32), torch.tensor([0, 1, 2, 3, 3], dtype=torch.int32)
    kw = dict(a1=0.4, a2=4.0, s8=0.8, d3_params=p)
    cell = torch.eye(3).reshape(1, 3, 3) * 10
    with pytest.raises(ValueError, match="Virial computation requires periodic boundary conditions"):
        dftd3(pos, z, neighbor_matrix=nm, compute_virial=True, **kw)
    with pytest.raises(ValueError, match="neighbor_matrix_shifts"):
        dftd3(pos, z, neighbor_matrix=nm, cell=cell, compute_virial=True, **kw)
    with pytest.raises(RuntimeError, match="DFT-D3 parameters must be explicitly provided"):
        dftd3(pos, z, a1=0.4, a2=4.0, s8=0.8, neighbor_matrix=nm)
    with pytest.raises(ValueError, match="Cannot provide both neighbor_matrix and neighbor_list"):
        dftd3(pos, z, neighbor_matrix=nm, neighbor_list=nl, neighbor_ptr=ptr, **kw)
    with pytest.raises(ValueError, match="Must provide either neighbor_matrix or neighbor_list"):
        dftd3(pos, z, **kw)
    with pytest.raises(ValueError, match="unit_shifts is for neighbor_list format"):
        dftd3(pos, z, neighbor_matrix=nm, unit_shifts=torch.zeros((3, 3), dtype=torch.int32), **kw)
    with pytest.raises(ValueError, match="neighbor_matrix_shifts is for neighbor_matrix format"):
        dftd3(pos, z, neighbor_list=nl, neighbor_ptr=ptr, neighbor_matrix_shifts=torch.zeros((4, 2, 3), dtype=torch.int32), **kw)
    with pytest.raises(ValueError, match="neighbor_ptr must be provided when using neighbor_list"):
        dftd3(pos, z, neighbor_list=nl, **kw)


def test_electrostatics_errors():
    from nvalchemiops.interactions.electrostatics import ewald_real_space, ewald_summation, pme_reciprocal_space
    from nvalchemiops.interactions.electrostatics.parameters import mesh_spacing_to_dimensions

    pos = torch.tensor([[2.0, 5, 5], [8.0, 5, 5]], dtype=torch.float64)
    q = torch.tensor([1.0, -1.0], dtype=torch.float64)
    cell = torch.eye(3, dtype=torch.float64).unsqueeze(0) * 10
    lst = dict(neighbor_list=torch.tensor([[0, 1], [1, 0]], dtype=torch.int32), neighbor_ptr=torch.tensor([0, 1, 2], dtype=torch.int32),
               neighbor_shifts=torch.zeros((2, 3), dtype=torch.int32))
    two = torch.tensor([0.3, 0.5], dtype=torch.float64)
    with pytest.raises(ValueError):
        pme_reciprocal_space(pos, q, cell, alpha=two, mesh_dimensions=(16, 16, 16))
    with pytest.raises(TypeError):
        pme_reciprocal_space(pos, q, cell, alpha="invalid", mesh_dimensions=(16, 16, 16))
    with pytest.raises(ValueError, match="Either mesh_dimensions or mesh_spacing must be provided"):
        pme_reciprocal_space(pos, q, cell, alpha=0.3, mesh_dimensions=None, mesh_spacing=None)
    with pytest.raises(ValueError):
        ewald_real_space(pos, q, cell, two[:1])
    with pytest.raises(ValueError):
        ewald_summation(pos, q, cell, two, **lst)
    with pytest.raises(TypeError):
        ewald_summation(pos, q, cell, alpha="invalid", **lst)
    with pytest.raises(ValueError):
        mesh_spacing_to_dimensions(torch.stack([torch.eye(3) * 20, torch.eye(3) * 30]), mesh_spacing=torch.tensor([0.5, 0.5, 0.5]))


def test_d3_parameters_container():
    """`D3Parameters` behaviours of test_dftd3.py:245-416: valid construction, custom interpolation mesh, `.to()` for device / dtype, float64 tables."""
    from nvalchemiops.interactions.dispersion import D3Parameters

    r = torch.rand
    p = D3Parameters(rcov=r(10), r4r2=r(10), c6ab=r(10, 10, 5, 5), cn_ref=r(10, 10, 5, 5))
    assert p.max_z == 9 and p.device == torch.device("cpu") and p.interp_mesh == 5
    q = D3Parameters(rcov=r(6), r4r2=r(6), c6ab=r(6, 6, 3, 3), cn_ref=r(6, 6, 3, 3), interp_mesh=3)
    assert q.interp_mesh == 3 and q.c6ab.shape == (6, 6, 3, 3)
    assert p.to(device="cpu").rcov.device == torch.device("cpu")
    d = D3Parameters(rcov=r(10, dtype=torch.float64), r4r2=r(10, dtype=torch.float64), c6ab=r(10, 10, 5, 5, dtype=torch.float64),
                     cn_ref=r(10, 10, 5, 5, dtype=torch.float64))
    assert d.rcov.dtype == torch.float64
    f = d.to(device="cpu", dtype=torch.float32)
    assert all(t.dtype == torch.float32 for t in (f.rcov, f.r4r2, f.c6ab, f.cn_ref)) and f.device == torch.device("cpu")


def test_d3_empty_systems():
    """Empty inputs return before any device work (test_dftd3.py:634-715, :1673-1770): energy per system from batch_idx, empty F / CN."""
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3

    r = torch.rand
    p = D3Parameters(rcov=r(10), r4r2=r(10), c6ab=r(10, 10, 5, 5), cn_ref=r(10, 10, 5, 5))
    pos, z, nm = torch.empty((0, 3)), torch.empty((0,), dtype=torch.int32), torch.empty((0, 5), dtype=torch.int32)
    kw = dict(a1=0.4, a2=4.0, s8=0.8, d3_params=p, neighbor_matrix=nm)
    for bi, nsys in ((None, 1), (torch.empty((0,), dtype=torch.int32), 1), (torch.tensor([0, 1, 2], dtype=torch.int32), 3)):
        e, f, cn = dftd3(pos, z, batch_idx=bi, **kw)
        assert e.shape == (nsys,) and float(e.abs().sum()) == 0.0 and f.shape == (0, 3) and cn.shape == (0,)
        assert e.dtype == torch.float32 and f.dtype == torch.float32



def test_electrostatics_empty_systems():
    """No atoms: every electrostatics entry point returns empty per-atom results without device work (test_pme.py:350, test_ewald.py,
    test_coulomb.py:1951-2175 shapes)."""
    from nvalchemiops.interactions.electrostatics import (ewald_real_space, ewald_reciprocal_space, ewald_summation, particle_mesh_ewald,
                                                          pme_reciprocal_space)
    from nvalchemiops.interactions.electrostatics.coulomb import coulomb_energy_forces

    pos, q = torch.zeros((0, 3), dtype=torch.float64), torch.zeros(0, dtype=torch.float64)
    cell = torch.eye(3, dtype=torch.float64).unsqueeze(0) * 10
    al = torch.tensor([0.3], dtype=torch.float64)
    nm, sh = torch.zeros((0, 4), dtype=torch.int32), torch.zeros((0, 4, 3), dtype=torch.int32)
    lst = dict(neighbor_matrix=nm, neighbor_matrix_shifts=sh)
    outs = [pme_reciprocal_space(pos, q, cell[0], alpha=0.3, mesh_dimensions=(16, 16, 16), compute_forces=True),
            particle_mesh_ewald(pos, q, cell[0], alpha=0.3, mesh_dimensions=(16, 16, 16), compute_forces=True, **lst),
            ewald_real_space(pos, q, cell, al, compute_forces=True, **lst),
            ewald_reciprocal_space(pos, q, cell, torch.zeros((5, 3), dtype=torch.float64), al, compute_forces=True),
            ewald_summation(pos, q, cell, alpha=0.3, k_cutoff=2.0, compute_forces=True, **lst),
            coulomb_energy_forces(pos, q, cell, 5.0, 0.0, **lst)]
    for e, f in outs:
        assert e.shape == (0,) and f.shape == (0, 3) and e.dtype == torch.float64

"""Parameter estimation is host logic in plain torch (interactions/electrostatics/parameters.py:68-437): the input forms and return
types the reference's tests exercise (test/interactions/electrostatics/test_parameters.py:38-615), on CPU tensors."""
import math

import pytest
import torch

from nvalchemiops.interactions.electrostatics.parameters import (EwaldParameters, PMEParameters, estimate_ewald_parameters,
                                                                 estimate_pme_mesh_dimensions, estimate_pme_parameters,
                                                                 mesh_spacing_to_dimensions)


def _pow2(n):
    return n > 0 and (n & (n - 1)) == 0


def test_single_and_batch_shapes_and_types():
    pos = torch.randn(100, 3)
    cell = torch.eye(3).unsqueeze(0) * 20.0
    p = estimate_pme_parameters(pos, cell, accuracy=1e-6)
    assert isinstance(p, PMEParameters) and p.alpha.shape == (1,) and p.real_space_cutoff.shape == (1,) and p.mesh_spacing.shape == (1, 3)
    assert isinstance(p.mesh_dimensions, tuple) and len(p.mesh_dimensions) == 3 and all(isinstance(d, int) and _pow2(d) for d in p.mesh_dimensions)
    lengths = torch.norm(cell, dim=2)
    assert torch.allclose(p.mesh_spacing, lengths / torch.tensor(p.mesh_dimensions, dtype=lengths.dtype))
    e = estimate_ewald_parameters(pos, cell, accuracy=1e-6)
    assert isinstance(e, EwaldParameters) and e.alpha.shape == (1,) and e.real_space_cutoff.shape == (1,) and e.reciprocal_space_cutoff.shape == (1,)
    assert torch.allclose(p.alpha, e.alpha) and torch.allclose(p.real_space_cutoff, e.real_space_cutoff)  # :462 consistency
    # 2-D cell is accepted (:277, :390)
    p2 = estimate_pme_parameters(pos, cell[0], accuracy=1e-6)
    assert torch.allclose(p2.alpha, p.alpha) and p2.mesh_dimensions == p.mesh_dimensions
    cells = torch.stack([torch.eye(3) * 20.0, torch.eye(3) * 30.0])
    bi = torch.tensor([0] * 50 + [1] * 50, dtype=torch.int32)
    pb = estimate_pme_parameters(pos, cells, batch_idx=bi, accuracy=1e-6)
    assert pb.alpha.shape == (2,) and pb.real_space_cutoff.shape == (2,) and pb.mesh_spacing.shape == (2, 3)
    assert isinstance(pb.mesh_dimensions, tuple) and all(isinstance(d, int) for d in pb.mesh_dimensions)
    assert torch.all(pb.mesh_spacing[1] > pb.mesh_spacing[0])  # same mesh, larger cell (:476)
    eb = estimate_ewald_parameters(pos, cells, batch_idx=bi, accuracy=1e-6)
    assert eb.alpha.shape == (2,) and eb.alpha[0] > eb.alpha[1]  # larger cell, smaller alpha (:164)


def test_trends_and_invariants():
    cell = torch.eye(3).unsqueeze(0) * 20.0
    a100 = estimate_ewald_parameters(torch.randn(100, 3), cell).alpha
    a800 = estimate_ewald_parameters(torch.randn(800, 3), cell).alpha
    assert 0.01 < float(a100) < 2.0 and a800 > a100  # (:152, :177)
    lo, hi = estimate_ewald_parameters(torch.randn(100, 3), cell, accuracy=1e-4), estimate_ewald_parameters(torch.randn(100, 3), cell, accuracy=1e-8)
    assert hi.real_space_cutoff > lo.real_space_cutoff and hi.reciprocal_space_cutoff > lo.reciprocal_space_cutoff  # (:190)
    # r_c * k_c = -2 ln(eps): independent of the cell (:207)
    prods = []
    for size in (10.0, 20.0, 40.0):
        e = estimate_ewald_parameters(torch.randn(100, 3), torch.eye(3).unsqueeze(0) * size, accuracy=1e-6)
        prods.append(float(e.real_space_cutoff * e.reciprocal_space_cutoff))
    assert max(prods) - min(prods) < 1e-3 * prods[0] and abs(prods[0] + 2.0 * math.log(1e-6)) < 1e-4


def test_mesh_dimension_helpers():
    cell = torch.eye(3).unsqueeze(0) * 20.0
    alpha = torch.tensor([0.3])
    d = estimate_pme_mesh_dimensions(cell, alpha, accuracy=1e-6)
    assert isinstance(d, tuple) and len(d) == 3 and all(_pow2(v) for v in d)
    assert estimate_pme_mesh_dimensions(cell, torch.tensor([0.6]), 1e-6)[0] >= d[0]  # larger alpha, more points (:316)
    assert estimate_pme_mesh_dimensions(cell, alpha, 1e-8)[0] >= d[0]               # higher accuracy (:333)
    rect = torch.diag(torch.tensor([10.0, 20.0, 40.0])).unsqueeze(0)
    r = estimate_pme_mesh_dimensions(rect, alpha, 1e-6)
    assert r[0] <= r[1] <= r[2]                                                       # (:344)
    cells = torch.stack([torch.eye(3) * 20.0, torch.eye(3) * 40.0])
    both = estimate_pme_mesh_dimensions(cells, torch.tensor([0.3, 0.3]), 1e-6)
    assert both == estimate_pme_mesh_dimensions(cells[1:], alpha, 1e-6)               # batch uses the maximum (:355-388)
    assert estimate_pme_mesh_dimensions(cell[0], alpha, 1e-6) == d                    # 2-D cell (:390)
    s = mesh_spacing_to_dimensions(cell, mesh_spacing=0.5)
    assert isinstance(s, tuple) and all(isinstance(v, int) and _pow2(v) for v in s)   # (:518-540)
    assert mesh_spacing_to_dimensions(cell, mesh_spacing=0.25)[0] >= s[0]             # (:541)
    assert len(mesh_spacing_to_dimensions(cells, mesh_spacing=torch.tensor([0.5, 0.5]))) == 3                          # (:564)
    assert len(mesh_spacing_to_dimensions(cells, mesh_spacing=torch.tensor([[0.5, 0.4, 0.3], [0.6, 0.5, 0.4]]))) == 3  # (:577)
    assert mesh_spacing_to_dimensions(cell[0], mesh_spacing=0.5) == s                                                  # (:603)
    with pytest.raises(ValueError):
        mesh_spacing_to_dimensions(cells, mesh_spacing=torch.tensor([0.5, 0.5, 0.5]))                                  # (:591)


def test_k_vector_generators_forms_and_gradients():
    """Shapes, safe k^2 and autograd through the cell, as test/interactions/electrostatics/test_kvectors.py:64-290 checks them."""
    from nvalchemiops.interactions.electrostatics import generate_k_vectors_ewald_summation, generate_k_vectors_pme

    cell = torch.eye(3, dtype=torch.float64).unsqueeze(0) * 10.0
    k = generate_k_vectors_ewald_summation(cell, k_cutoff=8.0)
    assert k.ndim == 2 and k.shape[1] == 3 and k.shape[0] > 0
    kb = generate_k_vectors_ewald_summation(cell.expand(3, -1, -1).contiguous(), k_cutoff=8.0)
    assert kb.shape == (3, k.shape[0], 3)
    assert generate_k_vectors_ewald_summation(cell, k_cutoff=10.0).shape[0] > generate_k_vectors_ewald_summation(cell, k_cutoff=5.0).shape[0]
    kv, k2 = generate_k_vectors_pme(cell, (16, 16, 16))
    assert kv.shape == (16, 16, 9, 3) and k2.shape == (16, 16, 9) and bool((k2 > 0).all())
    assert float(torch.norm(kv[0, 0, 0])) < 1e-10 and float(k2[0, 0, 0]) > 0
    for dims in ((8, 8, 8), (16, 32, 64), (12, 10, 14)):
        kv, k2 = generate_k_vectors_pme(cell, dims)
        assert kv.shape == (dims[0], dims[1], dims[2] // 2 + 1, 3) and k2.shape == kv.shape[:-1]
    c = cell.clone().requires_grad_(True)
    generate_k_vectors_ewald_summation(c, k_cutoff=8.0).sum().backward()
    assert c.grad is not None and bool(torch.isfinite(c.grad).all())
    c = cell.clone().requires_grad_(True)
    kv, k2 = generate_k_vectors_pme(c, (16, 16, 16))
    (kv.sum() + k2.sum()).backward()
    assert c.grad is not None and bool(torch.isfinite(c.grad).all())

"""`torch.compile` over the public functional API, as the reference's own tests do with a bare `@torch.compile`
(test/neighborlist/test_cell_list.py:599-760, test_naive.py:1200-1300, test_batch_cell_list.py, test/interactions/dispersion/
test_dftd3.py:1204-1330): the entry points are `torch.compiler.disable`d eager islands (nvalchemiops._capi.eager), the torch code
around them is compiled by Inductor, and results equal the eager call.  (The fullgraph seam -- `torch.ops.nvalchemiops.*` -- is
covered by tests/test_nlist_gpu.py::test_custom_ops_and_graph_capture.)"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import systems as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x.shape == y.shape and x.dtype == y.dtype and x.device == y.device
        assert torch.equal(x, y) if x.dtype in (torch.int32, torch.int64, torch.bool) else torch.allclose(x, y, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("half_fill", [False, True])
def test_naive_neighbor_list_compiles(half_fill):
    from nvalchemiops.neighborlist import naive_neighbor_list

    pos, cell = S.random_box(50, 6.0, seed=31, dtype=np.float32)
    tp, tc, pbc = _t(pos), _t(cell).reshape(1, 3, 3), torch.ones((1, 3), dtype=torch.bool, device=DEV)

    @torch.compile
    def free(positions, cutoff, neighbor_matrix, num_neighbors, half_fill):
        return naive_neighbor_list(positions=positions, cutoff=cutoff, neighbor_matrix=neighbor_matrix, num_neighbors=num_neighbors,
                                   half_fill=half_fill)

    nm = torch.full((50, 100), 50, dtype=torch.int32, device=DEV)
    num = torch.zeros(50, dtype=torch.int32, device=DEV)
    free(tp, 3.0, nm, num, half_fill)
    assert int(num.sum()) > 0
    _same((nm, num), naive_neighbor_list(tp, 3.0, max_neighbors=100, half_fill=half_fill))

    @torch.compile
    def periodic(positions, cutoff, cell, pbc, half_fill):
        nm, num, sh = naive_neighbor_list(positions, cutoff, cell=cell, pbc=pbc, max_neighbors=64, half_fill=half_fill)
        return nm, num * 1, sh  # some torch work for Inductor on either side of the island

    _same(periodic(tp, 2.0, tc, pbc, half_fill), naive_neighbor_list(tp, 2.0, cell=tc, pbc=pbc, max_neighbors=64, half_fill=half_fill))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_build_and_query_cell_list_compile(dtype):
    from nvalchemiops.neighborlist import allocate_cell_list, build_cell_list, estimate_cell_list_sizes, query_cell_list

    pos, cell = S.random_box(400, 14.0, seed=32, dtype=dtype)
    tp, tc, pbc = _t(pos), _t(cell).reshape(1, 3, 3), torch.tensor([True, True, True], device=DEV)
    ncell, radius = estimate_cell_list_sizes(tc, pbc, 4.0)
    eager_cache = allocate_cell_list(400, ncell, radius, tp.device)
    build_cell_list(tp, 4.0, tc, pbc, *eager_cache)
    comp_cache = allocate_cell_list(400, ncell, radius, tp.device)

    @torch.compile
    def compiled_build(positions, cutoff, cell, pbc, cpd, rad, aps, a2c, apc, start, lst):
        build_cell_list(positions, cutoff, cell, pbc, cpd, rad, aps, a2c, apc, start, lst)

    compiled_build(tp, 4.0, tc, pbc, *comp_cache)
    _same(eager_cache, comp_cache)

    def outputs():
        return (torch.full((400, 96), 400, dtype=torch.int32, device=DEV), torch.zeros((400, 96, 3), dtype=torch.int32, device=DEV),
                torch.zeros(400, dtype=torch.int32, device=DEV))

    eo, co = outputs(), outputs()
    query_cell_list(tp, 4.0, tc, pbc, *eager_cache, *eo, False)

    @torch.compile
    def compiled_query(positions, cutoff, cell, pbc, cache, nm, sh, num):
        query_cell_list(positions, cutoff, cell, pbc, *cache, nm, sh, num, False)
        return num.sum()

    total = compiled_query(tp, 4.0, tc, pbc, comp_cache, *co)
    _same(eo, co)
    assert int(total) == int(eo[2].sum()) > 0


def test_batch_cell_list_and_dispatcher_compile():
    from nvalchemiops.neighborlist import batch_cell_list, neighbor_list

    p0, c0 = S.random_box(300, 12.0, seed=33, dtype=np.float32)
    p1, c1 = S.random_box(200, 10.0, seed=34, dtype=np.float32, triclinic=True)
    tp, tc = _t(np.concatenate([p0, p1])), _t(np.stack([c0, c1]))
    pbc = torch.ones((2, 3), dtype=torch.bool, device=DEV)
    bi = torch.tensor([0] * 300 + [1] * 200, dtype=torch.int32, device=DEV)

    @torch.compile
    def f(p, c, b, i):
        return batch_cell_list(p, 3.5, c, b, i, max_neighbors=96)

    _same(f(tp, tc, pbc, bi), batch_cell_list(tp, 3.5, tc, pbc, bi, max_neighbors=96))

    @torch.compile
    def g(p, c, b, i):
        lst, ptr, sh = neighbor_list(p, 3.5, cell=c, pbc=b, batch_idx=i, method="batch_cell_list", max_neighbors=96, return_neighbor_list=True)
        return lst, ptr, sh

    _same(g(tp, tc, pbc, bi), neighbor_list(tp, 3.5, cell=tc, pbc=pbc, batch_idx=bi, method="batch_cell_list", max_neighbors=96,
                                            return_neighbor_list=True))


def test_dftd3_and_pme_compile():
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
    from nvalchemiops.neighborlist import cell_list

    t = O.d3_test_tables(17)
    params = D3Parameters(rcov=_t(t["rcov"]), r4r2=_t(t["r4r2"]), c6ab=_t(t["c6ab"]), cn_ref=_t(t["cn_ref"]))
    pos, cell = S.random_box(300, 22.0, seed=35, dtype=np.float32)
    g = np.random.default_rng(0)
    numbers = _t(g.choice([1, 6, 7, 8], 300).astype(np.int32))
    tp, tc, pbc = _t(pos), _t(cell).reshape(1, 3, 3), torch.tensor([True, True, True], device=DEV)
    nm, num, sh = cell_list(tp, 10.0, tc, pbc, max_neighbors=160)
    assert int(num.max()) <= 160
    kw = dict(a1=0.4, a2=4.0, s8=0.8, d3_params=params, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=tc, fill_value=300)
    compiled = torch.compile(dftd3)  # test_dftd3.py:1228
    e, f, cn = compiled(positions=tp, numbers=numbers, **kw)
    e0, f0, cn0 = dftd3(positions=tp, numbers=numbers, **kw)
    assert e.shape == (1,) and f.shape == (300, 3) and cn.shape == (300,) and f.dtype == torch.float32
    assert torch.allclose(e, e0, rtol=1e-6) and torch.allclose(f, f0, rtol=1e-5, atol=1e-9) and torch.allclose(cn, cn0, rtol=1e-6)

    q = torch.randn(300, dtype=torch.float64, device=DEV)
    q -= q.mean()
    pd, cd = tp.double(), tc.double()
    nm, num, sh = cell_list(pd, 6.0, cd, pbc, max_neighbors=64)

    @torch.compile
    def step(p, q, c, nm, sh):
        e, f = particle_mesh_ewald(p, q, c, alpha=0.4, mesh_dimensions=(16, 16, 16), spline_order=4, neighbor_matrix=nm,
                                   neighbor_matrix_shifts=sh, compute_forces=True)
        return e.sum() * 2.0, f

    et, ft = step(pd, q, cd, nm, sh)
    e0, f0 = particle_mesh_ewald(pd, q, cd, alpha=0.4, mesh_dimensions=(16, 16, 16), spline_order=4, neighbor_matrix=nm,
                                 neighbor_matrix_shifts=sh, compute_forces=True)
    assert torch.allclose(et, e0.sum() * 2.0, rtol=1e-12) and torch.allclose(ft, f0, rtol=1e-10, atol=1e-12)


# ---- the `alchemiops::*` custom-op seam of the electrostatics path (VERDICT r1 missing #1 / #2) ------------------------------
def _pme_inputs(dtype=torch.float64, n=120, batched=False):
    g = torch.Generator().manual_seed(5)
    box = 11.0
    cell = torch.tensor([[box, 0, 0], [0.1 * box, 0.95 * box, 0], [0.05 * box, -0.1 * box, 1.05 * box]], dtype=dtype)
    pos = (torch.rand((n, 3), generator=g, dtype=dtype) @ cell)
    q = torch.randn(n, generator=g, dtype=dtype)
    q -= q.mean()
    return pos.to(DEV), cell.to(DEV), q.to(DEV)


@pytest.mark.parametrize("batched", [False, True])
def test_particle_mesh_ewald_fullgraph_compile(batched):
    """`torch.compile(particle_mesh_ewald, fullgraph=True)`: the whole composition (real-space op, spread, FFTs, Green function,
    gather, corrections, force gather) is captured in ONE graph through the `alchemiops::*` ops and their fake implementations
    (which return the true dtype), and equals the eager (fused-kernel) result."""
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
    from nvalchemiops.neighborlist import batch_cell_list, cell_list

    torch._dynamo.reset()
    pos, cell, q = _pme_inputs()
    n = pos.shape[0]
    if batched:
        pos, q = torch.cat([pos, pos + 0.3]), torch.cat([q, -q])
        cells = torch.stack([cell, cell * 1.05])
        bi = torch.repeat_interleave(torch.arange(2, dtype=torch.int32, device=DEV), n)
        nm, num, sh = batch_cell_list(pos, 5.0, cells, torch.ones((2, 3), dtype=torch.bool, device=DEV), bi, max_neighbors=96)
        kw = dict(cell=cells, alpha=torch.tensor([0.4, 0.38], dtype=torch.float64, device=DEV), batch_idx=bi)
    else:
        nm, num, sh = cell_list(pos, 5.0, cell, torch.tensor([True] * 3, device=DEV), max_neighbors=96)
        kw = dict(cell=cell, alpha=0.4)
    assert int(num.max()) <= 96
    kw.update(mesh_dimensions=(16, 16, 16), spline_order=4, neighbor_matrix=nm, neighbor_matrix_shifts=sh, compute_forces=True,
              compute_charge_gradients=True)
    eager = particle_mesh_ewald(pos, q, **kw)
    compiled = torch.compile(particle_mesh_ewald, fullgraph=True, backend="aot_eager")(pos, q, **kw)
    for a, b in zip(compiled, eager):
        assert a.dtype == b.dtype == torch.float64 and a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=1e-9, atol=1e-11)
    # fp32 inputs: the fake implementations must report float32 (the reference's report float64: SURVEY Appendix B.11)
    kw32 = {k: (v.float() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in kw.items()}
    out32 = torch.compile(particle_mesh_ewald, fullgraph=True, backend="aot_eager")(pos.float(), q.float(), **kw32)
    assert all(o.dtype == torch.float32 for o in out32)
    torch.testing.assert_close(out32[0].double(), eager[0], rtol=2e-4, atol=2e-4)


def test_compiled_pme_is_differentiable():
    """Gradients through the compiled graph (aot_eager traces the registered backward formulas): d(sum E)/d positions = -forces."""
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
    from nvalchemiops.neighborlist import cell_list

    torch._dynamo.reset()
    pos, cell, q = _pme_inputs()
    nm, num, sh = cell_list(pos, 5.0, cell, torch.tensor([True] * 3, device=DEV), max_neighbors=96)
    kw = dict(cell=cell, alpha=0.4, mesh_dimensions=(20, 20, 20), spline_order=5, neighbor_matrix=nm, neighbor_matrix_shifts=sh)
    p = pos.clone().requires_grad_(True)
    e = torch.compile(particle_mesh_ewald, fullgraph=True, backend="aot_eager")(p, q, **kw)
    e.sum().backward()
    _, f = particle_mesh_ewald(pos, q, compute_forces=True, **kw)
    torch.testing.assert_close(-p.grad, f, rtol=1e-3, atol=1e-4)  # reference tolerance for explicit forces vs autograd (test_pme.py:1458)


def test_ewald_and_coulomb_fullgraph_compile():
    """The explicit-k reciprocal sum, the real-space sum and the cut-off Coulomb ops trace as ONE graph each through their registered
    ops (`alchemiops::_ewald_reciprocal_space_*`, `alchemiops::_ewald_real_space_*`, `nvalchemiops::_coulomb_*`) and equal eager;
    gradients through the compiled graphs equal the eager gradients."""
    from nvalchemiops.interactions.electrostatics import (ewald_real_space, ewald_reciprocal_space,
                                                          generate_k_vectors_ewald_summation)
    from nvalchemiops.interactions.electrostatics.coulomb import coulomb_energy, coulomb_energy_forces
    from nvalchemiops.neighborlist import cell_list

    torch._dynamo.reset()
    pos, cell, q = _pme_inputs()
    al = torch.tensor([0.4], dtype=torch.float64, device=DEV)
    kv = generate_k_vectors_ewald_summation(cell.unsqueeze(0), 3.0)
    nm, num, sh = cell_list(pos, 5.0, cell, torch.tensor([True] * 3, device=DEV), max_neighbors=96)
    nl, ptr, lsh = cell_list(pos, 5.0, cell, torch.tensor([True] * 3, device=DEV), return_neighbor_list=True)
    cases = [
        (ewald_reciprocal_space, (pos, q, cell.unsqueeze(0), kv, al), dict(compute_forces=True, compute_charge_gradients=True)),
        (ewald_reciprocal_space, (pos, q, cell.unsqueeze(0), kv, al), dict(compute_charge_gradients=True)),
        (ewald_real_space, (pos, q, cell.unsqueeze(0), al), dict(neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=lsh, compute_forces=True)),
        (coulomb_energy_forces, (pos, q, cell.unsqueeze(0), 5.0, 0.3), dict(neighbor_matrix=nm, neighbor_matrix_shifts=sh)),
        (coulomb_energy, (pos, q, cell.unsqueeze(0), 5.0, 0.0), dict(neighbor_list=nl, neighbor_ptr=ptr, neighbor_shifts=lsh)),
    ]
    for fn, args, kw in cases:
        eager = fn(*args, **kw)
        comp = torch.compile(fn, fullgraph=True, backend="aot_eager")(*args, **kw)
        eager, comp = (eager if isinstance(eager, tuple) else (eager,)), (comp if isinstance(comp, tuple) else (comp,))
        assert len(eager) == len(comp)
        for a, b in zip(comp, eager):
            assert a.dtype == b.dtype and a.shape == b.shape
            torch.testing.assert_close(a, b, rtol=1e-10, atol=1e-12)
    # gradients: compiled == eager for every differentiable input
    for fn, args, kw, n_t in ((ewald_reciprocal_space, (pos, q, cell.unsqueeze(0), kv, al), {}, 5),
                              (coulomb_energy, (pos, q, cell.unsqueeze(0), 5.0, 0.3), dict(neighbor_matrix=nm, neighbor_matrix_shifts=sh), 3)):
        grads = []
        for f in (fn, torch.compile(fn, fullgraph=True, backend="aot_eager")):
            leaves = [a.clone().requires_grad_(True) for a in args[:n_t]]
            out = f(*leaves, *args[n_t:], **kw)
            w = torch.linspace(0.5, 1.5, out.numel(), dtype=out.dtype, device=DEV)
            grads.append(torch.autograd.grad((out * w).sum(), leaves))
        for a, b in zip(*grads):
            torch.testing.assert_close(a, b, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("batched", [False, True])
def test_compiled_pme_runs_the_fused_reciprocal_op(batched):
    """Round 5: under `torch.compile` the reciprocal sum is ONE op (`nvalchemiops::pme_reciprocal_fused`: the inference kernels forward, the
    hand-written adjoint behind `::pme_reciprocal_fused_backward`) instead of the op-by-op composition.  Compiled values and gradients of a
    loss on energies, explicit forces and charge gradients equal the eager fused node's; the op really ran (launch counters)."""
    from nvalchemiops import _eops as E
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald, pme_reciprocal_space
    from nvalchemiops.neighborlist import batch_cell_list, cell_list

    torch._dynamo.reset()
    pos, cell, q = _pme_inputs()
    n = pos.shape[0]
    if batched:
        pos, q = torch.cat([pos, pos + 0.3]), torch.cat([q, -q])
        cells = torch.stack([cell, cell * 1.05])
        bi = torch.repeat_interleave(torch.arange(2, dtype=torch.int32, device=DEV), n)
        nm, num, sh = batch_cell_list(pos, 5.0, cells, torch.ones((2, 3), dtype=torch.bool, device=DEV), bi, max_neighbors=96)
        base = dict(cell=cells, alpha=torch.tensor([0.4, 0.38], dtype=torch.float64, device=DEV), batch_idx=bi)
    else:
        nm, num, sh = cell_list(pos, 5.0, cell, torch.tensor([True] * 3, device=DEV), max_neighbors=96)
        base = dict(cell=cell, alpha=torch.tensor([0.4], dtype=torch.float64, device=DEV))
    gen = torch.Generator(device=DEV).manual_seed(3)
    we, wc = (torch.randn(pos.shape[0], dtype=torch.float64, device=DEV, generator=gen) for _ in range(2))
    wf = torch.randn(pos.shape[0], 3, dtype=torch.float64, device=DEV, generator=gen)

    def run(fn, compiled, **extra):
        p, v = pos.clone().requires_grad_(True), q.clone().requires_grad_(True)
        c, a = base["cell"].clone().requires_grad_(True), base["alpha"].clone().requires_grad_(True)
        kw = dict(cell=c, alpha=a, mesh_dimensions=(16, 16, 16), spline_order=4, compute_forces=True, compute_charge_gradients=True, **extra)
        if batched:
            kw["batch_idx"] = bi
        f_ = torch.compile(fn, fullgraph=True, backend="aot_eager") if compiled else fn
        e, f, cg = f_(p, v, **kw)
        loss = (e * we).sum() + (f * wf).sum() + (cg * wc).sum()
        return (e.detach(), f.detach(), cg.detach()) + torch.autograd.grad(loss, (p, v, c, a))

    for fn, extra in ((pme_reciprocal_space, {}), (particle_mesh_ewald, dict(neighbor_matrix=nm, neighbor_matrix_shifts=sh))):
        before = list(E.PME_FUSED_OP_CALLS)
        got = run(fn, True, **extra)
        assert E.PME_FUSED_OP_CALLS[0] > before[0] and E.PME_FUSED_OP_CALLS[1] > before[1], (fn.__name__, before, E.PME_FUSED_OP_CALLS)
        want = run(fn, False, **extra)
        for k, (a, b) in enumerate(zip(got, want)):
            assert a.shape == b.shape and a.dtype == b.dtype, (fn.__name__, k)
            torch.testing.assert_close(a, b, rtol=1e-9, atol=1e-10)

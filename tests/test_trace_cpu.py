"""`torch.compile(fullgraph=True)` over the PUBLIC neighbour-list / DFT-D3 entry points, checked without a GPU.

The reference's wrappers are thin Python over mutation-annotated `torch.library.custom_op`s (neighborlist/cell_list.py:725-736, 892-895,
1037-1192; interactions/dispersion/dftd3.py:1792-1796) and its example compiles an MD step over them with
`@torch.compile(mode="default", fullgraph=True)` (examples/neighborlist/04_neighbors_list_torch_compile_performance.py:323-346).
Here the same public functions are `_capi.hybrid`: traced bodies whose launches become `torch.ops.nvalchemiops.*` calls.

Without a device the compiled graph cannot RUN -- the ops refuse CPU tensors (`NativeLibraryError`, raised by the op at run time).  That
is exactly what these tests wait for: reaching the op's run-time refusal under `fullgraph=True` proves that Dynamo captured the whole
function in one graph, that AOTAutograd functionalised the mutating ops, and (for the Inductor case) that Inductor generated code around
them.  The numbers are checked on the GPU (tests/test_fullgraph_gpu.py).
"""
import pytest
import torch
from torch._dynamo.backends.common import aot_autograd

from nvalchemiops._capi import NativeLibraryError
from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
from nvalchemiops.neighborlist import (allocate_cell_list, batch_build_cell_list, batch_cell_list, batch_naive_neighbor_list,
                                       batch_naive_neighbor_list_dual_cutoff, batch_query_cell_list, build_cell_list, cell_list,
                                       cell_list_needs_rebuild, naive_neighbor_list, naive_neighbor_list_dual_cutoff, neighbor_list,
                                       neighbor_list_needs_rebuild, query_cell_list)
from tests import systems as S

N, M = 64, 32


def _system():
    g = torch.Generator().manual_seed(5)
    pos = torch.rand(N, 3, generator=g) * 10
    cell = (torch.eye(3) * 10).reshape(1, 3, 3)
    pbc = torch.tensor([[True, True, True]])
    bi = torch.zeros(N, dtype=torch.int32)
    bptr = torch.tensor([0, N], dtype=torch.int32)
    return pos, cell, pbc, bi, bptr


def _outputs(fill=N):
    return (torch.full((N, M), fill, dtype=torch.int32), torch.zeros((N, M, 3), dtype=torch.int32), torch.zeros(N, dtype=torch.int32))


def _recording_backend(graphs):
    def keep(gm, example_inputs):
        graphs.append(gm)
        return gm.forward

    return aot_autograd(fw_compiler=keep)


def _ops_in(graphs):
    found = []
    for gm in graphs:
        for node in gm.graph.nodes:
            if node.op == "call_function" and "auto_functionalized" in str(node.target):
                found.append(str(node.args[0]))
            elif node.op == "call_function" and "nvalchemiops" in str(node.target):
                found.append(str(node.target))
    return found


def _traced(fn, *args, backend=None):
    """Compile `fn` fullgraph, run it on CPU tensors and expect the op's own run-time refusal; returns the op names of the one graph."""
    graphs = []
    compiled = torch.compile(fn, fullgraph=True, backend=backend or _recording_backend(graphs))
    with pytest.raises(NativeLibraryError, match="ROCm devices only"):
        compiled(*args)
    assert backend is not None or len(graphs) == 1, "the function must be captured as ONE graph"
    return _ops_in(graphs)


def test_reference_compiled_md_step_traces_fullgraph_under_inductor():
    """The low-level MD step of the reference's example 04 (:323-399): build -> refill -> query -> torch forces -> integrate -> rebuild
    -> query, plus the high-level `cell_list` with pre-allocated outputs of its first variant (:262-271), under Inductor."""
    pos, cell, pbc, _, _ = _system()
    cache = allocate_cell_list(N, 27, torch.ones(3, dtype=torch.int32), pos.device)
    nm, sh, num = _outputs(-1)

    def forces(positions):
        mask = nm >= 0
        dr = positions[nm.long()] - positions.unsqueeze(1) + sh.float() @ cell[0]
        r2 = (dr * dr).sum(-1).clamp(min=1e-10)
        s6 = (1.0 / r2) ** 3
        fmag = torch.where(mask, 24.0 / r2 * (s6 - 2 * s6 * s6), torch.zeros_like(r2))
        return (fmag.unsqueeze(-1) * dr).sum(1), torch.where(mask, 4 * (s6 * s6 - s6), torch.zeros_like(r2)).sum() * 0.5

    def md_step(positions, velocities):
        build_cell_list(positions, 3.0, cell, pbc, *cache)
        nm.fill_(-1), sh.fill_(0), num.fill_(0)
        query_cell_list(positions, 3.0, cell, pbc, *cache, nm, sh, num)
        f, _ = forces(positions)
        velocities = velocities + 0.5e-3 * f
        positions = (positions + 1e-3 * velocities) % cell[0, 0, 0]
        build_cell_list(positions, 3.0, cell, pbc, *cache)
        nm.fill_(-1), sh.fill_(0), num.fill_(0)
        query_cell_list(positions, 3.0, cell, pbc, *cache, nm, sh, num)
        f, potential = forces(positions)
        m2, n2, s2 = cell_list(positions, 3.0, cell, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num, fill_value=-1)
        return positions, velocities + 0.5e-3 * f, potential, n2.sum()

    ops = _traced(md_step, pos, torch.zeros(N, 3))
    assert ops.count("nvalchemiops.build_cell_list.default") == 2 and ops.count("nvalchemiops.query_cell_list.default") == 2
    assert "nvalchemiops.neighbor_search.default" in ops
    torch._dynamo.reset()
    _traced(md_step, pos, torch.zeros(N, 3), backend="inductor")  # Inductor's own code generation around the ops (C++ on this host)


def test_every_matrix_entry_point_traces_fullgraph():
    pos, cell, pbc, bi, bptr = _system()
    cases = {
        "cell_list": lambda p: neighbor_list(p, 3.0, cell=cell, pbc=pbc, method="cell_list", max_neighbors=M),
        "auto -> naive": lambda p: neighbor_list(p, 3.0, cell=cell, pbc=pbc, max_neighbors=M),
        "free-space naive": lambda p: naive_neighbor_list(p, 3.0, max_neighbors=M),
        "batch_cell_list": lambda p: batch_cell_list(p, 3.0, cell, pbc, bi, max_neighbors=M),
        "dispatcher batch": lambda p: neighbor_list(p, 3.0, cell=cell, pbc=pbc, batch_idx=bi, batch_ptr=bptr, max_neighbors=M),
        "batch_naive": lambda p: batch_naive_neighbor_list(p, 3.0, batch_idx=bi, batch_ptr=bptr, pbc=pbc, cell=cell, max_neighbors=M),
        "batch_naive free": lambda p: batch_naive_neighbor_list(p, 3.0, batch_idx=bi, batch_ptr=bptr, max_neighbors=M),
        "dual": lambda p: naive_neighbor_list_dual_cutoff(p, 2.0, 3.0, pbc=pbc, cell=cell, max_neighbors1=M),
        "batch dual": lambda p: batch_naive_neighbor_list_dual_cutoff(p, 2.0, 3.0, batch_idx=bi, batch_ptr=bptr, pbc=pbc, cell=cell,
                                                                      max_neighbors1=M),
    }
    expected = {"dual": "neighbor_search_dual", "batch dual": "neighbor_search_dual"}
    for name, fn in cases.items():
        torch._dynamo.reset()
        ops = _traced(fn, pos)
        assert any(expected.get(name, "neighbor_search") in o for o in ops), (name, ops)


def test_batch_build_query_and_rebuild_checks_trace_fullgraph():
    pos, cell, pbc, bi, _ = _system()
    bcache = allocate_cell_list(N, 27, torch.ones((1, 3), dtype=torch.int32), pos.device)
    nm, sh, num = _outputs()

    def batch_step(p):
        batch_build_cell_list(p, 3.0, cell, pbc, bi, *bcache)
        batch_query_cell_list(p, cell, pbc, 3.0, bi, *bcache, nm, sh, num)
        return num.sum()

    ops = _traced(batch_step, pos)
    assert "nvalchemiops.batch_build_cell_list.default" in ops and "nvalchemiops.batch_query_cell_list.default" in ops
    cache = allocate_cell_list(N, 27, torch.ones(3, dtype=torch.int32), pos.device)
    torch._dynamo.reset()
    assert any("_neighbor_list_needs_rebuild" in o for o in _traced(lambda p: neighbor_list_needs_rebuild(p, p + 0.1, 0.5) | False, pos))
    torch._dynamo.reset()
    assert any("_cell_list_needs_rebuild" in o for o in _traced(lambda p: cell_list_needs_rebuild(p, cache[3], cache[0], cell, pbc[0]), pos))


def test_dftd3_traces_fullgraph_in_both_list_formats():
    pos, cell, pbc, bi, _ = _system()
    t = S.d3_test_tables(17)
    params = D3Parameters(rcov=torch.tensor(t["rcov"]), r4r2=torch.tensor(t["r4r2"]), c6ab=torch.tensor(t["c6ab"]),
                          cn_ref=torch.tensor(t["cn_ref"]))
    numbers = torch.ones(N, dtype=torch.int32)
    nm, sh, _ = _outputs()
    bj = dict(a1=0.4, a2=4.0, s8=0.8)

    def matrix(p):
        e, f, cn, v = dftd3(p, numbers, d3_params=params, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=cell, num_systems=1,
                            compute_virial=True, **bj)
        return e.sum() + f.sum(), cn, v

    assert _traced(matrix, pos) == ["nvalchemiops.dftd3_nm.default"]
    lst, ptr = torch.zeros((2, 10), dtype=torch.int32), torch.zeros(N + 1, dtype=torch.int32)
    tables = {"rcov": params.rcov, "r4r2": params.r4r2, "c6ab": params.c6ab, "cn_ref": params.cn_ref}
    torch._dynamo.reset()
    assert _traced(lambda p: dftd3(p, numbers, d3_params=tables, neighbor_list=lst, neighbor_ptr=ptr, batch_idx=bi, num_systems=1, **bj),
                   pos) == ["nvalchemiops.dftd3_nl.default"]
    # the validation errors of the eager call survive tracing (dftd3.py:2668-2700)
    torch._dynamo.reset()
    with pytest.raises(Exception, match="Must provide either neighbor_matrix or neighbor_list"):
        torch.compile(dftd3, fullgraph=True, backend="aot_eager")(pos, numbers, 0.4, 4.0, 0.8)


def test_data_dependent_outputs_stay_eager_islands():
    """COO / CSR outputs have a device-decided length: one host read, hence a graph break -- as in the reference, whose conversion
    calls `.item()` (neighbor_utils.py:426).  A bare `@torch.compile` still runs them (tests/test_compile_gpu.py); fullgraph refuses."""
    pos, cell, pbc, _, _ = _system()
    with pytest.raises(Exception, match="disable|Unsupported|graph break"):
        torch.compile(lambda p: cell_list(p, 3.0, cell, pbc, return_neighbor_list=True), fullgraph=True, backend="aot_eager")(pos)

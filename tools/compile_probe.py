"""Probe: does bare torch.compile (inductor) run the public functions on the GPU box, and what does it cost?  (scratch tool)"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nvalchemi-toolkit-ops_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from nvalchemiops.neighborlist import naive_neighbor_list, cell_list, neighbor_list
from nvalchemiops.interactions.electrostatics import particle_mesh_ewald

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
pos = (torch.rand((600, 3), generator=g) * 12.0).to(dev)
cell = (torch.eye(3) * 12.0).reshape(1, 3, 3).to(dev)
pbc = torch.ones((1, 3), dtype=torch.bool, device=dev)

def run(name, fn, *a, backend=None, **k):
    t = time.time()
    try:
        c = torch.compile(fn) if backend is None else torch.compile(fn, backend=backend)
        out = c(*a, **k)
        torch.cuda.synchronize()
        ref = fn(*a, **k)
        same = all(torch.equal(x, y) if x.dtype in (torch.int32, torch.bool) else torch.allclose(x, y, rtol=1e-6, atol=1e-9) for x, y in zip(out, ref))
        print(f"{name} backend={backend}: ok same={same} {time.time()-t:.1f}s", flush=True)
    except Exception as e:  # noqa
        print(f"{name} backend={backend}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)

def f_naive(p):
    return naive_neighbor_list(p, 3.0, max_neighbors=64)
def f_cell(p, c, b):
    nm, num, sh = cell_list(p, 3.0, c, b, max_neighbors=64)
    return nm, num + 0, sh
def f_pme(p, q, c, nm, sh):
    e, f = particle_mesh_ewald(p, q, c, alpha=0.35, mesh_dimensions=(16, 16, 16), spline_order=4, neighbor_matrix=nm, neighbor_matrix_shifts=sh,
                               compute_forces=True)
    return e * 2.0, f
for be in ("eager", None):
    run("naive", f_naive, pos, backend=be)
    run("cell_list", f_cell, pos, cell, pbc, backend=be)
    nm, num, sh = cell_list(pos.double(), 5.0, cell.double(), pbc, max_neighbors=96)
    q = torch.randn(600, dtype=torch.float64, device=dev); q -= q.mean()
    run("pme", f_pme, pos.double(), q, cell.double(), nm, sh, backend=be)

"""Probe: reciprocal PME (E+F, order 5, fp64) wall per call, fused mesh solve vs hipFFT plans, over mesh size and batch."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops.interactions.electrostatics import pme as P
from nvalchemiops.interactions.electrostatics import pme_reciprocal_space
dev = "cuda:0"
def wall(f, it=30):
    for _ in range(4): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
g = torch.Generator(device="cpu").manual_seed(3)
for mesh, nsys, n in ((16, 1, 500), (32, 1, 4000), (32, 16, 32000), (32, 128, 256000), (64, 1, 30000), (64, 8, 100000), (128, 1, 100000), (128, 2, 200000)):
    L = 1.0 * mesh
    pos = (torch.rand(n, 3, generator=g, dtype=torch.float64) * L).to(dev)
    q = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
    cell = (torch.eye(3, dtype=torch.float64) * L).reshape(1, 3, 3).repeat(nsys, 1, 1).to(dev)
    bi = (torch.arange(n, device=dev) * nsys // n).to(torch.int32) if nsys > 1 else None
    al = torch.full((nsys,), 0.35, dtype=torch.float64, device=dev)
    row = []
    for solve in (True, False):
        P._MESH_SOLVE = solve
        for cf in (False, True):
            t = wall(lambda: pme_reciprocal_space(pos, q, cell if nsys > 1 else cell[0], alpha=al if nsys > 1 else 0.35, mesh_dimensions=(mesh,) * 3, spline_order=5,
                                                  batch_idx=bi, compute_forces=cf))
            row.append(f"{'solve' if solve else 'hipfft'} {'E+F' if cf else 'E'} {t:.3f}")
    print(f"mesh {mesh}^3 x {nsys} systems, {n} atoms: " + " | ".join(row), flush=True)

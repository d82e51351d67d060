"""Probe: reciprocal PME (E+F, order 4/5, fp64) wall per call over mesh size x atom count, tile pipeline vs atomic spread + per-atom gather.
Run with NVALCHEMIOPS_SPREAD_PATH=tile (tile pipeline wherever the mesh allows), =atomic and =auto (the shipped policy)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops.interactions.electrostatics import pme_reciprocal_space
dev = "cuda:0"
def wall(f, it=40):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
g = torch.Generator(device="cpu").manual_seed(3)
for mesh in (32, 48, 64, 96, 128):
    row = []
    for n in (1000, 8000, 32000, 100000):
        L = 1.0 * mesh
        pos = (torch.rand(n, 3, generator=g, dtype=torch.float64) * L).to(dev)
        q = torch.randn(n, generator=g, dtype=torch.float64).to(dev); q -= q.mean()
        cell = (torch.eye(3, dtype=torch.float64) * L).reshape(1, 3, 3).to(dev)
        for order in (4, 5):
            t = wall(lambda: pme_reciprocal_space(pos, q, cell, alpha=0.35, mesh_dimensions=(mesh,) * 3, spline_order=order, compute_forces=True))
            row.append(f"n={n} o{order} {t:.3f}")
    print(f"mesh {mesh}^3 ({(mesh // 8) ** 3} tiles): " + " | ".join(row), flush=True)

"""Probe: fused mesh solve vs hipFFT path vs oracle with torch's caching allocator handing out POISONED (NaN) blocks."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from oracle import oracle as O
from nvalchemiops.interactions.electrostatics import pme as P
from nvalchemiops.interactions.electrostatics import pme_reciprocal_space
from nvalchemiops import spline
dev = "cuda:0"
def poison():
    keep = [torch.full((n,), float("nan"), dtype=torch.float64, device=dev) for n in (1 << 8, 1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22)] * 3
    del keep
g = np.random.default_rng(5)
for path in ("tile", "auto"):
    spline._SPREAD_PATH = path
    for dims in ((8, 8, 8), (16, 8, 32), (8, 64, 16), (32, 32, 32), (64, 16, 128), (16, 256, 8)):
        for order in (4, 5):
            n = 300
            cell = np.array([[12.0, 0, 0], [2.4, 10.8, 0], [1.2, -1.8, 13.2]])
            pos = g.uniform(0, 1, (n, 3)) @ cell
            q = g.normal(size=n); q -= q.mean()
            tp, tq, tc = (torch.as_tensor(a, device=dev) for a in (pos, q, cell))
            res = {}
            for solve in (True, False):
                P._MESH_SOLVE = solve
                poison()
                e, f = pme_reciprocal_space(tp, tq, tc, 0.4, mesh_dimensions=dims, spline_order=order, compute_forces=True)
                res[solve] = (e.cpu().numpy(), f.cpu().numpy())
            with O.extended_splines():
                ref = O.pme_reciprocal_space(pos, q, cell, 0.4, dims, order, compute_forces=True)
            sc = np.abs(ref[0]).max()
            print(path, dims, order, "solve-ref %.2e  fft-ref %.2e  solve-fft %.2e" % (np.abs(res[True][0] - ref[0]).max() / sc, np.abs(res[False][0] - ref[0]).max() / sc,
                                                                              np.abs(res[True][0] - res[False][0]).max() / sc), flush=True)

"""Fused mesh solve (mixed radix, round 6) vs the hipFFT-plan path on meshes that are not powers of two, and on the headline mesh (regression check).
pme_reciprocal_space E + F, fp64 and fp32, HIP-event medians.   python tools/probe/r06_solve_nonpow2.py"""
import os, sys, statistics
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops.interactions.electrostatics import pme as P, pme_reciprocal_space
from tests import systems as S
dev = "cuda:0"
def med(fn, w=5, it=30):
    for _ in range(w): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(it):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return statistics.median(ts)
for n_atoms in (20000, 100000):
    pos, cell, q, _ = S.fcc_box(n_atoms, dtype=np.float64)
    for dims in ((96, 96, 96), (100, 100, 100), (120, 120, 120), (96, 100, 120), (128, 128, 128), (64, 64, 64), (80, 80, 80)):
        for dt in (torch.float64, torch.float32):
            tp, tq, tc = (torch.as_tensor(a, dtype=dt, device=dev) for a in (pos, q, cell))
            out = {}
            for solve in (True, False):
                P._MESH_SOLVE = solve
                fn = lambda: pme_reciprocal_space(tp, tq, tc, 0.35, mesh_dimensions=dims, spline_order=5, compute_forces=True)
                e, f = fn(); out[solve] = (e.clone(), f.clone(), med(fn))
            de = float((out[True][0] - out[False][0]).abs().max() / out[False][0].abs().max())
            df = float((out[True][1] - out[False][1]).abs().max() / out[False][1].abs().max())
            print(f"{n_atoms:7d} atoms mesh {dims} {str(dt)[6:]}: solve {out[True][2]:.4f} ms  plans {out[False][2]:.4f} ms   rel dE {de:.1e} dF {df:.1e}   fallbacks {len(P._FFT_FALLBACKS)}", flush=True)

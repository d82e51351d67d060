"""Timing aid: explicit-k Ewald reciprocal space (N3) over system sizes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from tests import systems as S
from nvalchemiops.interactions.electrostatics import ewald_reciprocal_space, generate_k_vectors_ewald_summation
dev = "cuda:0"
for n, kc in ((1000, 1.6), (8000, 1.2), (32000, 1.0)):
    pos, cell, q, _ = S.fcc_box(n, dtype=np.float64)
    tp, tc, tq = (torch.as_tensor(a, device=dev) for a in (pos, cell, q))
    kv = generate_k_vectors_ewald_summation(tc, kc)
    al = torch.tensor([0.35], dtype=torch.float64, device=dev)
    f = lambda: ewald_reciprocal_space(tp, tq, tc, kv, al, compute_forces=True)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"N={n} K={kv.shape[0]}: {dt * 1e3:.3f} ms  ({2 * n * kv.shape[0] / dt / 1e9:.1f} G sincos/s)", flush=True)

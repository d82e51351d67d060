"""Per-stage timings on the GPU (HIP events through torch), used while tuning.  Not the headline bench."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from tests import systems as S  # noqa: E402


def timeit(fn, warmup=3, iters=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    from nvalchemiops.neighborlist import batch_cell_list, cell_list

    dev = "cuda:0"
    pbc = torch.tensor([True] * 3, device=dev)
    for n, rc, m, dt in [(50000, 5.0, 928, np.float32), (50000, 5.0, 64, np.float32), (100000, 9.0, 256, np.float64),
                         (100000, 5.0, 64, np.float32), (32768, 5.0, 928, np.float32), (131072, 5.0, 928, np.float32)]:
        pos, cell, _, _ = S.fcc_box(n, dtype=dt)
        tp, tc = torch.as_tensor(pos, device=dev), torch.as_tensor(cell, device=dev)
        nm = torch.empty((n, m), dtype=torch.int32, device=dev)
        sh = torch.empty((n, m, 3), dtype=torch.int32, device=dev)
        num = torch.empty(n, dtype=torch.int32, device=dev)
        t = timeit(lambda: cell_list(tp, rc, tc, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num))
        byts = n * (3 * pos.itemsize + 4) + 16.0 * n * m
        print(f"nlist matrix N={n} rc={rc} M={m} {np.dtype(dt).name}: {t:.3f} ms  pairs={int(num.sum())}  {byts / t / 1e6:.1f} GB/s algorithmic", flush=True)
        t = timeit(lambda: cell_list(tp, rc, tc, pbc, return_neighbor_list=True))
        p = int(num.sum())
        print(f"nlist CSR    N={n} rc={rc}: {t:.3f} ms  {(n * (3 * pos.itemsize + 8) + 20.0 * p) / t / 1e6:.1f} GB/s algorithmic", flush=True)
    # big-cutoff list (D3 regime)
    pos, cell, _, _ = S.fcc_box(100000, dtype=np.float32)
    tp, tc = torch.as_tensor(pos * 1.8897261, device=dev), torch.as_tensor(cell * 1.8897261, device=dev)
    t = timeit(lambda: cell_list(tp, 40.0, tc, pbc, return_neighbor_list=True), warmup=1, iters=3)
    lst, nptr, _ = cell_list(tp, 40.0, tc, pbc, return_neighbor_list=True)
    print(f"nlist CSR N=100000 rc=40 Bohr: {t:.3f} ms pairs={lst.shape[1]} {20.0 * lst.shape[1] / t / 1e6:.1f} GB/s", flush=True)
    # batch
    parts, bis = [], []
    for s in range(256):
        p, c, _, _ = S.fcc_box(512, seed=1234 + s, dtype=np.float32)
        parts.append(p), bis.append(np.full(512, s, np.int32))
    tp = torch.as_tensor(np.concatenate(parts), device=dev)
    tc = torch.as_tensor(np.stack([c] * 256), device=dev)
    bi = torch.as_tensor(np.concatenate(bis), device=dev)
    pb = torch.ones((256, 3), dtype=torch.bool, device=dev)
    t = timeit(lambda: batch_cell_list(tp, 5.0, tc, pb, bi, max_neighbors=64))
    print(f"batch nlist 256x512 rc=5 M=64: {t:.3f} ms", flush=True)


if __name__ == "__main__":
    main()

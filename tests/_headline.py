"""The headline workload's D3 neighbour list built by the ORACLE (100 000-atom periodic box, rc = 40 Bohr, fp32, padded matrix M = 2560:
235 M directed pairs), shared by the list-parity test (tests/test_nlist_gpu.py) and the D3 full-size test (tests/test_d3_gpu.py).

The oracle's cell-list restatement runs through its -fopenmp build here (the serial loop nest needs ~2 minutes for 6.8e8 distance tests;
all host cores: seconds).  The per-pair arithmetic and the cutoff decision are the same code, only the slot order inside a row depends
on the thread schedule -- and rows are compared as SETS (tests/test_host_cpu.py::test_openmp_oracle_equals_serial_oracle pins the two
builds to each other)."""
import functools
import os

import numpy as np

from oracle import oracle as O
from tests import systems as S

N, CUTOFF, M = 100000, 40.0, 2560
BOHR = 1.8897261


@functools.lru_cache(maxsize=1)
def system():
    pos, cell, _, numbers = S.fcc_box(N, dtype=np.float32)
    pos, cell = (pos * BOHR).astype(np.float32), (cell * BOHR).astype(np.float32)
    return pos, cell, np.where(numbers == 6, 6, 8).astype(np.int32)


@functools.lru_cache(maxsize=1)
def oracle_list():
    """(neighbor_matrix[N,M], num_neighbors[N], shifts[N,M,3]) of the oracle; ~4 GB of host memory, kept for the session."""
    pos, cell, _ = system()
    with O.openmp(min(os.cpu_count() or 1, 64)):
        return O.cell_list(pos, CUTOFF, cell, [True] * 3, max_neighbors=M)


def row_keys(nm, sh, num, torch, device):
    """Per-row sorted int64 keys (j, S) of a padded matrix on `device`: key = j * 125 + shift code, padding = a key above every real one.
    Works on a row block; nm / sh / num may be numpy arrays (oracle) or tensors (product)."""
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a, device=device).to(dt)  # noqa: E731
    nm, sh, num = t(nm, torch.int64), t(sh, torch.int64), t(num, torch.int64)
    assert int(sh.abs().max()) <= 2
    key = nm * 125 + (sh[..., 0] + 2) * 25 + (sh[..., 1] + 2) * 5 + (sh[..., 2] + 2)
    pad = torch.arange(nm.shape[1], device=device)[None, :] >= num[:, None]
    key = torch.where(pad, torch.full_like(key, 1 << 62), key)
    return torch.sort(key, dim=1).values

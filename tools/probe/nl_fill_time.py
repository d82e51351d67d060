"""Probe: time of the headline 40-Bohr list fill (100 000 atoms, M = 2560, fp32, companion on) on K torch.empty buffer sets, library
HIP-event median per set.  For A/B of alternative builds of the list kernel (results are not checked here).
    python tools/probe/nl_fill_time.py [K] [reps]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops import _capi as C  # noqa: E402
from nvalchemiops.neighborlist import _engine as E  # noqa: E402
from nvalchemiops.neighborlist import cell_list  # noqa: E402
from tests import systems as S  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 9
dev = torch.device("cuda:0")
BOHR = 1.8897261246
n, M = 100000, 2560
pos, cell, q, numbers = S.fcc_box(n, seed=1234, dtype=np.float64)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
p32, c32 = t((pos * BOHR).astype(np.float32)), t((cell * BOHR).astype(np.float32))
pbc = torch.tensor([True] * 3, device=dev)
num = torch.empty(n, dtype=torch.int32, device=dev)
E._PACKED_POLICY = os.environ.get("PACKED", "1")


def med():
    buf = ctypes.create_string_buffer(1 << 16)
    C.lib().mi_timing_report_stats(buf, len(buf))
    for line in buf.value.decode().splitlines():
        name, cnt, tot, m, lo, hi = line.rsplit(" ", 5)
        if name == "nl_query_matrix_f32":
            return float(m)


sets = [(torch.empty((n, M), dtype=torch.int32, device=dev), torch.empty((n, M, 3), dtype=torch.int32, device=dev)) for _ in range(K)]
out = []
for nm, sh in sets:
    for _ in range(2):
        cell_list(p32, 40.0, c32, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    torch.cuda.synchronize()
    C.lib().mi_timing_enable(1)
    for _ in range(REPS):
        cell_list(p32, 40.0, c32, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    torch.cuda.synchronize()
    C.lib().mi_timing_enable(0)
    out.append(round(med(), 4))
print("fill ms per buffer set:", out, "num.max", int(num.max()), "num.mean", float(num.float().mean()))

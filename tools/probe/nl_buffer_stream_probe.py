"""Probe (round 5): do the "slow" buffer sets of the 40-Bohr fill also stream slower?  For K torch.empty sets: the fill's median (with its
packed companion) next to the plain streaming fill / read rate (`mi_calibrate_fill` / `_read`) of each of the set's three buffers.
If a slow set streams like a fast one, the state is about the fill's access pattern; if its buffers stream slower, it is the memory.

    python tools/probe/nl_buffer_stream_probe.py [K]"""
import ctypes
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops import _capi as C  # noqa: E402
from nvalchemiops.neighborlist import _engine as E  # noqa: E402
from nvalchemiops.neighborlist import cell_list  # noqa: E402
from tests import systems as S  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
BOHR = 1.8897261246
n, M = 100000, 2560
pos, cell, q, numbers = S.fcc_box(n, seed=1234, dtype=np.float64)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
p32, c32 = t((pos * BOHR).astype(np.float32)), t((cell * BOHR).astype(np.float32))
pbc = torch.tensor([True] * 3, device=dev)
num = torch.empty(n, dtype=torch.int32, device=dev)
E._PACKED_POLICY = "1"
L = C.lib()
sink = torch.zeros(1, dtype=torch.float32, device=dev)


def report():
    buf = ctypes.create_string_buffer(1 << 16)
    L.mi_timing_report_stats(buf, len(buf))
    for line in buf.value.decode().splitlines():
        name, cnt, tot, med, lo, hi = line.rsplit(" ", 5)
        if name == "nl_query_matrix_f32":
            return float(med)
    return None


def ev_median(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def stream_rates(buf):
    nb = buf.numel() * buf.element_size() // 16 * 16
    st = C.stream_of(buf)
    tf = ev_median(lambda: L.mi_calibrate_fill(C.ptr(buf), ctypes.c_size_t(nb), ctypes.c_float(0.0), st))
    tr = ev_median(lambda: L.mi_calibrate_read(C.ptr(buf), ctypes.c_size_t(nb), C.ptr(sink), st))
    return nb / tf / 1e9, nb / tr / 1e9  # TB/s


sets = []
for k in range(K):
    sets.append((torch.empty((n, M), dtype=torch.int32, device=dev), torch.empty((n, M, 3), dtype=torch.int32, device=dev)))
for k, (nm, sh) in enumerate(sets):
    for _ in range(2):
        cell_list(p32, 40.0, c32, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    torch.cuda.synchronize()
    L.mi_timing_enable(1)
    for _ in range(7):
        cell_list(p32, 40.0, c32, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    torch.cuda.synchronize()
    L.mi_timing_enable(0)
    ms = report()
    pk = getattr(nm, E._PACKED_ATTR).words
    r = [stream_rates(b) for b in (nm, sh, pk)]
    print(f"set {k}: fill {ms:.4f} ms | streaming fill / read TB/s: matrix {r[0][0]:.2f} / {r[0][1]:.2f}  shifts {r[1][0]:.2f} / {r[1][1]:.2f}  companion {r[2][0]:.2f} / {r[2][1]:.2f}", flush=True)

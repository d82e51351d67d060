"""Probe (round 5): is the two-state timing of the 40-Bohr fill address translation?  Hypothesis: a buffer set is "slow" when the driver
backed it with small physical fragments (the 20 000+ concurrent row streams of the fill then miss the TLBs on nearly every store run) and
"fast" when it got 2 MiB-contiguous fragments; a streaming kernel does not care either way, which is why the calibration never moved.

Two experiments in one process:
  1. K buffer sets from torch.empty (as `nl_buffer_shop.py`) and K sets allocated with hipExtMallocWithFlags(hipDeviceMallocContiguous):
     matrix, shifts AND the packed companion.  The fill is timed on each (library HIP-event median).
  2. run the same script under `rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --kernel-trace` and read the
     per-dispatch counters of nl_query_tiled_kernel<float> with `--dump <db>`: sets are searched in a fixed order, REPS times each.

    python tools/probe/nl_tlb_probe.py [K]            # timings
    python tools/probe/nl_tlb_probe.py --dump file.db # per-dispatch counters of a PMC run of the line above"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]

if len(sys.argv) > 2 and sys.argv[1] == "--dump":
    import sqlite3

    db = sqlite3.connect(sys.argv[2])
    cols = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
    disp = next((c for c in ("dispatch_id", "event_id", "stack_id") if c in cols), None)
    rows = db.execute(f"select {disp}, counter_name, sum(counter_value), min(start), max(end) from pmc_events where name like '%nl_query_tiled_kernel<float%' "
                      f"group by {disp}, counter_name order by min(start)").fetchall()
    per = {}
    for d, c, v, st, en in rows:
        per.setdefault(d, {"start": st, "ms": (en - st) / 1e6})[c] = v
    for k, (d, r) in enumerate(sorted(per.items(), key=lambda kv: kv[1]["start"])):
        if r["ms"] < 0.3:
            continue  # the instantiation that exits at once
        print(k, f"{r['ms']:.3f} ms", {c: int(v) for c, v in r.items() if c not in ("start", "ms")})
    sys.exit(0)

from nvalchemiops import _capi as C  # noqa: E402
from nvalchemiops.neighborlist import _engine as E  # noqa: E402
from nvalchemiops.neighborlist import cell_list  # noqa: E402
from tests import systems as S  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
REPS = int(os.environ.get("REPS", "7"))
dev = torch.device("cuda:0")
BOHR = 1.8897261246
n, M = 100000, 2560
pos, cell, q, numbers = S.fcc_box(n, seed=1234, dtype=np.float64)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
p32, c32 = t((pos * BOHR).astype(np.float32)), t((cell * BOHR).astype(np.float32))
pbc = torch.tensor([True] * 3, device=dev)
num = torch.empty(n, dtype=torch.int32, device=dev)
E._PACKED_POLICY = "1"


def hip():
    path = "libamdhip64.so"
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
    return ctypes.CDLL(path)


HIP = hip()


class Raw:
    """Device memory from hipExtMallocWithFlags, exposed through __cuda_array_interface__ (torch.as_tensor wraps it without a copy)."""

    def __init__(self, nbytes, flags):
        self.ptr = ctypes.c_void_p()
        rc = HIP.hipExtMallocWithFlags(ctypes.byref(self.ptr), ctypes.c_size_t(nbytes), ctypes.c_uint(flags))
        if rc != 0:
            raise RuntimeError(f"hipExtMallocWithFlags({nbytes}, {flags}) = {rc}")
        self.nbytes = nbytes
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.ptr.value, False), "version": 2}

    def __del__(self):
        if getattr(self, "ptr", None) and self.ptr.value:
            HIP.hipFree(self.ptr)


def make_set(mode):
    if mode == "torch":
        return torch.empty((n, M), dtype=torch.int32, device=dev), torch.empty((n, M, 3), dtype=torch.int32, device=dev), None, []
    raws = [Raw(n * M * 4, 0x4), Raw(n * M * 12, 0x4), Raw(int(C.lib().mi_nl_packed_bytes(n, M)), 0x4)]
    nm = torch.as_tensor(raws[0], device=dev).view(torch.int32).view(n, M)
    sh = torch.as_tensor(raws[1], device=dev).view(torch.int32).view(n, M, 3)
    words = torch.as_tensor(raws[2], device=dev)
    return nm, sh, words, raws


def report():
    buf = ctypes.create_string_buffer(1 << 16)
    C.lib().mi_timing_report_stats(buf, len(buf))
    for line in buf.value.decode().splitlines():
        name, cnt, tot, med, lo, hi = line.rsplit(" ", 5)
        if name == "nl_query_matrix_f32":
            return float(med)
    return None


def timed(nm, sh, words):
    if words is not None:  # seed the companion's storage: the search reuses a companion of the right size it finds on the matrix
        E._written(nm, sh)
        setattr(nm, E._PACKED_ATTR, E.PackedCompanion(words, nm, sh, n))
    for _ in range(2):
        cell_list(p32, 40.0, c32, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    torch.cuda.synchronize()
    C.lib().mi_timing_enable(1)
    for _ in range(REPS):
        cell_list(p32, 40.0, c32, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    torch.cuda.synchronize()
    C.lib().mi_timing_enable(0)
    return report()


sets = []
for mode in ("torch", "contig"):
    for k in range(K):
        try:
            sets.append((mode, k) + make_set(mode))
        except Exception as exc:
            print(mode, k, "allocation failed:", exc, flush=True)
for rnd in range(2):
    for mode, k, nm, sh, words, raws in sets:
        ms = timed(nm, sh, words)
        pk = getattr(nm, E._PACKED_ATTR).words
        print(f"round {rnd} {mode:6s} set {k}  matrix 0x{nm.data_ptr():x} shifts 0x{sh.data_ptr():x} companion 0x{pk.data_ptr():x}  fill {ms:.4f} ms", flush=True)

"""Probe (round 5): the 40-Bohr fill into buffers whose virtual -> physical map is constructed on purpose (tools/probe/probe_vmm.hip):
one virtual range per array, backed by physical chunks created in sequence and mapped in order / reversed / randomly permuted / strided,
at several chunk sizes; next to plain torch.empty sets and hipDeviceMallocContiguous sets.  Library HIP-event medians of the fill with
its packed companion (all three arrays get the same treatment).

    python tools/probe/nl_vmm_probe.py"""
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

dev = torch.device("cuda:0")
BOHR = 1.8897261246
n, M = 100000, 2560
pos, cell, q, numbers = S.fcc_box(n, seed=1234, dtype=np.float64)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
p32, c32 = t((pos * BOHR).astype(np.float32)), t((cell * BOHR).astype(np.float32))
pbc = torch.tensor([True] * 3, device=dev)
num = torch.empty(n, dtype=torch.int32, device=dev)
E._PACKED_POLICY = "1"
V = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe_vmm.so"))
V.vmm_alloc.restype = ctypes.c_void_p
V.vmm_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint]
V.vmm_granularity.restype = ctypes.c_size_t
print("VMM granularity: minimum", V.vmm_granularity(0), "recommended", V.vmm_granularity(1), flush=True)


class Ext:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def vmm_tensor(nbytes, chunk, mode, seed):
    p = V.vmm_alloc(nbytes, chunk, mode, seed)
    if not p:
        raise RuntimeError("vmm_alloc failed")
    return torch.as_tensor(Ext(p, nbytes), device=dev)


def report():
    buf = ctypes.create_string_buffer(1 << 16)
    C.lib().mi_timing_report_stats(buf, len(buf))
    for line in buf.value.decode().splitlines():
        name, cnt, tot, med, lo, hi = line.rsplit(" ", 5)
        if name == "nl_query_matrix_f32":
            return float(med)
    return None


def timed(nm, sh, words, reps=7):
    if words is not None:
        E._written(nm, sh)
        setattr(nm, E._PACKED_ATTR, E.PackedCompanion(words, nm, sh, n))
    for _ in range(2):
        cell_list(p32, 40.0, c32, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    torch.cuda.synchronize()
    C.lib().mi_timing_enable(1)
    for _ in range(reps):
        cell_list(p32, 40.0, c32, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    torch.cuda.synchronize()
    C.lib().mi_timing_enable(0)
    return report()


pkb = int(C.lib().mi_nl_packed_bytes(n, M))
keep = []
for k in range(3):
    nm, sh = torch.empty((n, M), dtype=torch.int32, device=dev), torch.empty((n, M, 3), dtype=torch.int32, device=dev)
    keep.append((nm, sh))
    print(f"torch.empty set {k}: fill {timed(nm, sh, None):.4f} ms", flush=True)
MB = 1 << 20
for chunk in (2 * MB, 8 * MB, 32 * MB, 256 * MB):
    for mode, name in ((0, "in order"), (2, "random"), (3, "strided")):
        for rep in range(2 if mode == 2 else 1):
            try:
                nm = vmm_tensor(n * M * 4, chunk, mode, 11 + rep).view(torch.int32).view(n, M)
                sh = vmm_tensor(n * M * 12, chunk, mode, 23 + rep).view(torch.int32).view(n, M, 3)
                w = vmm_tensor(pkb, chunk, mode, 37 + rep)
                keep.append((nm, sh, w))
                print(f"VMM chunk {chunk // MB:4d} MiB, chunks mapped {name:8s}: fill {timed(nm, sh, w):.4f} ms", flush=True)
            except Exception as exc:
                print(f"VMM chunk {chunk // MB} MiB {name}: failed: {exc}", flush=True)

"""Probe (round 5): streaming fill / read rate of a 3 GiB buffer built with HIP virtual memory management from chunks of a given size mapped in
order or in a random permutation (tools/probe/probe_vmm.hip), next to torch.empty and hipDeviceMallocContiguous buffers of the same size.
Does scattering the physical backing at SOME granularity reproduce the fast write class (6.7 - 6.9 TB/s instead of 5.9)?

    python tools/probe/vmm_stream_probe.py"""
import ctypes
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops import _capi as C  # noqa: E402

dev = torch.device("cuda:0")
L = C.lib()
sink = torch.zeros(1, dtype=torch.float32, device=dev)
V = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe_vmm.so"))
V.vmm_alloc.restype = ctypes.c_void_p
V.vmm_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint]
NB = 3 * (1 << 30)


def hiprt():
    path = "libamdhip64.so"
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
    return ctypes.CDLL(path)


HIP = hiprt()


class Ext:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def ev_median(fn, reps=5):
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


def rates(buf):
    st = C.stream_of(buf)
    tf = ev_median(lambda: L.mi_calibrate_fill(C.ptr(buf), ctypes.c_size_t(NB), ctypes.c_float(0.0), st))
    tr = ev_median(lambda: L.mi_calibrate_read(C.ptr(buf), ctypes.c_size_t(NB), C.ptr(sink), st))
    return NB / tf / 1e9, NB / tr / 1e9


keep = []
for k in range(3):
    b = torch.empty(NB, dtype=torch.uint8, device=dev)
    keep.append(b)
    print("torch.empty        : fill %.2f read %.2f TB/s" % rates(b), flush=True)
p = ctypes.c_void_p()
if HIP.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(NB), ctypes.c_uint(0x4)) == 0:
    b = torch.as_tensor(Ext(p.value, NB), device=dev)
    keep.append(b)
    print("contiguous flag    : fill %.2f read %.2f TB/s" % rates(b), flush=True)
KB, MB = 1 << 10, 1 << 20
for chunk in (64 * KB, 256 * KB, 1 * MB, 2 * MB, 16 * MB):
    for mode, name in ((0, "in order"), (2, "random")):
        t0 = time.perf_counter()
        ptr = V.vmm_alloc(NB, chunk, mode, 5)
        dt = time.perf_counter() - t0
        if not ptr:
            print(f"VMM {chunk // KB} KiB {name}: failed", flush=True)
            continue
        b = torch.as_tensor(Ext(ptr, NB), device=dev)
        keep.append(b)
        f, r = rates(b)
        print(f"VMM chunk {chunk // KB:6d} KiB {name:8s}: fill {f:.2f} read {r:.2f} TB/s   (built in {dt:.2f} s)", flush=True)

"""Probe (round 5): plain streaming fill / read rate of successive 2 GiB allocations covering most of the HBM, in allocation order.
Question: the write rate of a buffer is one of two classes (~5.9 vs ~6.85 TB/s, nl_buffer_stream_probe.py) and the classes come in runs
of consecutive allocations -- is there a regular physical structure (period, fraction) behind it?

    python tools/probe/hbm_region_probe.py [GiB per buffer] [count]"""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops import _capi as C  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda:0")
L = C.lib()
sink = torch.zeros(1, dtype=torch.float32, device=dev)
nb = int(gib * (1 << 30)) // 16 * 16


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


bufs, line = [], []
free0 = torch.cuda.mem_get_info(dev)[0]
print(f"free {free0 / 2**30:.1f} GiB; {gib} GiB per buffer", flush=True)
for k in range(count):
    if torch.cuda.mem_get_info(dev)[0] < nb + (2 << 30):
        break
    b = torch.empty(nb, dtype=torch.uint8, device=dev)
    bufs.append(b)
    st = C.stream_of(b)
    tf = ev_median(lambda: L.mi_calibrate_fill(C.ptr(b), ctypes.c_size_t(nb), ctypes.c_float(0.0), st))
    tr = ev_median(lambda: L.mi_calibrate_read(C.ptr(b), ctypes.c_size_t(nb), C.ptr(sink), st))
    f, r = nb / tf / 1e9, nb / tr / 1e9
    line.append(f)
    print(f"{k:3d} va 0x{b.data_ptr():x}  fill {f:.2f}  read {r:.2f} TB/s  {'#' * int((f - 5.0) * 20)}", flush=True)
print("classes:", "".join("F" if f > 6.45 else ("s" if f < 6.2 else "m") for f in line))

"""Streaming writes in address order vs scattered over footprints of 0.25 .. 4 GiB (probe_scatter.hip): is the write rate a function of the
live footprint?   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe/libprobe_scatter.so tools/probe/probe_scatter.hip"""
import ctypes, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools/probe/libprobe_scatter.so"))
lib.probe_scatter.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
buf = torch.empty(4 << 30, dtype=torch.uint8, device="cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
TOTAL = 4 << 30
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for blocks in (1536, 8192):
    for chunk in (1024, 4096, 32768):
        row = []
        for fp_gib in (0.25, 1, 2, 4):
            fp = int(fp_gib * (1 << 30))
            for mult, name in ((1, "lin"), (0x9E3779B1, "scat")):
                ms = t(lambda: lib.probe_scatter(buf.data_ptr(), fp, chunk, mult, TOTAL, blocks, st))
                row.append(f"{fp_gib:g}G {name} {TOTAL / ms / 1e6:.0f}")
        print(f"blocks {blocks} chunk {chunk:6d} B  GB/s: " + "  ".join(row), flush=True)

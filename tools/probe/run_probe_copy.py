"""What a plain copy / fill / read stream reaches on this box, by access variant and grid size (GB/s; copy counts read + written bytes).
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe/libprobe_copy.so tools/probe/probe_copy.hip"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools/probe/libprobe_copy.so"))
dev = "cuda:0"
nbytes = 4 << 30
s = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
out = torch.zeros(4, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {0: "U1", 1: "U4", 2: "U1 nt", 3: "U4 nt", 4: "U8"}
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for what, label, mult in ((0, "copy", 2.0), (1, "fill", 1.0), (2, "read", 1.0)):
    for blocks in (2048, 4096, 16384, 65536):
        row = []
        for v in range(5):
            ms = t(lambda: lib.probe_copy(what, v, blocks, ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(d.data_ptr()), ctypes.c_longlong(nbytes),
                                          ctypes.c_void_p(out.data_ptr()), st))
            row.append(f"{names[v]} {mult * nbytes / ms / 1e6:6.0f}")
        print(f"{label} blocks {blocks:6d}: " + "  ".join(row), flush=True)
ms = t(lambda: d.copy_(s)); print(f"torch copy_ (hipMemcpy D2D): {2 * nbytes / ms / 1e6:.0f} GB/s")
ms = t(lambda: d.zero_()); print(f"torch zero_ (fill kernel): {nbytes / ms / 1e6:.0f} GB/s")

"""Write side of the 40-Bohr matrix fill with emulated compute: per-hit stores vs LDS-staged 256-slot chunks (probe_write2.hip)."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools/probe/libprobe_write2.so"))
dev = "cuda:0"; n, m = 100000, 2560
nm = torch.empty((n, m), dtype=torch.int32, device=dev); sh = torch.empty((n, m, 3), dtype=torch.int32, device=dev)
work = torch.zeros(4, dtype=torch.int32, device=dev); sink = torch.zeros(4, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def t(fn, reps=7):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for blocks in (1280,):
    for delay in (0, 4, 8, 12, 16, 24):
        row = []
        for mode, name in ((3, "no stores"), (0, "per-hit"), (1, "staged")):
            ms = t(lambda: lib.probe_write2(mode, P(nm), P(sh), n, m, delay, blocks, P(work), P(sink), st, -1))
            if mode == 3:
                row.append(f"{name} {ms:.3f}")
                continue
            ms_l2 = t(lambda: lib.probe_write2(mode, P(nm), P(sh), n, m, delay, blocks, P(work), P(sink), st, 63))
            row.append(f"{name} {ms:.3f} (rows in L2: {ms_l2:.3f}, drain cost {ms - ms_l2:+.3f})")
        print(f"blocks {blocks} delay {delay:3d} FMAs/group: " + "   ".join(row), flush=True)

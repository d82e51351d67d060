import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools/probe/libprobe.so"))
dev = "cuda:0"
n, per = 100000, 2368
P = n * per
nptr = (torch.arange(n + 1, device=dev, dtype=torch.int64) * per).to(torch.int32)
sh = torch.zeros((P, 3), dtype=torch.int32, device=dev)
apos = torch.rand((n, 4), device=dev)
out = torch.zeros(n, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
e = torch.arange(P, device=dev, dtype=torch.int64)
row = e // per
def runs(R, share=1):   # runs of R consecutive atoms at pseudo-random places near the row's atom (within +-2500 atoms)
    blk = (e % per) // R
    r2 = (row // share) * share
    h = (blk * 2654435761 + r2 * 40503) % 5000
    return ((r2 + h - 2500 + (e % R)) % n).to(torch.int32)
def window_shuffle():  # 64 consecutive atoms per wave-iteration, shuffled across lanes
    it = (e % per) // 64
    lane = e % 64
    perm = torch.randperm(64, device=dev)
    return ((row + it * 64 + perm[lane]) % n).to(torch.int32)
def quad_scatter():  # 4 consecutive lanes contiguous (64 B), quads scattered
    return runs(4)
pats = {"runs1": runs(1), "runs1_shared4": runs(1, 4), "runs2": runs(2), "runs2_shared4": runs(2, 4), "runs4_shared4": runs(4, 4)}
for name, idx in pats.items():
    for v in (33, 34):
        def run():
            rc = lib.probe_walk(v, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sh.data_ptr()), ctypes.c_void_p(nptr.data_ptr()),
                                ctypes.c_void_p(apos.data_ptr()), n, ctypes.c_longlong(P), ctypes.c_void_p(out.data_ptr()), st)
            assert rc == 0
        for _ in range(2): run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run()
        b.record(); torch.cuda.synchronize()
        print(f"{name:20s} gather {'free-running' if v == 33 else 'lockstep'}: {a.elapsed_time(b) / 5:.3f} ms", flush=True)

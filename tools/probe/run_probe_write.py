import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools/probe/libprobe_write.so"))
dev = "cuda:0"; n, m = 100000, 2560
nm = torch.empty((n, m), dtype=torch.int32, device=dev); sh = torch.empty((n, m, 3), dtype=torch.int32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(2):
    for ch in (22, 64):
        f = lambda: lib.probe_write(ch, ctypes.c_void_p(nm.data_ptr()), ctypes.c_void_p(sh.data_ptr()), n, m, st)
        for _ in range(2): f()
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): f()
        b.record(); torch.cuda.synchronize(); t = a.elapsed_time(b) / 5
        print(f"chunk {ch}: {t:.3f} ms  {16.0 * n * m / t / 1e6:.0f} GB/s", flush=True)

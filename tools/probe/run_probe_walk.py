import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from tests import systems as S
from nvalchemiops.neighborlist import cell_list
lib = ctypes.CDLL(os.path.join(ROOT, "tools/probe/libprobe.so"))
dev = "cuda:0"
n = 100000
pos, cell, _, z = S.fcc_box(n, dtype=np.float32)
B = 1.8897261
tp, tc = torch.as_tensor(pos * B, device=dev), torch.as_tensor(cell * B, device=dev)
lst, nptr, sh = cell_list(tp, 40.0, tc, torch.tensor([True] * 3, device=dev), return_neighbor_list=True)
idx = lst[1].contiguous()
P = idx.shape[0]
apos = torch.cat([tp, torch.full((n, 1), 1.4, device=dev)], 1).contiguous()
out = torch.zeros(n, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
print("pairs", P, "bytes/pass %.2f GB" % (16 * P / 1e9))
row = idx[nptr[50000]:nptr[50001]].cpu().numpy()
d = np.diff(row)
print("row len", len(row), "frac consecutive (+1):", float((d == 1).mean()), "mean run:", len(row) / max(1, int((d != 1).sum()) + 1))
for v in [3, 40, 41, 42]:
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
    t = a.elapsed_time(b) / 5
    byts = (4 if v == 0 else 16) * P
    print(f"variant {v}: {t:.3f} ms  {byts / t / 1e6:.0f} GB/s", flush=True)

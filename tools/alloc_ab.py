"""Does WHERE the caller's output buffers sit change the 40-Bohr matrix fill?  One process, one box: the same fill into differently
allocated / aligned buffers, interleaved.   python tools/alloc_ab.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from tests import systems as S
from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import _engine as E
dev = torch.device("cuda", 0)
n, M = 100000, 2560
pos, cell, _, _ = S.fcc_box(n, seed=1234, dtype=np.float64)
B = 1.8897261246
tp = torch.as_tensor((pos * B).astype(np.float32), device=dev)
tc = torch.as_tensor((cell * B).astype(np.float32), device=dev).reshape(1, 3, 3)
pbc = torch.ones((1, 3), dtype=torch.bool, device=dev)
num = torch.empty(n, dtype=torch.int32, device=dev)
ws = E.workspace(n, 1, tp.dtype, dev)
def carve(buf, off, shape):
    cnt = int(np.prod(shape))
    return buf[off:off + 4 * cnt].view(torch.int32).view(*shape)
sets = {}
keep = []
# recycled: a 4 GiB block allocated, touched and freed first; the two tensors are then carved out of it by torch's allocator
big0 = torch.empty(4 << 30, dtype=torch.uint8, device=dev); big0.zero_(); torch.cuda.synchronize(); del big0
sets["recycled 4 GiB block"] = (torch.empty((n, M), dtype=torch.int32, device=dev), torch.empty((n, M, 3), dtype=torch.int32, device=dev))
hold = torch.empty(256 << 20, dtype=torch.uint8, device=dev)  # keep the rest of that block busy
torch.cuda.empty_cache()
b1 = torch.empty(4 * n * M + (8 << 20), dtype=torch.uint8, device=dev); b2 = torch.empty(12 * n * M + (8 << 20), dtype=torch.uint8, device=dev)
sets["own allocations, 2 MiB aligned"] = (carve(b1, 0, (n, M)), carve(b2, 0, (n, M, 3)))
sets["own allocations, + 0x725000"] = (carve(b1, 0x725000, (n, M)), carve(b2, 0x7b5000, (n, M, 3)))
sets["own allocations, + 128 B"] = (carve(b1, 128, (n, M)), carve(b2, 128, (n, M, 3)))
big = torch.empty(16 * n * M + (16 << 20), dtype=torch.uint8, device=dev)
sets["one 4.1 GB allocation"] = (carve(big, 0, (n, M)), carve(big, ((4 * n * M + (2 << 20) - 1) >> 21) << 21, (n, M, 3)))
def t(nm, nsh, reps=7):
    fn = lambda: E.run(tp, tc, pbc, None, 40.0, C.NL_MATRIX, 0, nm=nm, nsh=nsh, num=num, max_neighbors=M, fill_value=n, ws=ws)
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for name, (nm, nsh) in sets.items():
    print(f"{name:34s} nm 0x{nm.data_ptr():x} nsh 0x{nsh.data_ptr():x}")
for rnd in range(3):
    print("round", rnd, "  ".join(f"{name.split(',')[-1].strip()[:22]}: {t(nm, nsh):.3f}" for name, (nm, nsh) in sets.items()), flush=True)

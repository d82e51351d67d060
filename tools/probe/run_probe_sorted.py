"""Probe: cost of the cn-like walk when the atoms are numbered in the cell-sorted order of the search grid (what a sorted
renumbering inside D3 would give) vs the lattice order of the benchmark generator vs a random order."""
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
pos, cell = pos * B, cell * B
def order(kind):
    if kind == "lattice":
        return np.arange(n)
    if kind == "random":
        return np.random.default_rng(0).permutation(n)
    w = cell[0, 0] / 17  # the search grid of the 40-Bohr list: 17 cells per axis, x fastest
    c = np.floor(pos / w).astype(np.int64).clip(0, 16)
    return np.lexsort((np.arange(n), c[:, 0], c[:, 1], c[:, 2]))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for kind in ("lattice", "cell-sorted", "random"):
    o = order(kind)
    tp, tc = torch.as_tensor(pos[o], device=dev), torch.as_tensor(cell, device=dev)
    lst, nptr, sh = cell_list(tp, 40.0, tc, torch.tensor([True] * 3, device=dev), return_neighbor_list=True)
    idx = lst[1].contiguous(); P = idx.shape[0]
    apos = torch.cat([tp, torch.full((n, 1), 1.4, device=dev)], 1).contiguous(); out = torch.zeros(n, device=dev)
    res = []
    for v in (1, 31, 3, 41):
        def run():
            assert lib.probe_walk(v, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sh.data_ptr()), ctypes.c_void_p(nptr.data_ptr()),
                                  ctypes.c_void_p(apos.data_ptr()), n, ctypes.c_longlong(P), ctypes.c_void_p(out.data_ptr()), st) == 0
        for _ in range(2): run()
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run()
        b.record(); torch.cuda.synchronize(); res.append(a.elapsed_time(b) / 5)
    print(f"{kind:12s}: stream only {res[0]:.3f}  +4B gather {res[1]:.3f}  +16B gather (cn walk) {res[2]:.3f}  cn walk lock-step x8 {res[3]:.3f} ms", flush=True)
    del lst, sh, idx

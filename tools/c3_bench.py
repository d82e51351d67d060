"""BASELINE config 3: 256 x 512-atom molecules (non-periodic), batched neighbour list (40 Bohr) + DFT-D3(BJ), fp32.  Timing aid."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from oracle import oracle as O
from tests import systems as S
from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
from nvalchemiops.neighborlist import neighbor_list
dev = "cuda:0"
B, n = 256, 512
pos, zs = [], []
for s in range(B):
    p, z = S.molecule(n, seed=2000 + s)[:2]
    pos.append(p * 1.8897261); zs.append(z)
pos = torch.as_tensor(np.concatenate(pos), device=dev); z = torch.as_tensor(np.concatenate(zs), device=dev)
bi = torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(n)
t = O.d3_test_tables(17)
prm = D3Parameters(**{k: torch.as_tensor(v, device=dev) for k, v in t.items()})
def timeit(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it
for fmt in ("matrix", "csr"):
    if fmt == "matrix":
        nl = lambda: neighbor_list(pos, 40.0, batch_idx=bi, max_neighbors=512)
        nm, num, sh = nl()
        d3 = lambda: dftd3(pos, z, 0.4289, 4.4407, 0.7875, d3_params=prm, neighbor_matrix=nm, batch_idx=bi, num_systems=B, fill_value=B * n)
        pairs = int(num.sum())
    else:
        nl = lambda: neighbor_list(pos, 40.0, batch_idx=bi, return_neighbor_list=True)
        lst, ptr, lsh = nl()
        d3 = lambda: dftd3(pos, z, 0.4289, 4.4407, 0.7875, d3_params=prm, neighbor_list=lst, neighbor_ptr=ptr, batch_idx=bi, num_systems=B)
        pairs = lst.shape[1]
    tn, td = timeit(nl), timeit(d3)
    print(f"C3 {fmt}: pairs {pairs}  nlist {tn:.3f} ms  d3 {td:.3f} ms  -> {B * n / (tn + td) * 1e3 / 1e6:.1f} M atom-steps/s, {pairs / td / 1e6:.1f} G pairs/s", flush=True)
import ctypes
from nvalchemiops import _capi as C
C.lib().mi_timing_enable(1)
for _ in range(5): neighbor_list(pos, 40.0, batch_idx=bi, max_neighbors=512)
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 14); C.lib().mi_timing_report(buf, len(buf)); print(buf.value.decode())
import time
t0 = time.perf_counter()
for _ in range(5): neighbor_list(pos, 40.0, batch_idx=bi, max_neighbors=512)
torch.cuda.synchronize(); print("wall per call ms", (time.perf_counter() - t0) / 5 * 1e3)

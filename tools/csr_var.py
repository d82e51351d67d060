"""Tuning aid: per-call time of the 40-Bohr CSR list (count / fill kernels via the library timers) over repeated calls."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from tests import systems as S
from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import cell_list
dev = "cuda:0"
pos, cell, _, _ = S.fcc_box(100000, dtype=np.float32)
B = 1.8897261
tp, tc = torch.as_tensor(pos * B, device=dev), torch.as_tensor(cell * B, device=dev)
pbc = torch.tensor([True] * 3, device=dev)
keep = []
for it in range(24):
    C.lib().mi_timing_enable(1)
    lst, nptr, sh = cell_list(tp, 40.0, tc, pbc, return_neighbor_list=True)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14); C.lib().mi_timing_report(buf, len(buf)); C.lib().mi_timing_enable(0)
    d = {l.rsplit(" ", 2)[0]: float(l.rsplit(" ", 2)[2]) for l in buf.value.decode().splitlines()}
    print(it, f"count {d.get('nl_query_count', 0):.3f} csr {d.get('nl_query_csr', 0):.3f}  ptr lst={lst.data_ptr():#x} sh={sh.data_ptr():#x}", flush=True)
    if it % 8 == 7:
        keep.append((lst, sh))  # hold on to the buffers: the next calls get different addresses

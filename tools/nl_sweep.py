"""Tuning aid: cell_list time vs N for the FCC generator (matrix M=64, rc=5, fp32) with the library's per-kernel timers."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from tests import systems as S
from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import cell_list
dev = "cuda:0"
pbc = torch.tensor([True] * 3, device=dev)
for n in [50000, 90000, 100000, 108000, 110000, 131072]:
    pos, cell, _, _ = S.fcc_box(n, dtype=np.float32)
    tp, tc = torch.as_tensor(pos, device=dev), torch.as_tensor(cell, device=dev)
    nm = torch.empty((n, 64), dtype=torch.int32, device=dev); sh = torch.empty((n, 64, 3), dtype=torch.int32, device=dev); num = torch.empty(n, dtype=torch.int32, device=dev)
    f = lambda: cell_list(tp, 5.0, tc, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    for _ in range(3): f()
    torch.cuda.synchronize()
    C.lib().mi_timing_enable(1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14); C.lib().mi_timing_report(buf, len(buf)); C.lib().mi_timing_enable(0)
    print(n, "box", cell[0, 0], f"{a.elapsed_time(b) / 10:.3f} ms/call |", buf.value.decode().replace("\n", " ; "), flush=True)

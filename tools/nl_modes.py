"""Diagnostic: the 40-Bohr headline list in count mode / matrix without shifts / full matrix, plus the box calibration -- which part of
`nl_query_matrix_f32` moves when a box is slow?   python tools/nl_modes.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from tests import systems as S
from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import _engine as E
import bench
dev = torch.device("cuda", 0)
cal = bench.hbm_calibration(dev)
print("calibration", {k: round(v) for k, v in cal.items() if k.endswith("GBps")})
n, M = 100000, 2560
pos, cell, _, _ = S.fcc_box(n, seed=1234, dtype=np.float64)
B = 1.8897261246
tp = torch.as_tensor((pos * B).astype(np.float32), device=dev)
tc = torch.as_tensor((cell * B).astype(np.float32), device=dev).reshape(1, 3, 3)
pbc = torch.ones((1, 3), dtype=torch.bool, device=dev)
nm = torch.empty((n, M), dtype=torch.int32, device=dev)
nsh = torch.empty((n, M, 3), dtype=torch.int32, device=dev)
num = torch.empty(n, dtype=torch.int32, device=dev)
ws = E.workspace(n, 1, tp.dtype, dev)
def t(fn, reps=9):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)), min(ts), max(ts)
full = lambda: E.run(tp, tc, pbc, None, 40.0, C.NL_MATRIX, 0, nm=nm, nsh=nsh, num=num, max_neighbors=M, fill_value=n, ws=ws)
noshift = lambda: E.run(tp, tc, pbc, None, 40.0, C.NL_MATRIX, C.NL_NO_SHIFTS, nm=nm, nsh=None, num=num, max_neighbors=M, fill_value=n, ws=ws)
count = lambda: E.run(tp, tc, pbc, None, 40.0, C.NL_COUNT, 0, num=num, ws=ws)
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, fn in (("count (no stores)", count), ("matrix, indices only (1.0 GB)", noshift), ("matrix + shifts (4.1 GB)", full)):
    if only and not name.startswith(only):
        continue
    print(f"{name:32s} median %.3f  min %.3f  max %.3f ms  (incl. ~0.05 ms binning)" % t(fn), flush=True)

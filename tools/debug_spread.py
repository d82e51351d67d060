"""Debug aid: tile-owned vs atomic spread on the same inputs (orders 1-6)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops import _capi as C
dev = "cuda:0"
g = np.random.default_rng(0)
cell = np.array([[12.0, 0, 0], [2.4, 10.8, 0], [1.2, -1.8, 13.2]])
n = 260
pos = torch.as_tensor(g.uniform(-0.5, 1.5, (n, 3)) @ cell, device=dev)
q = torch.as_tensor(g.normal(size=n), device=dev)
cit = torch.linalg.inv(torch.as_tensor(cell, device=dev)).T.contiguous().reshape(1, 3, 3)
for order in range(1, 7):
    for dims in ((8, 8, 8), (16, 8, 24)):
        out = []
        for tiled in (False, True):
            mesh = torch.zeros((1,) + dims, dtype=torch.float64, device=dev)
            wsb = int(C.lib().mi_spline_spread_workspace_bytes(n, 1, *dims)) if tiled else 0
            ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
            rc = C.lib().mi_spline_spread(C.ptr(pos), C.ptr(q), None, C.ptr(cit), n, 1, *dims, order, 0, 1, C.ptr(mesh), C.ptr(ws) if tiled else None,
                                          ctypes.c_size_t(wsb), C.stream_of(pos))
            assert rc == 0
            out.append(mesh)
        d = (out[0] - out[1]).abs()
        print(f"order {order} dims {dims}: max |atomic - tiled| = {float(d.max()):.3e}  sums {float(out[0].sum()):.3e} {float(out[1].sum()):.3e}  q {float(q.sum()):.3e}")
print("conservation (tiled path), atoms inside the cell vs shifted by lattice vectors:")
for shift in (0, 1):
    p2 = torch.as_tensor((g.uniform(0, 1, (n, 3)) + shift * g.integers(-1, 2, (n, 3))) @ cell, device=dev)
    for order in (4, 5, 6):
        dims = (16, 16, 16)
        mesh = torch.zeros((1,) + dims, dtype=torch.float64, device=dev)
        wsb = int(C.lib().mi_spline_spread_workspace_bytes(n, 1, *dims)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        C.lib().mi_spline_spread(C.ptr(p2), C.ptr(q), None, C.ptr(cit), n, 1, *dims, order, 0, 1, C.ptr(mesh), C.ptr(ws), ctypes.c_size_t(wsb), C.stream_of(p2))
        ones = torch.ones(n, dtype=torch.float64, device=dev)
        m1 = torch.zeros((1,) + dims, dtype=torch.float64, device=dev)
        C.lib().mi_spline_spread(C.ptr(p2), C.ptr(ones), None, C.ptr(cit), n, 1, *dims, order, 0, 1, C.ptr(m1), C.ptr(ws), ctypes.c_size_t(wsb), C.stream_of(p2))
        print(f"  shifted={shift} order {order}: sum(mesh) - sum(q) = {float(mesh.sum() - q.sum()):.3e}   sum of weights - N = {float(m1.sum()) - n:.3e}")

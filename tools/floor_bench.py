"""Launch-floor check: wall time per call of the public ops on small systems (Python + ctypes + kernels)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from oracle import oracle as O
from tests import systems as S
from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
from nvalchemiops.neighborlist import cell_list, naive_neighbor_list
dev = "cuda:0"
pbc = torch.tensor([True] * 3, device=dev)
t = O.d3_test_tables(17)
prm = D3Parameters(**{k: torch.as_tensor(v, device=dev) for k, v in t.items()})
def wall(f, it=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
for n in (64, 1024, 4096, 8192, 16384, 32768):
    pos, cell, q, z = S.fcc_box(n, dtype=np.float32)
    tp, tc = torch.as_tensor(pos, device=dev), torch.as_tensor(cell, device=dev)
    tq, tz = torch.as_tensor(q, device=dev), torch.as_tensor(z, device=dev)
    nm, num, sh = cell_list(tp, 5.0, tc, pbc, max_neighbors=64)
    a = wall(lambda: cell_list(tp, 5.0, tc, pbc, max_neighbors=64))
    b = wall(lambda: cell_list(tp, 5.0, tc, pbc, return_neighbor_list=True))
    c = wall(lambda: dftd3(tp * 1.89, tz, 0.4289, 4.4407, 0.7875, d3_params=prm, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=tc[None] * 1.89, fill_value=n))
    d = wall(lambda: particle_mesh_ewald(tp, tq, tc, alpha=0.35, mesh_dimensions=(32, 32, 32), spline_order=4, neighbor_matrix=nm,
                                         neighbor_matrix_shifts=sh, compute_forces=True))
    # naive method (reference: O(N^2) tile kernels, 0.267 ms at 1024 atoms / 4.53 ms at 16384 on H100): served by the O(N) pipeline here
    e = wall(lambda: naive_neighbor_list(tp, 5.0, cell=tc.reshape(1, 3, 3), pbc=pbc.reshape(1, 3), max_neighbors=64))
    f = wall(lambda: naive_neighbor_list(tp, 5.0, max_neighbors=64))
    print(f"N={n}: cell_list matrix {a:.3f} ms, CSR {b:.3f} ms, dftd3 {c:.3f} ms, PME(E+F) {d:.3f} ms, naive pbc {e:.3f} ms, naive free {f:.3f} ms", flush=True)

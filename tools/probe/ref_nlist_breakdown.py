"""Where the time of the reference's 32 768-atom cell_list benchmark call goes: wall per call, per timed-kernel-group HIP-event medians, and
the same call without the caller-side cache tensors."""
import os, sys, time, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
import bench
from nvalchemiops import _capi as C
from nvalchemiops.neighborlist import neighbor_list
from nvalchemiops.neighborlist.neighbor_utils import allocate_cell_list, estimate_max_neighbors
from nvalchemiops.neighborlist.cell_list import estimate_cell_list_sizes
dev = torch.device("cuda:0")
for n in (131072, 524288):
    pos, cell, _ = bench._lattice(n, [[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]], 4.0, np.float32)
    tp, tc = torch.as_tensor(pos, device=dev), torch.as_tensor(cell, device=dev).reshape(1, 3, 3)
    pbc = torch.ones((1, 3), dtype=torch.bool, device=dev)
    m = estimate_max_neighbors(5.0, atomic_density=0.35, safety_factor=1.0)
    nm = torch.full((n, m), n, dtype=torch.int32, device=dev); sh = torch.zeros((n, m, 3), dtype=torch.int32, device=dev); num = torch.zeros(n, dtype=torch.int32, device=dev)
    max_cells, radius = estimate_cell_list_sizes(tc, pbc, 5.0)
    cache = allocate_cell_list(n, max_cells, radius, dev)
    kw = dict(zip(("cells_per_dimension", "neighbor_search_radius", "atom_periodic_shifts", "atom_to_cell_mapping", "atoms_per_cell_count", "cell_atom_start_indices", "cell_atom_list"), cache))
    variants = {"cache+prealloc": lambda: neighbor_list(tp, 5.0, cell=tc, pbc=pbc, method="cell_list", neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num, **kw),
                "prealloc only": lambda: neighbor_list(tp, 5.0, cell=tc, pbc=pbc, method="cell_list", neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num),
                "plain": lambda: neighbor_list(tp, 5.0, cell=tc, pbc=pbc, method="cell_list", max_neighbors=m)}
    for name, fn in variants.items():
        for _ in range(10): fn()
        torch.cuda.synchronize()
        C.lib().mi_timing_enable(1)
        t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 50 * 1e3
        C.lib().mi_timing_enable(0)
        rep = bench.kernel_report()
        med, lo, hi = bench._median_ms(fn, 5, 50)
        print(n, name, "wall/call %.3f ms, event median %.3f |" % (wall, med), {k: round(v[2], 4) for k, v in rep.items()}, flush=True)

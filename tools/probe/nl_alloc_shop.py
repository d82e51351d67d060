"""Probe (round 4): is the list fill's two-state timing a property of the ALLOCATION?  One process, caching allocator off,
K pairs of row buffers each from its own hipMalloc (all kept alive), the search timed on every pair (median of 7), twice round."""
import os, statistics, sys
os.environ["PYTORCH_NO_HIP_MEMORY_CACHING"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nvalchemi-toolkit-ops_amd"))
import torch
import bench
from nvalchemiops.neighborlist import cell_list

dev = torch.device("cuda:0")
sysd, _ = bench.build_system(100000, 1234, dev)
n, md = 100000, 2560
num = torch.empty(n, dtype=torch.int32, device=dev)

def timed(dm, ds, reps=7):
    ts = []
    for _ in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        cell_list(sysd["pos32b"], 40.0, sysd["cell32b"], sysd["pbc"], neighbor_matrix=dm, neighbor_matrix_shifts=ds, num_neighbors=num)
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts[2:])

pairs = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    if k % 3 == 2:
        pad = torch.empty((k + 1) * (37 << 20), dtype=torch.uint8, device=dev)  # perturb the driver's free list between pairs
        pairs.append((None, None, pad))
        continue
    pairs.append((torch.empty((n, md), dtype=torch.int32, device=dev), torch.empty((n, md, 3), dtype=torch.int32, device=dev), None))
for rnd in range(2):
    for k, (dm, ds, pad) in enumerate(pairs):
        if dm is None:
            continue
        print("round %d pair %d  matrix 0x%x shifts 0x%x  %.4f ms" % (rnd, k, dm.data_ptr(), ds.data_ptr(), timed(dm, ds)))
# cross pairs: matrix of pair a with shifts of pair b
real = [(dm, ds) for dm, ds, _ in pairs if dm is not None]
for a in range(min(3, len(real))):
    for b in range(min(3, len(real))):
        if a != b:
            print("cross matrix %d + shifts %d: %.4f ms" % (a, b, timed(real[a][0], real[b][1])))

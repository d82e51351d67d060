"""Probe (round 4): does the 40-Bohr list fill's two-state timing (0.82 / 1.13 ms, fixed per process) depend on WHERE the two
row buffers sit?  One process, one 12 GiB arena, the padded matrix (4 B/slot) at `base + shift` and the shift matrix (12 B/slot) at
`base + shift + gap`: sweep `gap` and `shift`, time the search (median of 7) for each placement."""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nvalchemi-toolkit-ops_amd"))
import numpy as np, torch
import bench
from nvalchemiops.neighborlist import cell_list

dev = torch.device("cuda:0")
sysd, _ = bench.build_system(100000, 1234, dev)
n, md = 100000, 2560
nb_j, nb_s = n * md * 4, n * md * 12
arena = torch.empty(12 << 30, dtype=torch.uint8, device=dev)
a0 = (-arena.data_ptr()) % (1 << 30)  # 1 GiB-aligned start inside the arena
num = torch.empty(n, dtype=torch.int32, device=dev)
print("arena at 0x%x, aligned start +0x%x" % (arena.data_ptr(), a0))

def timed(off_j, off_s, reps=7):
    dm = arena[off_j:off_j + nb_j].view(torch.int32).view(n, md)
    ds = arena[off_s:off_s + nb_s].view(torch.int32).view(n, md, 3)
    ts = []
    for _ in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        cell_list(sysd["pos32b"], 40.0, sysd["cell32b"], sysd["pbc"], neighbor_matrix=dm, neighbor_matrix_shifts=ds, num_neighbors=num)
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts[2:])

MiB = 1 << 20
base_gap = (nb_j + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
print("gap sweep (matrix at aligned start, shifts at +%d MiB + gap):" % (base_gap // MiB))
for gap in (0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, MiB, 2 * MiB, 4 * MiB, 16 * MiB, 64 * MiB, 256 * MiB, 1024 * MiB):
    print("  gap %10d  %.4f ms" % (gap, timed(a0, a0 + base_gap + gap)))
print("shift sweep (both buffers moved together, gap 0):")
for sh in (0, 256, 4096, 65536, MiB, 16 * MiB, 256 * MiB, 1024 * MiB, 2048 * MiB):
    print("  shift %10d  %.4f ms" % (sh, timed(a0 + sh, a0 + sh + base_gap)))
print("row-major swap (shifts first, matrix behind):  %.4f ms" % timed(a0 + 4096 * MiB, a0))
# torch's own allocations for comparison
dm = torch.empty((n, md), dtype=torch.int32, device=dev); ds = torch.empty((n, md, 3), dtype=torch.int32, device=dev)
ts = []
for _ in range(9):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); cell_list(sysd["pos32b"], 40.0, sysd["cell32b"], sysd["pbc"], neighbor_matrix=dm, neighbor_matrix_shifts=ds, num_neighbors=num); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b))
print("torch.empty buffers at 0x%x / 0x%x: %.4f ms" % (dm.data_ptr(), ds.data_ptr(), statistics.median(ts[2:])))

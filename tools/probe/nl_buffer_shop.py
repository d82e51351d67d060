"""Probe (round 5): how much does the 40-Bohr fill (with its packed companion) depend on WHICH buffers it writes, inside one process with
torch's caching allocator on?  K buffer sets (matrix, shifts, companion) allocated one after the other and all kept alive, the search and
the D3 CN pass timed on each (library HIP-event medians), two rounds.  Round 4 saw 1.13 - 1.49 ms across six hipMalloc pairs with the
allocator off (profiles/r04_probe_nl_alloc_shop.log); if the spread is there with the allocator on, choosing the buffers by a trial fill
at set-up time removes the "slow state" instead of fencing it with a median over processes.

    python tools/probe/nl_buffer_shop.py [K]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops import _capi as C  # noqa: E402
from nvalchemiops.interactions.dispersion import D3Parameters, dftd3  # noqa: E402
from nvalchemiops.neighborlist import _engine as E  # noqa: E402
from nvalchemiops.neighborlist import cell_list  # noqa: E402
from tests import systems as S  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
BOHR = 1.8897261246
n, M = 100000, 2560
pos, cell, q, numbers = S.fcc_box(n, seed=1234, dtype=np.float64)
tables = S.d3_test_tables(94, seed=7)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
p32, c32, z = t((pos * BOHR).astype(np.float32)), t((cell * BOHR).astype(np.float32)), t(numbers)
params = D3Parameters(rcov=t(tables["rcov"]), r4r2=t(tables["r4r2"]), c6ab=t(tables["c6ab"]), cn_ref=t(tables["cn_ref"]))
pbc = torch.tensor([True] * 3, device=dev)
num = torch.empty(n, dtype=torch.int32, device=dev)
E._PACKED_POLICY = "1"


def report():
    buf = ctypes.create_string_buffer(1 << 16)
    C.lib().mi_timing_report_stats(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot, med, lo, hi = line.rsplit(" ", 5)
        out[name] = float(med)
    return out


def timed(nm, sh, reps=9):
    for _ in range(2):
        cell_list(p32, 40.0, c32, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    torch.cuda.synchronize()
    C.lib().mi_timing_enable(1)
    for _ in range(reps):
        cell_list(p32, 40.0, c32, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
        dftd3(p32, z, a1=0.4289, a2=4.4407, s8=0.7875, d3_params=params, neighbor_matrix=nm, neighbor_matrix_shifts=sh, fill_value=n,
              cell=c32.unsqueeze(0), compute_virial=True, num_systems=1)
    torch.cuda.synchronize()
    C.lib().mi_timing_enable(0)
    r = report()
    return r.get("nl_query_matrix_f32"), r.get("d3_cn"), r.get("d3_energy"), r.get("d3_chain")


sets = []
for k in range(K):
    nm = torch.empty((n, M), dtype=torch.int32, device=dev)
    sh = torch.empty((n, M, 3), dtype=torch.int32, device=dev)
    sets.append((nm, sh))
for rnd in range(2):
    for k, (nm, sh) in enumerate(sets):
        f, cn, en, ch = timed(nm, sh)
        pk = getattr(nm, E._PACKED_ATTR).words
        print("round %d set %d  matrix 0x%x shifts 0x%x companion 0x%x   fill %.4f  cn %.4f  energy %.4f  chain %.4f ms" %
              (rnd, k, nm.data_ptr(), sh.data_ptr(), pk.data_ptr(), f, cn, en, ch), flush=True)
# which of the three buffers carries the effect?  best set's matrix with the worst set's shifts, and the other way round

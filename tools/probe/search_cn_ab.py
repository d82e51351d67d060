"""Same-process A/B of the coordination numbers summed by the neighbour search (round 6): the headline D3 branch -- 40-Bohr
list + companion into fixed buffers, then dftd3 with virial -- timed with the search's CN by-product off and on, alternating, per-kernel
HIP-event times from the library.  Same buffers in both modes.  Kill criterion of the verdict: the fill must not grow by more than 0.25 ms.

    python tools/probe/search_cn_ab.py [--atoms 100000] [--reps 30] [--rounds 2]"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops import _capi as C  # noqa: E402
from nvalchemiops.interactions.dispersion import D3Parameters, dftd3  # noqa: E402
from nvalchemiops.neighborlist import _engine as E  # noqa: E402
from nvalchemiops.neighborlist import cell_list  # noqa: E402
from tests import systems as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--atoms", type=int, default=100000)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--m", type=int, default=2560)
args = ap.parse_args()
dev = torch.device("cuda:0")
BOHR = 1.8897261246
n = args.atoms
pos, cell, q, numbers = S.fcc_box(n, seed=1234, dtype=np.float64)
tables = S.d3_test_tables(94, seed=7)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
p32, c32, z = t((pos * BOHR).astype(np.float32)), t((cell * BOHR).astype(np.float32)), t(numbers)
params = D3Parameters(rcov=t(tables["rcov"]), r4r2=t(tables["r4r2"]), c6ab=t(tables["c6ab"]), cn_ref=t(tables["cn_ref"]))
pbc = torch.tensor([True] * 3, device=dev)
M = args.m
nm = torch.empty((n, M), dtype=torch.int32, device=dev)
sh = torch.empty((n, M, 3), dtype=torch.int32, device=dev)
num = torch.empty(n, dtype=torch.int32, device=dev)
E._PACKED_POLICY = "1"
ctx = E.D3SearchContext(z, params.rcov, 16.0)


def report():
    buf = ctypes.create_string_buffer(1 << 16)
    C.lib().mi_timing_report_stats(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot, med, lo, hi = line.rsplit(" ", 5)
        out[name] = (int(cnt), float(med), float(lo), float(hi))
    return out


def branch():
    cell_list(p32, 40.0, c32, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
    return dftd3(p32, z, a1=0.4289, a2=4.4407, s8=0.7875, d3_params=params, neighbor_matrix=nm, neighbor_matrix_shifts=sh, fill_value=n,
                 cell=c32.unsqueeze(0), compute_virial=True, num_systems=1)


res, outs = {}, {}
for rnd in range(args.rounds):
    for mode in ("off", "on"):
        if mode == "on":
            setattr(nm, E._D3CTX_ATTR, ctx)
        elif hasattr(nm, E._D3CTX_ATTR):
            delattr(nm, E._D3CTX_ATTR)
        for _ in range(3):
            out = branch()
        torch.cuda.synchronize()
        assert (getattr(nm, E._PACKED_ATTR).cn is not None) == (mode == "on")
        C.lib().mi_timing_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = branch()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.reps * 1e3
        C.lib().mi_timing_enable(0)
        k = report()
        res[f"round{rnd}_search_cn_{mode}"] = {"branch_ms_wall": round(wall, 4), **{kk: {"median_ms": v[1], "min_ms": v[2], "max_ms": v[3]} for kk, v in k.items()
                                                                                      if kk.startswith(("d3_", "nl_query_matrix", "nl_build"))}}
        outs[mode] = [o.clone() for o in out]
    d = {name: float((a.double() - b.double()).abs().max()) for name, a, b in zip(("energy", "forces", "cn", "virial"), outs["off"], outs["on"])}
    res[f"round{rnd}_max_abs_diff_on_vs_off"] = d
    res[f"round{rnd}_scales"] = {name: float(a.abs().max()) for name, a in zip(("energy", "forces", "cn", "virial"), outs["off"])}
hdr = getattr(nm, E._PACKED_ATTR).cn[:16].view(torch.int32)
res["cn_flag"] = int(hdr[0])
res["num_max"] = int(num.max())
print(json.dumps(res, indent=1))

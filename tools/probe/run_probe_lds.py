"""A/B on the real 40-Bohr headline list: D3 chain walk with L2 gathers (library, 4 B/slot packed list) vs records staged in LDS
(probe_lds.hip, 2 B/slot neighbourhood-local slots).  Prints the library's own cn / energy / chain times on the same box first.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe/libprobe_lds.so tools/probe/probe_lds.hip
    python tools/probe/run_probe_lds.py [n_atoms]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from tests import systems as S  # noqa: E402
from nvalchemiops import _capi as C  # noqa: E402
from nvalchemiops.interactions.dispersion import D3Parameters, dftd3  # noqa: E402
from nvalchemiops.neighborlist import cell_list  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools/probe/libprobe_lds.so"))
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
M = 2560
BOHR = 1.8897261246
RC = 40.0
pos, cell, _, z = S.fcc_box(n, seed=1234, dtype=np.float64)
pb, cb = (pos * BOHR).astype(np.float32), (cell * BOHR).astype(np.float32)
tp, tc = torch.as_tensor(pb, device=dev), torch.as_tensor(cb, device=dev)
pbc = torch.tensor([True] * 3, device=dev)
nm = torch.empty((n, M), dtype=torch.int32, device=dev)
nsh = torch.empty((n, M, 3), dtype=torch.int32, device=dev)
num = torch.empty(n, dtype=torch.int32, device=dev)
cell_list(tp, RC, tc, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=nsh, num_neighbors=num)
torch.cuda.synchronize()
pairs = int(num.sum().item())
print(f"atoms {n}  pairs {pairs}  max row {int(num.max())}  M {M}", flush=True)


def timed(fn, reps=7):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), min(ts)


# ---- the library on this box --------------------------------------------------------------------------------------------------
from tests.systems import d3_test_tables  # noqa: E402

tables = d3_test_tables(94, seed=7)
t = lambda a: torch.as_tensor(a, device=dev)  # noqa: E731
params = D3Parameters(rcov=t(tables["rcov"]), r4r2=t(tables["r4r2"]), c6ab=t(tables["c6ab"]), cn_ref=t(tables["cn_ref"]))
numbers = t(z)
kw = dict(a1=0.4289, a2=4.4407, s8=0.7875, d3_params=params, cell=tc.unsqueeze(0), compute_virial=True, num_systems=1, neighbor_matrix=nm,
          neighbor_matrix_shifts=nsh, fill_value=n)
for _ in range(3):
    dftd3(tp, numbers, **kw)
torch.cuda.synchronize()
C.lib().mi_timing_enable(1)
for _ in range(7):
    dftd3(tp, numbers, **kw)
torch.cuda.synchronize()
C.lib().mi_timing_enable(0)
buf = ctypes.create_string_buffer(1 << 16)
C.lib().mi_timing_report_stats(buf, len(buf))
for line in buf.value.decode().splitlines():
    name, cnt, tot, med, lo, hi = line.rsplit(" ", 5)
    print(f"library {name}: median {float(med):.3f} ms  min {float(lo):.3f}", flush=True)

# ---- binning (torch; the production path would do this with binsort.h) ---------------------------------------------------------
L = float(cb[0, 0])
ncell = int(L / (RC * (1 + 2e-6) / 3))
R = int(np.ceil(RC * ncell / L * (1 + 1e-6)))
assert ncell <= 128 and 2 * R + 1 <= ncell, (ncell, R)
c = torch.clamp((tp / L * ncell).floor().to(torch.int64), 0, ncell - 1)  # clamped, not wrapped: the grid is a lookup structure only
key = c[:, 0] + ncell * (c[:, 1] + ncell * c[:, 2])
order = torch.argsort(key, stable=True)
counts = torch.bincount(key, minlength=ncell ** 3)
cell_start = torch.zeros(ncell ** 3 + 1, dtype=torch.int64, device=dev)
cell_start[1:] = torch.cumsum(counts, 0)
rank_sorted = torch.arange(n, device=dev) - cell_start[key[order]]
rank = torch.empty(n, dtype=torch.int64, device=dev)
rank[order] = rank_sorted
assert int(rank.max()) < 128, int(rank.max())
spec = (numbers == 8).to(torch.int64)  # two species
meta = (c[:, 0] | (c[:, 1] << 7) | (c[:, 2] << 14) | (rank << 21) | (spec << 28)).to(torch.int32)
apos = torch.cat([tp, meta.view(torch.float32).unsqueeze(1)], 1).contiguous()
g = torch.Generator(device=dev).manual_seed(5)
dEdCN = (torch.randn(n, device=dev, generator=g) * 1e-3).float()
srec = torch.cat([tp[order], dEdCN[order].unsqueeze(1)], 1).contiguous()
sspec = spec[order].to(torch.uint8).contiguous()
sidx = order.to(torch.int32).contiguous()
cs32 = cell_start.to(torch.int32).contiguous()
rcov_tab = torch.zeros(16, device=dev)
rcov_tab[0], rcov_tab[1] = 1.45, 1.25
print(f"grid {ncell}^3  R {R}  atoms/cell mean {n / ncell ** 3:.1f} max {int(counts.max())}  neighbourhood max "
      f"{int(counts.max()) * (2 * R + 1) ** 3} (bound)", flush=True)

Mp = (M + 255) // 256 * 256
pk = torch.full((n, Mp), 0xFFFF, dtype=torch.int32, device=dev).to(torch.int16)  # 2 B / slot
cn = torch.zeros(n, device=dev)
flags = torch.zeros(4, dtype=torch.int32, device=dev)
work = torch.zeros(4, dtype=torch.int32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731
f = ctypes.c_float


def translate(do_pack):
    rc = lib.probe_translate(P(apos), n, P(nm), P(nsh), M, Mp, P(cs32), ncell, ncell, ncell, R, f(L), f(L), f(L), P(rcov_tab), f(16.0), P(pk), P(cn), P(flags),
                             do_pack, st)
    assert rc == 0


med0, lo0 = timed(lambda: translate(0))
print(f"probe cn walk, no packed write        : median {med0:.3f} ms  min {lo0:.3f}", flush=True)
flags.zero_()
med1, lo1 = timed(lambda: translate(1))
print(f"probe cn walk + 2 B/slot slot list    : median {med1:.3f} ms  min {lo1:.3f}   (library d3_cn writes 4 B/slot)", flush=True)
print("out-of-window entries per launch:", int(flags[0].item()) // 9, flush=True)

F_ref = torch.zeros((n, 3), device=dev)
V_ref = torch.zeros((n, 9), dtype=torch.float64, device=dev)
rc = lib.probe_chain_ref(P(apos), P(dEdCN), n, P(nm), P(nsh), M, f(L), f(L), f(L), P(rcov_tab), f(16.0), 1, P(F_ref), P(V_ref), st)
assert rc == 0
torch.cuda.synchronize()

for blocks in (256, 512):
    F = torch.zeros((n, 3), device=dev)
    V = torch.zeros((n, 9), dtype=torch.float64, device=dev)
    flags.zero_()

    def chain():
        rc = lib.probe_chain_lds(P(srec), P(sspec), P(sidx), P(cs32), ncell, ncell, ncell, R, f(L), f(L), f(L), P(pk), Mp, P(rcov_tab), f(16.0), 1, P(F), P(V),
                                 P(work), P(flags), blocks, st)
        assert rc == 0

    med, lo = timed(chain)
    torch.cuda.synchronize()
    dF = (F - F_ref).abs().max().item()
    dV = (V - V_ref).abs().max().item()
    print(f"probe chain, LDS-staged, {blocks} blocks x 16 waves: median {med:.3f} ms  min {lo:.3f}   max|dF| {dF:.3e} (max|F| {F_ref.abs().max().item():.3e})  "
          f"max|dV| {dV:.3e}  cells over LDS capacity: {int(flags[1].item())}", flush=True)


# ---- v2: pruned neighbourhood, 20 B records staged by global_load_lds, 16-bit word = slot | shifted-axis bits, pipelined LDS reads ----
wcell = L / ncell
dxlim = np.full((5, 5), -1, np.int8)
for az in range(R + 1):
    for ay in range(R + 1):
        gz, gy = max(az - 1, 0) * wcell, max(ay - 1, 0) * wcell
        rem = RC * RC - gz * gz - gy * gy
        if rem < -1e-6 * RC * RC:
            continue
        dxlim[az, ay] = min(R, int(np.sqrt(max(rem, 0.0)) / wcell * (1 + 1e-6)) + 1)
ncells_kept = sum(2 * int(dxlim[abs(dz), abs(dy)]) + 1 for dz in range(-R, R + 1) for dy in range(-R, R + 1) if dxlim[abs(dz), abs(dy)] >= 0)
print(f"pruned neighbourhood: {ncells_kept} of {(2 * R + 1) ** 3} cells -> ~{ncells_kept * n / ncell ** 3:.0f} records per block", flush=True)
dx_buf = (ctypes.c_byte * 25)(*[int(v) for v in dxlim.reshape(-1)])
srcov = rcov_tab[spec[order]].contiguous()
pk2 = torch.full((n, Mp // 2), -1, dtype=torch.int32, device=dev)
rowflag = torch.zeros(n, dtype=torch.int32, device=dev)
flags.zero_()


def translate2():
    rc = lib.probe_translate2(P(apos), n, P(nm), P(nsh), M, Mp, P(cs32), ncell, ncell, ncell, R, f(L), f(L), f(L), dx_buf, P(rcov_tab), f(16.0), P(pk2), P(cn),
                              P(flags), P(rowflag), st)
    assert rc == 0


med2, lo2 = timed(translate2)
print(f"probe cn walk + 2 B/slot list (v2: 4-byte stores, shifted-axis bits): median {med2:.3f} ms  min {lo2:.3f}   rows flagged for the plain path: "
      f"{int(rowflag.sum().item())}", flush=True)
for blocks in (256,):
    F = torch.zeros((n, 3), device=dev)
    V = torch.zeros((n, 9), dtype=torch.float64, device=dev)
    flags.zero_()

    def chain2():
        rc = lib.probe_chain_lds2(P(srec), P(srcov), P(sidx), P(cs32), ncell, ncell, ncell, R, f(L), f(L), f(L), dx_buf, P(pk2), Mp, f(16.0), 1, P(F), P(V),
                                  P(work), P(flags), blocks, st)
        assert rc == 0

    med, lo = timed(chain2)
    torch.cuda.synchronize()
    dF = (F - F_ref).abs().max().item()
    dV = (V - V_ref).abs().max().item()
    print(f"probe chain v2, LDS-staged (DMA staging, pipelined), {blocks} blocks x 16 waves: median {med:.3f} ms  min {lo:.3f}   max|dF| {dF:.3e} "
          f"(max|F| {F_ref.abs().max().item():.3e})  max|dV| {dV:.3e} (max|V| {V_ref.abs().max().item():.3e})  cells over LDS capacity per launch: "
          f"{int(flags[1].item()) // 9}", flush=True)

# VALU sensitivity of the LDS-staged walk: the same kernel without the virial sums (9 of ~28 accumulations per pair)
def chain2_nv():
    rc = lib.probe_chain_lds2(P(srec), P(srcov), P(sidx), P(cs32), ncell, ncell, ncell, R, f(L), f(L), f(L), dx_buf, P(pk2), Mp, f(16.0), 0, P(F), P(V),
                              P(work), P(flags), 256, st)
    assert rc == 0


med, lo = timed(chain2_nv)
print(f"probe chain v2 without the virial sums: median {med:.3f} ms  min {lo:.3f}", flush=True)
kw_nv = dict(kw, compute_virial=False)
C.lib().mi_timing_enable(1)
for _ in range(7):
    dftd3(tp, numbers, **kw_nv)
torch.cuda.synchronize()
C.lib().mi_timing_enable(0)
C.lib().mi_timing_report_stats(buf, len(buf))
for line in buf.value.decode().splitlines():
    name, cnt, tot, med, lo, hi = line.rsplit(" ", 5)
    print(f"library without virial {name}: median {float(med):.3f} ms  min {float(lo):.3f}", flush=True)

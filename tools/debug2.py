import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from oracle import oracle as O
from tests import systems as S
from nvalchemiops.neighborlist import cell_list, naive_neighbor_list
from nvalchemiops.interactions.dispersion import dftd3, D3Parameters
dev = "cuda:0"
_t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
# ---- naive triclinic partial pbc
for dtype in (np.float64,):
    pos, cell = S.random_box(300, 9.0, seed=4, dtype=dtype, triclinic=True)
    for pbc in ([True, True, False], [True, True, True]):
        onm, onum, osh = O.naive(pos, 3.3, cell, pbc, max_neighbors=64)
        nm, num, sh = naive_neighbor_list(_t(pos), 3.3, cell=_t(cell), pbc=torch.tensor(pbc, device=dev), max_neighbors=64)
        a = set(map(tuple, O.canonical_pairs(nm.cpu().numpy(), num.cpu().numpy(), sh.cpu().numpy()).tolist()))
        b = set(map(tuple, O.canonical_pairs(onm, onum, osh).tolist()))
        bf = set(map(tuple, O.brute_force_pairs(pos, 3.3, cell, pbc).tolist()))
        cnm, cnum, csh = O.cell_list(pos, 3.3, cell, pbc, max_neighbors=64)
        c = set(map(tuple, O.canonical_pairs(cnm, cnum, csh).tolist()))
        print("naive", pbc, "gpu", len(a), "oracle-naive", len(b), "brute", len(bf), "oracle-cell", len(c), "gpu^brute", len(a ^ bf), "oracle^brute", len(b ^ bf))
        for tag, dset in (("only gpu", a - b), ("only oracle", b - a)):
            for (i, j, x, y, z) in sorted(dset)[:6]:
                d = pos[j] - pos[i] + np.array([x, y, z], float) @ cell
                print("  ", tag, i, j, (x, y, z), "dist", np.sqrt((d * d).sum()))
# ---- d3 periodic
t = O.d3_test_tables(17)
p = D3Parameters(rcov=_t(t["rcov"]), r4r2=_t(t["r4r2"]), c6ab=_t(t["c6ab"]), cn_ref=_t(t["cn_ref"]))
FP = dict(a1=0.4, a2=4.0, s8=0.8, k1=16.0, k3=-4.0, s6=1.0)
pos, cell = S.random_box(180, 14.0, seed=3, dtype=np.float64, triclinic=True)
z = np.random.default_rng(1).choice(np.array([1, 6, 8, 17], np.int32), 180)
nm, num, sh = cell_list(_t(pos), 9.0, _t(cell), torch.tensor([True] * 3, device=dev), max_neighbors=320)
print("max num", int(num.max()))
ref = O.dftd3(pos, z, t, neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy(), cell=cell, compute_virial=True, **FP)
out = dftd3(_t(pos), _t(z), d3_params=p, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=_t(cell)[None], compute_virial=True, **FP)
f, rf = out[1].cpu().numpy(), ref[1]
err = np.abs(f - rf)
print("E", out[0].cpu().numpy(), ref[0], "max|F|", np.abs(rf).max(), "max err", err.max(), "at", np.unravel_index(err.argmax(), err.shape))
worst = np.argsort(-err.max(1))[:5]
for i in worst:
    nb = nm[i, : int(num[i])].cpu().numpy()
    d = pos[nb] - pos[i] + sh[i, : int(num[i])].cpu().numpy() @ cell
    r = np.sqrt((d * d).sum(1))
    print(" atom", i, "Z", z[i], "F gpu", f[i], "ref", rf[i], "min r", r.min(), "cn gpu", out[2][i].item(), "ref", ref[2][i])
print("cn err", np.abs(out[2].cpu().numpy() - ref[2]).max(), "virial err", np.abs(out[3].cpu().numpy() - ref[3]).max())
# non periodic same geometry
ref2 = O.dftd3(pos, z, t, neighbor_matrix=nm.cpu().numpy(), **FP)
out2 = dftd3(_t(pos), _t(z), d3_params=p, neighbor_matrix=nm, **FP)
print("non-periodic: F err", np.abs(out2[1].cpu().numpy() - ref2[1]).max(), "max|F|", np.abs(ref2[1]).max())

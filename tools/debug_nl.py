import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from oracle import oracle as O
from tests import systems as S
from nvalchemiops.neighborlist import cell_list
dev = "cuda:0"
for dtype in (np.float32, np.float64):
    pos, cell = S.random_box(700, 14.0, seed=42, dtype=dtype)
    for cutoff, m in ((3.2, 64), (6.5, 320)):
        onm, onum, osh = O.cell_list(pos, cutoff, cell, [True]*3, max_neighbors=m)
        nm, num, sh = cell_list(torch.as_tensor(pos, device=dev), cutoff, torch.as_tensor(cell, device=dev), torch.tensor([True]*3, device=dev), max_neighbors=m)
        a = O.canonical_pairs(nm.cpu().numpy(), num.cpu().numpy(), sh.cpu().numpy())
        b = O.canonical_pairs(onm, onum, osh)
        sa = set(map(tuple, a.tolist())); sb = set(map(tuple, b.tolist()))
        print(dtype.__name__, cutoff, len(a), len(b), len(sa), len(sb), "only gpu", len(sa - sb), "only oracle", len(sb - sa))
        for tag, dset in (("gpu", sa - sb), ("oracle", sb - sa)):
            for (i, j, x, y, z) in sorted(dset)[:8]:
                d = pos[j].astype(np.float64) - pos[i].astype(np.float64) + np.array([x, y, z], float) @ cell.astype(np.float64)
                print("  ", tag, i, j, (x, y, z), "dist", np.sqrt((d * d).sum()), "pos_i", pos[i], "pos_j", pos[j])
        bf = O.brute_force_pairs(pos, cutoff, cell, [True]*3)
        sbf = set(map(tuple, bf.tolist()))
        print("   brute: vs gpu", len(sbf ^ sa), "vs oracle", len(sbf ^ sb))

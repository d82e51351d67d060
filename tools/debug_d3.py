"""Debug aid: HIP D3 vs oracle on a small periodic box, prints the largest deviations."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from oracle import oracle as O
from tests import systems as S
from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
from nvalchemiops.neighborlist import cell_list

dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pos, cell, _, z = S.fcc_box(n, a=7.5, dtype=np.float32)
tabs = O.d3_test_tables(17)
prm = D3Parameters(**{k: torch.as_tensor(v, device=dev) for k, v in tabs.items()})
tp, tc = torch.as_tensor(pos, device=dev), torch.as_tensor(cell, device=dev)
pbc = torch.tensor([True] * 3, device=dev)
nm, num, sh = cell_list(tp, 20.0, tc, pbc, max_neighbors=1200)
print("max nbr", int(num.max()))
for rep in range(2):
    e, f, cn = dftd3(tp, torch.as_tensor(z, device=dev), 0.4289, 4.4407, 0.7875, d3_params=prm, neighbor_matrix=nm, neighbor_matrix_shifts=sh,
                     cell=tc.unsqueeze(0), fill_value=n)
    oe, of, ocn = O.dftd3(pos, z, tabs, 0.4289, 4.4407, 0.7875, neighbor_matrix=nm.cpu().numpy(), neighbor_matrix_shifts=sh.cpu().numpy(),
                          cell=cell[None], fill_value=n)[:3]
    print("E", e.item(), float(np.asarray(oe).sum()), "cn err", np.abs(cn.cpu().numpy() - ocn).max(), "F err", np.abs(f.cpu().numpy() - of).max(),
          "F max", np.abs(of).max())

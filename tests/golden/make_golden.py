"""Writes the committed fixtures of tests/golden/ (run from the repo root: `python tests/golden/make_golden.py`).

1. reference_known_answers.json -- inputs and expected outputs that the REFERENCE's own test-suite holds for this path, with the
   file:line each comes from (data values only; the reference itself cannot be imported here: Warp is absent).
2. oracle_vectors.npz -- seeded inputs and the outputs of the pinned CPU oracle (oracle/), so the GPU parity tests also check the
   HIP path against committed vectors (periodic D3 with virial, PME energies/forces, explicit-k Ewald, a triclinic neighbour list).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import systems as S  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

known = {
    "provenance": "data values transcribed from the reference test-suite (NVIDIA/nvalchemi-toolkit-ops @ 2026-01-09); paths relative to the reference root",
    "neighbor_counts": {
        "source": "test/neighborlist/test_cell_list.py:391-419, test/neighborlist/test_batch_cell_list.py:516-541, crystals from test/neighborlist/test_utils.py:252-301",
        "HoTlPd": {"positions": S.HOTLPD_POS.tolist(), "cell": S.HOTLPD_CELL.tolist(), "pbc": [True, True, True],
                   "num_neighbors": {str(k): v for k, v in S.HOTLPD_COUNTS.items()}},
        "SiCu": {"positions": S.SICU_POS.tolist(), "cell": S.SICU_CELL.tolist(), "pbc": [True, True, True],
                 "num_neighbors": {str(k): v for k, v in S.SICU_COUNTS.items()}},
    },
    "two_atom_pbc_pair": {"source": "test/neighborlist/test_cell_list.py:83-101", "positions": [[0.5, 5.0, 5.0], [9.5, 5.0, 5.0]], "box": 10.0,
                          "cutoff": 2.0, "num_neighbors": [1, 1], "shifts": [[-1, 0, 0], [1, 0, 0]]},
    "dftd3": {
        "source": "test/interactions/dispersion/conftest.py:38-208 (analytic tables, functional parameters), :641-730 (expected outputs)",
        "functional": {"a1": 0.4, "a2": 4.0, "s8": 0.8, "k1": 16.0, "k3": -4.0, "s6": 1.0},
        "Ne2": {"positions_bohr": [[0, 0, 0], [5.8, 0, 0]], "numbers": [10, 10], "energy": [-1.4161492698e-02],
                "coord_num": [4.4183229329e-04, 4.4183229329e-04], "forces": [[3.2497653738e-03, 0, 0], [-3.2497653738e-03, 0, 0]]},
        "HCl_dimer": {"positions_bohr": [[0, 0, 0], [2.4, 0, 0], [0, 7, 0], [2.4, 7, 0]], "numbers": [1, 17, 1, 17],
                      "energy": [-2.2127663717e-02], "coord_num": [5.0002193451e-01, 5.0044161081e-01, 5.0002193451e-01, 5.0044161081e-01],
                      "forces": [[6.2320637517e-03, 8.8818743825e-04, 0], [-6.2320632860e-03, 1.9026985392e-03, 0],
                                 [6.2320632860e-03, -8.8818743825e-04, 0], [-6.2320632860e-03, -1.9026985392e-03, 0]]},
    },
    "max_neighbors_rule": {"source": "nvalchemiops/neighborlist/neighbor_utils.py:296-340 and README example", "5.0": 928, "6.0": 1584},
}
json.dump(known, open(os.path.join(HERE, "reference_known_answers.json"), "w"), indent=1)

out = {}
FP = dict(a1=0.4, a2=4.0, s8=0.8, k1=16.0, k3=-4.0, s6=1.0)
# D3, periodic triclinic, matrix format, with virial
pos, cell = S.random_box(120, 24.0, seed=31, dtype=np.float32, triclinic=True)
z = np.random.default_rng(4).choice(np.array([1, 6, 8, 17], np.int32), 120)
nm, num, sh = O.cell_list(pos, 12.0, cell, [True] * 3, max_neighbors=256)
assert num.max() < 256
e, f, cn, vir = O.dftd3(pos, z, O.d3_test_tables(17), neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=cell, compute_virial=True, **FP)
out.update(d3_pos=pos, d3_cell=cell, d3_numbers=z, d3_nm=nm, d3_num=num, d3_shifts=sh, d3_energy=e, d3_forces=f, d3_cn=cn, d3_virial=vir)
# PME, fp64, triclinic, order 4, energies + forces
g = np.random.default_rng(12)
cellp = np.array([[11.0, 0, 0], [2.0, 10.0, 0], [1.0, -1.5, 12.0]])
posp = g.uniform(0, 1, (90, 3)) @ cellp
q = g.normal(size=90)
q -= q.mean()
nmp, nump, shp = O.cell_list(posp, 8.0, cellp, [True] * 3, max_neighbors=200)
assert nump.max() < 200
ep, fpm = O.particle_mesh_ewald(posp, q, cellp, 0.4, (24, 24, 24), 4, neighbor_matrix=nmp, neighbor_matrix_shifts=shp, compute_forces=True)
out.update(pme_pos=posp, pme_cell=cellp, pme_q=q, pme_nm=nmp, pme_shifts=shp, pme_energies=ep, pme_forces=fpm)
# explicit-k Ewald reciprocal space on the same system
kv = O.generate_k_vectors_ewald_summation(cellp, 3.0)
er, fr, cg = O.ewald_reciprocal_space(posp, q, cellp, kv, 0.4)
out.update(ewald_kvec=kv, ewald_energies=er, ewald_forces=fr, ewald_cgrad=cg)
# neighbour list: triclinic box with atoms outside the cell, canonical (i, j, S) rows
posn, celln = S.random_box(200, 9.0, seed=77, dtype=np.float64, triclinic=True, outside=True)
nmn, numn, shn = O.cell_list(posn, 3.3, celln, [True, True, False], max_neighbors=128)
assert numn.max() < 128
out.update(nl_pos=posn, nl_cell=celln, nl_pairs=O.canonical_pairs(nmn, numn, shn).astype(np.int32), nl_num=numn)
np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
print("wrote", os.path.join(HERE, "reference_known_answers.json"), "and oracle_vectors.npz",
      os.path.getsize(os.path.join(HERE, "oracle_vectors.npz")), "bytes")

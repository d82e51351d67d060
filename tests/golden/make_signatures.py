"""Writes tests/golden/reference_signatures.json: the argument names and default values of the reference's public functions on the
hot path, read from the reference sources with `ast` (run in the build container, where /root/reference exists).  The drop-in claim
is checked against this file by tests/test_host_cpu.py::test_public_signatures_match_reference."""
import ast
import json
import os

REF = "/root/reference/nvalchemiops/"
HERE = os.path.dirname(os.path.abspath(__file__))
FUNCS = {
    "neighborlist/neighborlist.py": ["neighbor_list"],
    "neighborlist/cell_list.py": ["cell_list", "build_cell_list", "query_cell_list", "estimate_cell_list_sizes"],
    "neighborlist/batch_cell_list.py": ["batch_cell_list", "batch_build_cell_list", "batch_query_cell_list", "estimate_batch_cell_list_sizes"],
    "neighborlist/naive.py": ["naive_neighbor_list"],
    "neighborlist/batch_naive.py": ["batch_naive_neighbor_list"],
    "neighborlist/naive_dual_cutoff.py": ["naive_neighbor_list_dual_cutoff"],
    "neighborlist/batch_naive_dual_cutoff.py": ["batch_naive_neighbor_list_dual_cutoff"],
    "neighborlist/neighbor_utils.py": ["estimate_max_neighbors", "allocate_cell_list", "get_neighbor_list_from_neighbor_matrix", "compute_naive_num_shifts"],
    "neighborlist/rebuild_detection.py": ["cell_list_needs_rebuild", "neighbor_list_needs_rebuild", "check_cell_list_rebuild_needed",
                                          "check_neighbor_list_rebuild_needed"],
    "interactions/dispersion/dftd3.py": ["dftd3"],
    "interactions/electrostatics/pme.py": ["particle_mesh_ewald", "pme_reciprocal_space", "pme_green_structure_factor", "pme_energy_corrections",
                                           "pme_energy_corrections_with_charge_grad"],
    "interactions/electrostatics/ewald.py": ["ewald_real_space", "ewald_reciprocal_space", "ewald_summation"],
    "interactions/electrostatics/coulomb.py": ["coulomb_energy", "coulomb_forces", "coulomb_energy_forces"],
    "interactions/electrostatics/k_vectors.py": ["generate_k_vectors_pme", "generate_k_vectors_ewald_summation"],
    "interactions/electrostatics/parameters.py": ["estimate_pme_parameters", "estimate_ewald_parameters", "estimate_pme_mesh_dimensions",
                                                  "mesh_spacing_to_dimensions"],
    "spline.py": ["spline_spread", "spline_gather", "spline_gather_vec3", "spline_gather_gradient", "spline_spread_channels",
                  "spline_gather_channels", "compute_bspline_deconvolution", "compute_bspline_deconvolution_1d"],
}


def signature(path, name):
    for node in ast.walk(ast.parse(open(path).read())):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            a = node.args
            names = [x.arg for x in a.posonlyargs + a.args]
            defaults = [None] * (len(names) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
            return {"args": [[n, d] for n, d in zip(names, defaults)],
                    "kwonly": [[x.arg, ast.unparse(d) if d is not None else None] for x, d in zip(a.kwonlyargs, a.kw_defaults)],
                    "vararg": a.vararg.arg if a.vararg else None, "kwarg": a.kwarg.arg if a.kwarg else None}
    raise KeyError(f"{name} not found in {path}")


if __name__ == "__main__":
    out = {f: {n: signature(REF + f, n) for n in names} for f, names in FUNCS.items()}
    json.dump(out, open(os.path.join(HERE, "reference_signatures.json"), "w"), indent=1)
    print("wrote", sum(len(v) for v in out.values()), "signatures")

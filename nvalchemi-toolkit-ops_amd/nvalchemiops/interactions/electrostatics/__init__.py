"""Electrostatics on the MI355X hot path: PME = real-space erfc sum + B-spline/FFT reciprocal sum
(reference exports: interactions/electrostatics/__init__.py:33-80; plus the explicit-k Ewald sum; plain Coulomb is outside this path)."""
from nvalchemiops.interactions.electrostatics.ewald import ewald_real_space, ewald_reciprocal_space, ewald_summation
from nvalchemiops.interactions.electrostatics.k_vectors import generate_k_vectors_ewald_summation, generate_k_vectors_pme
from nvalchemiops.interactions.electrostatics.parameters import (EwaldParameters, PMEParameters, estimate_ewald_parameters,
                                                                 estimate_pme_mesh_dimensions, estimate_pme_parameters,
                                                                 mesh_spacing_to_dimensions)
from nvalchemiops.interactions.electrostatics.pme import (particle_mesh_ewald, pme_energy_corrections,
                                                          pme_energy_corrections_with_charge_grad, pme_green_structure_factor,
                                                          pme_reciprocal_space)

__all__ = [
    "particle_mesh_ewald", "pme_reciprocal_space", "ewald_real_space", "pme_green_structure_factor", "pme_energy_corrections",
    "pme_energy_corrections_with_charge_grad", "generate_k_vectors_pme", "estimate_pme_parameters", "estimate_ewald_parameters",
    "estimate_pme_mesh_dimensions", "mesh_spacing_to_dimensions", "PMEParameters", "EwaldParameters", "ewald_reciprocal_space", "ewald_summation", "generate_k_vectors_ewald_summation",
]

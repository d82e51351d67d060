"""Neighbour-driven interactions of the MI355X hot path: dispersion (DFT-D3) and electrostatics (PME)."""

"""DFT-D3(BJ) dispersion (reference: nvalchemiops/interactions/dispersion/__init__.py)."""
from nvalchemiops.interactions.dispersion.dftd3 import D3Parameters, dftd3

__all__ = ["D3Parameters", "dftd3"]

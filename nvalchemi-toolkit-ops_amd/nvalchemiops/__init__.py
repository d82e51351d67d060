"""nvalchemiops -- MI355X-native (gfx950) build of the nvalchemi-toolkit-ops hot path.

Drop-in for the reference's functional API on this path (same import paths, signatures and return tuples):

    from nvalchemiops.neighborlist import neighbor_list
    from nvalchemiops.interactions.dispersion import dftd3, D3Parameters
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald

Host code is Python on PyTorch-ROCm; every kernel is hand-written HIP in ``libnvalchemiops_hip.so`` reached
through a C ABI (``include/nvalchemiops_hip.h``) with ctypes.  There is no CPU fallback and no second backend:
tensors must live on a ROCm device and the library must be built (``build_native.py``), otherwise the ops raise.

Reference counterpart: nvalchemiops/__init__.py:16-26 (wp.init() at import; here the .so is loaded lazily).
"""
__version__ = "0.2.0+mi355x.1"

from nvalchemiops import _capi  # noqa: F401  (lazy: does not load the .so until first use)
from nvalchemiops import _ops  # noqa: F401,E402  registers torch.ops.nvalchemiops.* (reference op names) for the hot-path seam

"""DFT-D3(BJ) two-body dispersion -- drop-in for interactions/dispersion/dftd3.py of the reference
(`D3Parameters` :146-332, `dftd3` :2468-2874; ops `nvalchemiops::dftd3_nm` :1792, `::dftd3_nl` :2125).

Energies [num_systems], forces [N,3], coordination numbers [N] and (optionally) virials [num_systems,3,3], all
float32, from a FULL neighbour list given either as a padded neighbour matrix or as CSR (`neighbor_list[1]` +
`neighbor_ptr`).  The three passes (CN; C6 interpolation + BJ damping + energy + direct force + dE/dCN; chain-rule
force) run as hand-written HIP kernels (csrc/d3.hip) behind `mi_d3` of the C ABI.  As in the reference, positions
and cell are detached: explicit forces are returned, there is no autograd through D3.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import torch

from nvalchemiops import _capi as C

_FLOAT_TYPES = (torch.float32, torch.float64)


@dataclass
class D3Parameters:
    """Validated container of the element tables: ``rcov[Z+1]``, ``r4r2[Z+1]``, ``c6ab[Z+1,Z+1,m,m]``, ``cn_ref[Z+1,Z+1,m,m]``
    (index 0 = padding, ``m = interp_mesh = 5``).  Same checks and exception types as dftd3.py:221-280."""

    rcov: torch.Tensor
    r4r2: torch.Tensor
    c6ab: torch.Tensor
    cn_ref: torch.Tensor
    interp_mesh: int = 5

    def __post_init__(self) -> None:
        named = {"rcov": self.rcov, "r4r2": self.r4r2, "c6ab": self.c6ab, "cn_ref": self.cn_ref}
        for name, value in named.items():
            if not isinstance(value, torch.Tensor):
                raise TypeError(f"Parameter '{name}' must be a torch.Tensor, got {type(value)}")
            if value.dtype not in _FLOAT_TYPES:
                raise TypeError(f"Parameter '{name}' must be float32 or float64, got {value.dtype}")
        if self.rcov.ndim != 1:
            raise ValueError(f"rcov must be 1D tensor [max_Z+1], got shape {self.rcov.shape}")
        nz = self.rcov.size(0)
        if nz < 2:
            raise ValueError(f"rcov must have at least 2 elements (padding + 1 element), got {nz}")
        if self.r4r2.shape != (nz,):
            raise ValueError(f"r4r2 must have shape [{nz}] to match rcov, got {self.r4r2.shape}")
        grid = (nz, nz, self.interp_mesh, self.interp_mesh)
        if self.c6ab.shape != grid:
            raise ValueError(f"c6ab must have shape {grid}, got {self.c6ab.shape}")
        if self.cn_ref.shape != grid:
            raise ValueError(f"cn_ref must have shape {grid}, got {self.cn_ref.shape}")
        if len({str(v.device) for v in named.values()}) > 1:
            raise ValueError("All parameters must be on the same device. Got devices: "
                             + ", ".join(f"{k}={v.device}" for k, v in named.items()))

    @property
    def max_z(self) -> int:
        return self.rcov.size(0) - 1

    @property
    def device(self) -> torch.device:
        return self.rcov.device

    def to(self, device: str | torch.device | None = None, dtype: torch.dtype | None = None) -> "D3Parameters":
        mv = lambda t: t.to(device=device, dtype=dtype)  # noqa: E731
        return D3Parameters(rcov=mv(self.rcov), r4r2=mv(self.r4r2), c6ab=mv(self.c6ab), cn_ref=mv(self.cn_ref),
                            interp_mesh=self.interp_mesh)


_LIB_OVERRIDE = None  # tests only: a ctypes handle of the IEEE-arithmetic build of d3.hip (error budget, tests/test_d3_gpu.py)


def _launch(positions, numbers, idx, shifts, nptr, max_neighbors, fill_value, cell, batch_idx, num_systems, tables, scalars,
            compute_virial, energy, forces, coord_num, virial, packed=None) -> None:
    """`packed`: the companion record the neighbour search left next to (idx, shifts) (`neighborlist/_engine.py`: `.words`, and `.cn` when the
    search also summed the coordination numbers), already validated by the caller against tensor identity / versions; the passes then stream
    4 B/slot (`mi_d3_packed_cn`), the CN pass is skipped when the device-side fingerprint check lets the search's numbers in, and every call
    re-derives a rotating sample of the companion's rows from (idx, shifts) on the device before trusting it."""
    dev = positions.device
    n = positions.shape[0]
    pos = positions.detach().contiguous()
    code = C.dtype_code(pos.dtype)
    f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731  (dftd3.py:1912-1915)
    rcov, r4r2, c6ab, cnref = (f32(t) for t in tables)
    if c6ab.shape[-1] != 5 or c6ab.shape[-2] != 5:
        raise ValueError("this build supports the standard 5x5 CN interpolation mesh only")
    par = C.MiD3Params(rcov=rcov.data_ptr(), r4r2=r4r2.data_ptr(), c6ab=c6ab.data_ptr(), cn_ref=cnref.data_ptr(), nz=rcov.shape[0],
                       **{k: float(v) for k, v in scalars.items()})
    periodic = cell is not None and shifts is not None
    cell_t = cell.detach().to(dtype=pos.dtype, device=dev).reshape(-1, 3, 3).contiguous() if periodic else None
    sh = C.i32(shifts.to(dev)) if periodic else None
    bi = None if batch_idx is None else C.i32(batch_idx)
    # a periodic padded matrix is streamed by all three passes: the larger workspace lets the CN pass leave a 4 B/slot copy for the others
    # NVALCHEMIOPS_D3_PACKED_LIST: "1" (default) padded matrix only; "0" never (A/B on one box); "2" also CSR lists -- measured neutral
    # there (unaligned rows make the CN pass's extra write cost what the other two passes gain), so it is not the default
    mode = os.environ.get("NVALCHEMIOPS_D3_PACKED_LIST", "1")
    pack = periodic and mode != "0" and (nptr is None or mode == "2")
    n_entries = (idx.shape[0] if nptr is not None else n * int(max_neighbors)) if pack else 0
    L = _LIB_OVERRIDE or C.lib()
    ws_bytes = int(C.lib().mi_d3_workspace_bytes_entries(n, num_systems, rcov.shape[0], int(n_entries)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    vir = virial if compute_virial else None
    z = C.i32(numbers)  # converted tensors stay referenced until the launch is enqueued (the allocator may otherwise reuse their blocks)
    if packed is not None and periodic and nptr is None and mode != "0":
        from nvalchemiops.neighborlist import _engine as E

        words = packed.words
        cn = packed.cn if os.environ.get("NVALCHEMIOPS_D3_SEARCH_CN", "1") != "0" else None  # (A/B switch: "0" ignores the search's coordination numbers)
        stride, phase = E.verify_args()
        rc = L.mi_d3_packed_cn(C.ptr(pos), C.ptr(z), n, code, C.ptr(idx), C.ptr(sh), int(max_neighbors), int(fill_value), C.ptr(cell_t), C.ptr(bi),
                               int(num_systems), ctypes.byref(par), int(bool(compute_virial)), C.ptr(energy), C.ptr(forces), C.ptr(coord_num),
                               C.ptr(vir), C.ptr(ws), ctypes.c_size_t(ws_bytes), C.ptr(words), ctypes.c_size_t(words.numel() * words.element_size()),
                               C.ptr(cn), ctypes.c_size_t(cn.numel() if cn is not None else 0), int(stride), int(phase), C.stream_of(pos))
        if rc != 0 and _LIB_OVERRIDE is not None:
            raise C.NativeLibraryError(f"mi_d3_packed_cn (override library) failed with code {rc}")
        C.check(rc, "mi_d3_packed_cn")
        return
    rc = L.mi_d3(C.ptr(pos), C.ptr(z), n, code, C.ptr(idx), C.ptr(sh), C.ptr(nptr), int(max_neighbors),
                 ctypes.c_longlong(idx.shape[0] if nptr is not None else 0), int(fill_value),  # CSR: the entry count (packing is decided by the workspace size)
                 C.ptr(cell_t), C.ptr(bi), int(num_systems), ctypes.byref(par), int(bool(compute_virial)), C.ptr(energy), C.ptr(forces),
                 C.ptr(coord_num), C.ptr(vir), C.ptr(ws), ctypes.c_size_t(ws_bytes), C.stream_of(pos))
    if rc != 0 and _LIB_OVERRIDE is not None:
        raise C.NativeLibraryError(f"mi_d3 (override library) failed with code {rc}")
    C.check(rc, "mi_d3")


@C.hybrid
def dftd3(positions: torch.Tensor, numbers: torch.Tensor, a1: float, a2: float, s8: float, k1: float = 16.0, k3: float = -4.0,
          s6: float = 1.0, s5_smoothing_on: float = 1e10, s5_smoothing_off: float = 1e10, fill_value: int | None = None,
          d3_params: D3Parameters | dict[str, torch.Tensor] | None = None, covalent_radii: torch.Tensor | None = None,
          r4r2: torch.Tensor | None = None, c6_reference: torch.Tensor | None = None, coord_num_ref: torch.Tensor | None = None,
          batch_idx: torch.Tensor | None = None, cell: torch.Tensor | None = None, neighbor_matrix: torch.Tensor | None = None,
          neighbor_matrix_shifts: torch.Tensor | None = None, neighbor_list: torch.Tensor | None = None,
          neighbor_ptr: torch.Tensor | None = None, unit_shifts: torch.Tensor | None = None, compute_virial: bool = False,
          num_systems: int | None = None, device: str | None = None):
    """Returns ``(energy[num_systems], forces[N,3], coord_num[N])`` (+ ``virial[num_systems,3,3]`` if ``compute_virial``).

    Validation, parameter resolution, num_systems inference and empty-input behaviour follow dftd3.py:2668-2804."""
    use_matrix, use_list = neighbor_matrix is not None, neighbor_list is not None
    if use_matrix and use_list:
        raise ValueError("Cannot provide both neighbor_matrix and neighbor_list. Please provide only one neighbor representation format.")
    if not use_matrix and not use_list:
        raise ValueError("Must provide either neighbor_matrix or neighbor_list.")
    if use_matrix and unit_shifts is not None:
        raise ValueError("unit_shifts is for neighbor_list format. Use neighbor_matrix_shifts for neighbor_matrix format.")
    if use_list and neighbor_matrix_shifts is not None:
        raise ValueError("neighbor_matrix_shifts is for neighbor_matrix format. Use unit_shifts for neighbor_list format.")
    if use_list and neighbor_ptr is None:
        raise ValueError("neighbor_ptr must be provided when using neighbor_list format. "
                         "Obtain it from the neighbor list API by setting return_neighbor_list=True.")
    if a1 is None or a2 is None or s8 is None:
        raise ValueError("Functional parameters a1, a2, and s8 must be provided. "
                         "These are functional-dependent parameters required for DFT-D3(BJ) calculations.")
    if compute_virial:
        need = "Virial computation requires periodic boundary conditions. "
        if cell is None:
            raise ValueError(need + "Please provide unit cell parameters (cell) and shifts (neighbor_matrix_shifts or unit_shifts) "
                             "when compute_virial=True or when passing a virial tensor.")
        if use_matrix and neighbor_matrix_shifts is None:
            raise ValueError(need + "Please provide neighbor_matrix_shifts along with cell when using neighbor_matrix format "
                             "and compute_virial=True or passing a virial tensor.")
        if use_list and unit_shifts is None:
            raise ValueError(need + "Please provide unit_shifts along with cell when using neighbor_list format "
                             "and compute_virial=True or passing a virial tensor.")
    # explicit tensors win over d3_params entries (dftd3.py:2727-2757)
    explicit = (covalent_radii, r4r2, c6_reference, coord_num_ref)
    if any(t is None for t in explicit):
        if d3_params is None:
            raise RuntimeError("DFT-D3 parameters must be explicitly provided. Either supply all individual parameters "
                               "(covalent_radii, r4r2, c6_reference, coord_num_ref), provide a D3Parameters instance, "
                               "or provide a d3_params dictionary. See the function docstring for details.")
        if isinstance(d3_params, D3Parameters):
            src = {"rcov": d3_params.rcov, "r4r2": d3_params.r4r2, "c6ab": d3_params.c6ab, "cn_ref": d3_params.cn_ref}
        else:
            src = d3_params
        covalent_radii = src["rcov"] if covalent_radii is None else covalent_radii
        r4r2 = src["r4r2"] if r4r2 is None else r4r2
        c6_reference = src["c6ab"] if c6_reference is None else c6_reference
        coord_num_ref = src["cn_ref"] if coord_num_ref is None else coord_num_ref

    n, dev = positions.size(0), positions.device
    f32 = dict(dtype=torch.float32, device=dev)
    if n == 0:
        nsys = 1 if (batch_idx is None or batch_idx.numel() == 0) else int(batch_idx.max().item()) + 1
        out = (torch.zeros(nsys, **f32), torch.zeros((0, 3), **f32), torch.zeros((0,), **f32))
        return out + (torch.zeros((0, 3, 3), **f32),) if compute_virial else out
    if num_systems is None:
        if batch_idx is None:
            num_systems = 1
        elif cell is not None:
            num_systems = cell.size(0)
        else:
            num_systems = int(batch_idx.max().item()) + 1
    energy = torch.empty(num_systems, **f32)  # zeroed inside mi_d3
    forces = torch.empty((n, 3), **f32)
    coord_num = torch.empty(n, **f32)
    virial = torch.empty((num_systems, 3, 3), **f32) if compute_virial else torch.zeros((0, 3, 3), **f32)
    if C.tracing():
        # torch.compile: the reference's own seam -- one mutating custom op per call (dftd3.py:1792-1796 / :2125-2128, called from
        # :2806-2870); the conversions the eager path does below happen inside the op
        if use_matrix:
            torch.ops.nvalchemiops.dftd3_nm(positions, numbers, neighbor_matrix, covalent_radii, r4r2, c6_reference, coord_num_ref, a1, a2, s8,
                                            energy, forces, coord_num, virial, k1, k3, s6, s5_smoothing_on, s5_smoothing_off, fill_value,
                                            batch_idx, cell, neighbor_matrix_shifts, compute_virial, None)
        else:
            torch.ops.nvalchemiops.dftd3_nl(positions, numbers, neighbor_list[1], neighbor_ptr, covalent_radii, r4r2, c6_reference,
                                            coord_num_ref, a1, a2, s8, energy, forces, coord_num, virial, k1, k3, s6, s5_smoothing_on,
                                            s5_smoothing_off, batch_idx, cell, unit_shifts, compute_virial, None)
        return (energy, forces, coord_num, virial) if compute_virial else (energy, forces, coord_num)
    C.require_device(positions, numbers, neighbor_matrix, neighbor_list, neighbor_ptr, batch_idx)
    scalars = dict(a1=a1, a2=a2, s6=s6, s8=s8, k1=k1, k3=k3, s5_on=s5_smoothing_on, s5_off=s5_smoothing_off)
    tables = (covalent_radii, r4r2, c6_reference, coord_num_ref)
    if use_matrix:
        nm = C.i32(neighbor_matrix)
        fill = n if fill_value is None else fill_value
        packed = None
        if nm is neighbor_matrix and cell is not None and neighbor_matrix_shifts is not None and int(fill) >= n:
            from nvalchemiops.neighborlist import _engine as E

            # valid only while matrix and shifts are provably what the search wrote (tensor identity + version counters); else None
            packed = E.packed_companion(nm, neighbor_matrix_shifts, fill)
            E.learn_dftd3_context(nm, numbers, covalent_radii, k1)  # "auto" policy: the next search into this buffer also sums the CNs
        _launch(positions, numbers, nm, neighbor_matrix_shifts, None, nm.size(1), fill, cell,
                batch_idx, num_systems, tables, scalars, compute_virial, energy, forces, coord_num, virial, packed=packed)
    else:
        idx_j = C.i32(neighbor_list[1])
        _launch(positions, numbers, idx_j, unit_shifts, C.i32(neighbor_ptr), 0, 0, cell, batch_idx, num_systems, tables, scalars,
                compute_virial, energy, forces, coord_num, virial)
    return (energy, forces, coord_num, virial) if compute_virial else (energy, forces, coord_num)

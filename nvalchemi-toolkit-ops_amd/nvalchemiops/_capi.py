"""ctypes binding of libnvalchemiops_hip.so (C ABI declared in include/nvalchemiops_hip.h).

Replaces the reference's Warp launch layer (``wp.from_torch`` + ``wp.launch``: e.g. neighborlist/cell_list.py:980-1034,
autograd.py:300-360).  Zero-copy: raw device pointers of torch tensors and torch's current HIP stream are handed
to the library, which never allocates, frees or retains them.
"""
from __future__ import annotations

import contextvars
import ctypes
import functools
import os

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libnvalchemiops_hip.so")
_LIB = None

MI_F32, MI_F64 = 0, 1
NL_MATRIX, NL_COUNT, NL_CSR = 0, 1, 2
NL_HALF_FILL, NL_NAIVE_EXPR, NL_REUSE_GRID, NL_NO_SHIFTS, NL_NO_PAD = 1, 2, 4, 8, 16
EW_FORCES, EW_CHARGE_GRAD = 1, 2


class NativeLibraryError(RuntimeError):
    """The HIP library is missing or failed: there is deliberately no fallback path."""


class MiD3Params(ctypes.Structure):
    _fields_ = [("rcov", ctypes.c_void_p), ("r4r2", ctypes.c_void_p), ("c6ab", ctypes.c_void_p), ("cn_ref", ctypes.c_void_p),
                ("nz", ctypes.c_int), ("a1", ctypes.c_float), ("a2", ctypes.c_float), ("s6", ctypes.c_float), ("s8", ctypes.c_float),
                ("k1", ctypes.c_float), ("k3", ctypes.c_float), ("s5_on", ctypes.c_float), ("s5_off", ctypes.c_float)]


class MiNlCnRequest(ctypes.Structure):
    """`mi_nl_cn_request` of include/nvalchemiops_hip.h: what a search needs to sum DFT-D3 coordination numbers over the list it writes."""
    _fields_ = [("numbers", ctypes.c_void_p), ("covalent_radii", ctypes.c_void_p), ("nz", ctypes.c_int), ("k1", ctypes.c_float)]


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIB_PATH):
            raise NativeLibraryError(
                f"{_LIB_PATH} not found: build it with `python nvalchemi-toolkit-ops_amd/build_native.py` "
                "(hipcc --offload-arch=gfx950). This package has no CPU or PyTorch fallback.")
        L = ctypes.CDLL(_LIB_PATH)
        L.mi_last_error.restype = ctypes.c_char_p
        L.mi_nl_workspace_bytes.restype = ctypes.c_size_t
        L.mi_nl_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.mi_spline_spread_workspace_bytes.restype = ctypes.c_size_t
        L.mi_pme_solve_scratch_bytes.restype = ctypes.c_size_t
        L.mi_pme_solve_scratch_bytes.argtypes = [ctypes.c_int] * 6
        L.mi_fft_lds_scratch_bytes.restype = ctypes.c_size_t
        L.mi_fft_lds_scratch_bytes.argtypes = [ctypes.c_int] * 5
        L.mi_spline_spread_workspace_bytes.argtypes = [ctypes.c_int] * 5
        L.mi_spline_spread_workspace_bytes_for.restype = ctypes.c_size_t
        L.mi_spline_spread_workspace_bytes_for.argtypes = [ctypes.c_int] * 7
        L.mi_spline_spread_order_offset.restype = ctypes.c_longlong
        L.mi_spline_spread_order_offset.argtypes = [ctypes.c_int] * 6
        L.mi_nl_packed_bytes.restype = ctypes.c_size_t
        L.mi_nl_packed_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        L.mi_nl_cn_bytes.restype = ctypes.c_size_t
        L.mi_nl_cn_bytes.argtypes = [ctypes.c_int]
        if hasattr(L, "mi_d3_workspace_bytes"):
            L.mi_d3_workspace_bytes.restype = ctypes.c_size_t
            L.mi_d3_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
            L.mi_d3_workspace_bytes_packed.restype = ctypes.c_size_t
            L.mi_d3_workspace_bytes_packed.argtypes = [ctypes.c_int] * 4
            L.mi_d3_workspace_bytes_entries.restype = ctypes.c_size_t
            L.mi_d3_workspace_bytes_entries.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong]
        _LIB = L
    return _LIB


def ewald_sym_words() -> int:
    L = lib()
    L.mi_ewald_symmetry_scratch_bytes.restype = ctypes.c_size_t
    return int(L.mi_ewald_symmetry_scratch_bytes()) // 8


def ewald_scratch_bytes(n_atoms: int, dtype: int) -> int:
    L = lib()
    L.mi_ewald_real_scratch_bytes.restype = ctypes.c_size_t
    return int(L.mi_ewald_real_scratch_bytes(int(n_atoms), int(dtype)))


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise NativeLibraryError(f"{what} failed (code {rc}): {lib().mi_last_error().decode()}")


def ptr(t: torch.Tensor | None):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def dtype_code(dtype: torch.dtype) -> int:
    # reference: types.py:20-29 raises ValueError for anything but f16/f32/f64; this path supports f32/f64
    if dtype == torch.float32:
        return MI_F32
    if dtype == torch.float64:
        return MI_F64
    raise ValueError(f"Unsupported dtype: {dtype}")


def require_device(*tensors: torch.Tensor | None) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NativeLibraryError(
                "nvalchemiops (MI355X build) computes on ROCm devices only; got a tensor on "
                f"'{t.device}'. There is no CPU path in this package.")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


SPLINE_REFERENCE_ORDERS = 0x100  # MI_SPLINE_REFERENCE_ORDERS of include/nvalchemiops_hip.h
_REFERENCE_SPLINE_ORDERS = os.environ.get("NVALCHEMIOPS_REFERENCE_SPLINE_ORDERS", "0") not in ("", "0")  # process default
# context-local override (`with nvalchemiops.spline.reference_spline_orders():`): a contextvars.ContextVar, so the setting belongs to the
# thread / asyncio task that entered the block -- another thread launching PME at the same time keeps its own setting, and an exception
# anywhere cannot leave the switch flipped for someone else
_REFERENCE_SPLINE_ORDERS_CTX: contextvars.ContextVar = contextvars.ContextVar("nvalchemiops_reference_spline_orders", default=None)


def reference_spline_orders_active() -> bool:
    v = _REFERENCE_SPLINE_ORDERS_CTX.get()
    return _REFERENCE_SPLINE_ORDERS if v is None else bool(v)


SPLINE_RESOLVED = 0x200  # Python-side only (never reaches the C ABI): the reference-compatibility decision has been taken for this order value


def resolve_spline_order(order: int) -> int:
    """The spline order with the reference-compatibility decision FROZEN into it: low byte = the order, SPLINE_REFERENCE_ORDERS if
    `nvalchemiops.spline.reference_spline_orders` is active where this is called, SPLINE_RESOLVED so that later calls leave it alone.
    Every public electrostatics / spline entry point calls this once, on the caller's thread, and hands the result down; autograd nodes and
    custom-op contexts store it, so a backward pass -- which the autograd engine runs on ITS OWN worker thread for device tensors, where
    the caller's `contextvars` context does not exist -- launches with the same evaluation as its forward."""
    order = int(order)
    if order & SPLINE_RESOLVED or torch.compiler.is_compiling():
        # (while TorchDynamo traces the caller the value stays plain: a context variable cannot be guarded on, and freezing today's setting
        # into a compiled graph would outlive the `with` block; the ops then read the switch at launch time, as before round 6)
        return order
    return (order & 0xff) | SPLINE_RESOLVED | (SPLINE_REFERENCE_ORDERS if reference_spline_orders_active() else 0)


def plain_spline_order(order: int) -> int:
    """The order itself (1 - 6 ...), whatever flags ride on the value."""
    return int(order) & 0xff


def spline_order_arg(order: int) -> int:
    """The `order` argument of the C ABI: the spline order, plus the reference-compatibility bit (orders 5 / 6 evaluated as the reference
    does: zero weights, exponent 4).  A value that went through `resolve_spline_order` keeps the decision taken there; a plain order takes
    the one of the calling context.  The bit travels with every launch: the library itself holds no such state."""
    return resolve_spline_order(order) & (0xff | SPLINE_REFERENCE_ORDERS)


def cdouble(x: float):
    return ctypes.c_double(float(x))


def i32(t: torch.Tensor | None) -> torch.Tensor | None:
    if t is None:
        return None
    return t.to(torch.int32).contiguous()


def _on_device_of(args, kwargs):
    """Device of the first tensor argument when it is not the current one (HIP launches go to the CURRENT device: tensors on another
    GPU of the process need the guard the reference gets from Warp's per-device launch, `wp.launch(device=...)`)."""
    for a in (args if args else kwargs.values()):
        if isinstance(a, torch.Tensor):
            if a.is_cuda and a.device.index != torch.cuda.current_device():
                return a.device
            break
    return None


def eager(fn):
    """Decorator of the public entry points whose result SHAPE depends on device data (COO / CSR conversion, size estimates: one host
    read each).  The launch layer hands raw device pointers and the current HIP stream to ctypes, which TorchDynamo can neither trace nor
    guard, so under ``torch.compile`` these run as eager islands (a graph break around the call) -- the reference's versions of the same
    functions break the graph at their ``.item()`` as well (neighbor_utils.py:426, cell_list.py:700-722)."""
    inner = torch.compiler.disable(fn)

    @functools.wraps(fn)
    def entry(*args, **kwargs):
        # a plain frame around the disabled callable: `torch.compile(f)` unwraps a directly disabled function and would trace its body
        dev = _on_device_of(args, kwargs)
        if dev is not None:
            with torch.cuda.device(dev):
                return inner(*args, **kwargs)
        return inner(*args, **kwargs)

    return entry


def hybrid(fn):
    """Decorator of the public neighbour-list / DFT-D3 entry points.  Outside a trace the body runs as plain eager code and launches
    through ctypes (no custom-op dispatch on a path whose kernels take tens of microseconds).  While TorchDynamo traces the caller
    (`torch.compile`, also ``fullgraph=True`` under Inductor) the SAME body is traced: every launch in it branches on `tracing()` into the
    registered, mutation-annotated ``torch.ops.nvalchemiops.*`` custom op (nvalchemiops/_ops.py), which is what the reference's thin
    wrappers over ``torch.library.custom_op`` do (neighborlist/cell_list.py:725-736, 892-895, 1037-1192; dftd3.py:1792-1796) and what its
    compiled MD-step example relies on (examples/neighborlist/04_neighbors_list_torch_compile_performance.py:323-346)."""
    inner = torch.compiler.disable(fn)

    @functools.wraps(fn)
    def entry(*args, **kwargs):
        if torch.compiler.is_compiling():
            return fn(*args, **kwargs)
        dev = _on_device_of(args, kwargs)
        if dev is not None:
            with torch.cuda.device(dev):
                return inner(*args, **kwargs)
        return inner(*args, **kwargs)

    return entry


def tracing() -> bool:
    """True while TorchDynamo traces the caller (torch.compile): the public electrostatics functions then take their custom-op path."""
    return torch.compiler.is_compiling()


def traceable(fn):
    """Decorator of the public electrostatics entry points.  Unlike `eager` the body is visible to TorchDynamo: it branches on
    `tracing()` into the `alchemiops::*` custom-op composition (nvalchemiops/_eops.py), so `torch.compile(..., fullgraph=True)`
    captures it; outside a trace it only adds the device guard `eager` has."""
    @functools.wraps(fn)
    def entry(*args, **kwargs):
        if torch.compiler.is_compiling():
            return fn(*args, **kwargs)
        dev = _on_device_of(args, kwargs)
        if dev is not None:
            with torch.cuda.device(dev):
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)

    return entry

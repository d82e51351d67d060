"""Shared launcher of the fused HIP neighbour pipeline (csrc/nlist.hip) for every public neighbour-list entry point.

Reference seam replaced: the `nvalchemiops::*cell_list*` / `_naive_*` custom ops (neighborlist/cell_list.py:725-1034,
batch_cell_list.py:739-1067, naive.py:221-397).
"""
from __future__ import annotations

import ctypes
import os
import weakref

import torch

from nvalchemiops import _capi as C

_WS_BYTES: dict[tuple[int, int, int], int] = {}

# ---- packed companion of a padded matrix (round 5) ---------------------------------------------------------------------------------------
# `mi_nl_neighbors_packed` can leave, next to the API-format matrix, the 4 B/slot word list the D3 passes stream (include/nvalchemiops_hip.h).
# The companion rides on the returned matrix tensor as a Python attribute and is only ever USED by `dftd3` when the matrix and the shifts
# are provably the ones it was built with: same tensor objects, same torch version counters (every in-place torch op bumps them, and every
# launch of this package that writes a matrix bumps them by hand, see `_written`).  Anything else -- a clone, a view, an edited entry --
# silently takes the ordinary path; results are bit-identical either way, the companion only removes HBM traffic.
# Policy (NVALCHEMIOPS_NL_PACKED): "0" never, "1" whenever the list qualifies, "auto" (default): only for (device, n_atoms, row width) shapes
# that `dftd3` has been handed before WITHOUT a companion -- so a caller who only wants neighbour lists never pays the extra 4 B/slot
# of writes, and an MD loop gets the fused path from its second step on.
_PACKED_POLICY = os.environ.get("NVALCHEMIOPS_NL_PACKED", "auto")
_PACKED_WANTED: set[tuple[int, int, int]] = set()
_PACKED_ATTR = "_nvalchemiops_packed"
_BUILT_ATTR = "_nvalchemiops_built"
_FULL_ATTR = "_nvalchemiops_full_list"  # FullListRecord: "this matrix is the unmodified output of a full (symmetric) search"
_D3CTX_ATTR = "_nvalchemiops_d3ctx"
_D3CTX_BY_SHAPE: dict = {}  # "auto" policy: (device, n_atoms, row width) -> the species `dftd3` was last run with on a matrix of that shape (<= 64 entries)
# Device-side check of a companion against the arrays it describes, run by `dftd3` on every call (csrc/d3.hip, D3Guard): every
# NVALCHEMIOPS_NL_PACKED_VERIFY-th row is re-derived from matrix + shifts and compared word by word (default 64: +1.6 % of the list's
# traffic; "1" compares every row; "0" switches the check off).  The sampled rows rotate from call to call, so an edit that persists is
# seen within `stride` calls wherever it sits.
_VERIFY_STRIDE = max(0, int(os.environ.get("NVALCHEMIOPS_NL_PACKED_VERIFY", "64") or 0))
_verify_calls = 0


def verify_args() -> tuple[int, int]:
    """(stride, phase) of the sampled companion check for the next `dftd3` call."""
    global _verify_calls
    _verify_calls += 1
    return _VERIFY_STRIDE, (_verify_calls - 1) & 0x3fffffff  # the raw call counter: row offset = phase mod stride, stretch of the row = f(phase div stride)


class D3SearchContext:
    """What a neighbour search needs in order to sum the DFT-D3 coordination numbers of the list while it writes it (round 6,
    `mi_nl_neighbors_packed_cn`): atomic numbers, covalent radii (fp32 table indexed by Z, `D3Parameters.rcov`) and the counting function's
    steepness k1.  Attached to the OUTPUT buffer (`attach_dftd3_context`, or `tuned_neighbor_buffers(for_dftd3=ctx)`), so the reference
    signatures of `cell_list` / `batch_cell_list` stay what they are.  The tensors are converted once, here; a caller who later hands
    `dftd3` other species, radii, positions or cell simply gets the ordinary CN pass: the numbers are only adopted when the fingerprint
    `dftd3` computes on the device from ITS arguments equals the one the search stored (csrc/common.h)."""
    __slots__ = ("numbers", "rcov", "k1", "_src")

    def __init__(self, numbers: torch.Tensor, covalent_radii: torch.Tensor, k1: float = 16.0):
        self.numbers = numbers.detach().to(torch.int32).contiguous()
        self.rcov = covalent_radii.detach().to(device=numbers.device, dtype=torch.float32).contiguous()
        self.k1 = float(k1)
        # what the converted copies were made from (identity + torch version): lets `dftd3` tell cheaply whether the context it finds on a
        # buffer still describes the species it is being called with (a host-side hint only -- adoption is decided on the device)
        self._src = (weakref.ref(numbers), numbers._version, weakref.ref(covalent_radii), covalent_radii._version)

    def describes(self, numbers: torch.Tensor, covalent_radii: torch.Tensor, k1: float) -> bool:
        n_ref, n_ver, r_ref, r_ver = self._src
        return (n_ref() is numbers and numbers._version == n_ver and r_ref() is covalent_radii and covalent_radii._version == r_ver
                and float(k1) == self.k1)


def attach_dftd3_context(neighbor_matrix: torch.Tensor, numbers: torch.Tensor, covalent_radii: torch.Tensor, k1: float = 16.0) -> None:
    """Every later search into `neighbor_matrix` that qualifies for a packed companion also sums the DFT-D3 coordination numbers."""
    setattr(neighbor_matrix, _D3CTX_ATTR, D3SearchContext(numbers, covalent_radii, k1))
    want_packed_companion(neighbor_matrix.device, int(neighbor_matrix.shape[0]), int(neighbor_matrix.shape[1]))


def learn_dftd3_context(neighbor_matrix: torch.Tensor, numbers: torch.Tensor, covalent_radii: torch.Tensor, k1: float) -> None:
    """Policy "auto", the coordination numbers' half: `dftd3` was handed a padded matrix this package built into the caller's buffer (it carries a
    companion, or the "built here" mark).  An MD loop searches into the same buffers every step, so the species it runs D3 with are attached
    to the buffer and the NEXT search into it also sums the coordination numbers -- the unmodified reference call sequence gets the fused
    path from its second or third step on.  A caller whose species differ from call to call just has the context replaced each time (and
    `dftd3` falls back to its own pass whenever the device-side fingerprints disagree)."""
    if _PACKED_POLICY != "auto" or not neighbor_matrix.is_cuda:
        return
    if getattr(neighbor_matrix, _PACKED_ATTR, None) is None and getattr(neighbor_matrix, _BUILT_ATTR, None) is None:
        return
    ctx = getattr(neighbor_matrix, _D3CTX_ATTR, None)
    try:
        if ctx is not None and ctx.describes(numbers, covalent_radii, k1):
            return
        if numbers.shape[0] != neighbor_matrix.shape[0]:
            return
        key = (neighbor_matrix.device.index, int(neighbor_matrix.shape[0]), int(neighbor_matrix.shape[1]))
        known = _D3CTX_BY_SHAPE.get(key)
        ctx = known if (known is not None and known.describes(numbers, covalent_radii, k1)) else D3SearchContext(numbers, covalent_radii, k1)
        setattr(neighbor_matrix, _D3CTX_ATTR, ctx)
        # a caller who lets every search allocate its outputs never searches into the same tensor twice: the shape remembers the context
        if len(_D3CTX_BY_SHAPE) > 64:
            _D3CTX_BY_SHAPE.clear()
        _D3CTX_BY_SHAPE[key] = ctx
    except Exception:
        pass


def invalidate(*tensors: torch.Tensor) -> None:
    """Drop the packed companion (and the coordination numbers riding with it) of a neighbour matrix.  Call this after writing
    `neighbor_matrix` / `neighbor_matrix_shifts` through anything torch's version counters do not see -- a raw-pointer kernel, DLPack,
    `tensor.data[...] = ...`, `untyped_storage().copy_` -- and pass either tensor (or both): `dftd3` then reads the arrays themselves."""
    for t in tensors:
        if t is None:
            continue
        for attr in (_PACKED_ATTR, _BUILT_ATTR, _FULL_ATTR):
            if hasattr(t, attr):
                delattr(t, attr)
        owner = getattr(t, "_nvalchemiops_owner", None)  # the shifts tensor knows its matrix
        nm = owner() if owner is not None else None
        if nm is not None and nm is not t:
            for attr in (_PACKED_ATTR, _BUILT_ATTR, _FULL_ATTR):
                if hasattr(nm, attr):
                    delattr(nm, attr)


class PackedCompanion:
    """What `dftd3` needs to trust a companion: the words, and the identity + version of the two tensors they describe.  `cn`: the block of
    coordination numbers the same search summed (or None); its validity is checked on the device, not here."""
    __slots__ = ("words", "cn", "nm_version", "nsh_ref", "nsh_version", "n_atoms", "row_width", "fill_value")

    def __init__(self, words, nm, nsh, fill_value, cn=None):
        self.words = words
        self.cn = cn
        self.nm_version = nm._version
        self.nsh_ref = weakref.ref(nsh)
        self.nsh_version = nsh._version
        self.n_atoms, self.row_width = int(nm.shape[0]), int(nm.shape[1])
        self.fill_value = int(fill_value)

    def matches(self, nm: torch.Tensor, nsh: torch.Tensor | None, fill_value: int) -> bool:
        return (nsh is not None and self.nsh_ref() is nsh and nm._version == self.nm_version and nsh._version == self.nsh_version
                and tuple(nm.shape) == (self.n_atoms, self.row_width) and int(fill_value) == self.fill_value
                and nm.dtype == torch.int32 and nsh.dtype == torch.int32 and nm.is_contiguous() and nsh.is_contiguous())


class FullListRecord:
    """What lets `ewald_real_space` skip its per-entry symmetry checksums (round 6): the matrix is the output of a FULL (not half-filled),
    padded search of this package -- symmetric by construction unless a row overflowed, which the search's own `num_neighbors` shows -- and
    matrix, shifts and counts still carry the version counters that search left.  An edit through torch bumps a counter and the record is
    dead; edits behind torch's back are the caller's to announce (`invalidate`) -- the consumer also samples: `mi_ewald_real_listed`."""
    __slots__ = ("nm_version", "nsh_ref", "nsh_version", "num_ref", "num_version", "n_atoms", "row_width", "fill_value")

    def __init__(self, nm, nsh, num, fill_value):
        self.nm_version = nm._version
        self.nsh_ref, self.nsh_version = weakref.ref(nsh), nsh._version
        self.num_ref, self.num_version = weakref.ref(num), num._version
        self.n_atoms, self.row_width = int(nm.shape[0]), int(nm.shape[1])
        self.fill_value = int(fill_value)

    def counts(self, nm: torch.Tensor, nsh: torch.Tensor | None):
        """The search's num_neighbors tensor if (nm, nsh) are still what it wrote, else None."""
        num = self.num_ref()
        ok = (nsh is not None and num is not None and self.nsh_ref() is nsh and nm._version == self.nm_version and nsh._version == self.nsh_version
              and num._version == self.num_version and tuple(nm.shape) == (self.n_atoms, self.row_width) and nm.dtype == torch.int32
              and nsh.dtype == torch.int32 and num.dtype == torch.int32 and nm.is_contiguous() and nsh.is_contiguous() and num.is_contiguous()
              and num.device == nm.device)
        return num if ok else None


def full_list_counts(nm: torch.Tensor | None, nsh: torch.Tensor | None):
    """`num_neighbors` of the full search that wrote (nm, nsh), while that is provably still their content; else None."""
    rec = getattr(nm, _FULL_ATTR, None) if nm is not None else None
    if rec is None:
        return None
    try:
        return rec.counts(nm, nsh)
    except Exception:
        return None


def _record_full_list(nm, nsh, num, fill_value, qualifies: bool) -> None:
    if not qualifies or num is None:
        return
    try:
        setattr(nm, _FULL_ATTR, FullListRecord(nm, nsh, num, fill_value))
        setattr(nsh, "_nvalchemiops_owner", weakref.ref(nm))
    except Exception:  # no version counter (inference tensor): no record
        pass


def want_packed_companion(device: torch.device, n_atoms: int, row_width: int) -> None:
    """Tell the "auto" policy up front that matrices of this shape feed `dftd3` (what it otherwise learns from the first dftd3 call)."""
    if len(_PACKED_WANTED) > 64:
        _PACKED_WANTED.clear()
    _PACKED_WANTED.add((device.index, int(n_atoms), int(row_width)))


def _written(*tensors) -> None:
    """A launch of this package wrote these tensors through raw pointers: bump their version counters as an in-place torch op would, and
    drop whatever companion described the old contents."""
    for t in tensors:
        if t is None:
            continue
        for attr in (_PACKED_ATTR, _BUILT_ATTR, _FULL_ATTR):
            if hasattr(t, attr):
                delattr(t, attr)
        try:
            torch.autograd.graph.increment_version(t)
        except Exception:  # inference tensors carry no version counter; they can carry no companion either (PackedCompanion reads it)
            pass


def packed_companion(nm: torch.Tensor, nsh: torch.Tensor | None, fill_value: int):
    """The companion record of (nm, nsh) if it is still valid (`.words`, `.cn`), else None.  Also where the "auto" policy learns: a matrix
    this package built without a companion, handed to a consumer that would have used one, registers its shape."""
    rec = getattr(nm, _PACKED_ATTR, None)
    if rec is not None:
        try:
            if rec.matches(nm, nsh, fill_value):
                return rec
        except Exception:
            pass
        return None
    if _PACKED_POLICY == "auto" and nsh is not None and getattr(nm, _BUILT_ATTR, None) == nm._version and nm.is_cuda:
        if len(_PACKED_WANTED) > 64:
            _PACKED_WANTED.clear()
        _PACKED_WANTED.add((nm.device.index, int(nm.shape[0]), int(nm.shape[1])))
    return None


def workspace(n_atoms: int, n_systems: int, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    key = (n_atoms, n_systems, C.dtype_code(dtype))
    nbytes = _WS_BYTES.get(key)
    if nbytes is None:
        nbytes = int(C.lib().mi_nl_workspace_bytes(*key))
        _WS_BYTES[key] = nbytes
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def canon_positions(positions: torch.Tensor) -> torch.Tensor:
    """Detached, contiguous positions in a dtype the kernels take.  The reference also instantiates its neighbour kernels for
    float16 (naive.py:186-188, types.py:26); here half-precision coordinates are read exactly and the search itself runs in
    float32, so a pair within half-precision rounding of the cutoff may be classified differently than by fp16 arithmetic."""
    pos = positions.detach()
    if pos.dtype in (torch.float16, torch.bfloat16):
        pos = pos.float()
    return pos.contiguous()


def canon_geometry(positions: torch.Tensor, cell: torch.Tensor, pbc: torch.Tensor, n_systems: int | None = None):
    """positions contiguous; cell -> (B,3,3) in the positions dtype; pbc -> (B,3) bool (1 byte per flag)."""
    pos = canon_positions(positions)
    cell = cell.detach()
    cell = (cell if cell.ndim == 3 else cell.unsqueeze(0)).to(dtype=pos.dtype, device=pos.device).contiguous()
    pbc = pbc.reshape(-1, 3).to(device=pos.device, dtype=torch.bool)
    if pbc.shape[0] != cell.shape[0]:
        pbc = pbc.expand(cell.shape[0], 3)
    return pos, cell, pbc.contiguous()


def run(pos, cell, pbc, batch_idx, cutoff, mode, flags, *, nm=None, nsh=None, num=None, max_neighbors=0, fill_value=0,
        nptr=None, list_ij=None, list_sh=None, n_pairs=0, ws=None, origin=None):
    n = pos.shape[0]
    nsys = cell.shape[0]
    if ws is None:
        ws = workspace(n, nsys, pos.dtype, pos.device)
    if mode == C.NL_MATRIX:
        _written(nm, nsh, num)
    rc = C.lib().mi_nl_neighbors(
        C.ptr(pos), n, C.ptr(cell), C.ptr(pbc), C.ptr(batch_idx), nsys, C.cdouble(cutoff), C.dtype_code(pos.dtype), mode, flags,
        C.ptr(nm), C.ptr(nsh), C.ptr(num), int(max_neighbors), int(fill_value), C.ptr(nptr), C.ptr(list_ij), C.ptr(list_sh),
        ctypes.c_longlong(int(n_pairs)), C.ptr(origin), C.ptr(ws), ctypes.c_size_t(ws.numel()), C.stream_of(pos))
    C.check(rc, "mi_nl_neighbors")
    return ws


def neighbor_matrix(pos, cell, pbc, batch_idx, cutoff, max_neighbors, fill_value, half_fill, nm, nsh, num, *, naive=False,
                    want_shifts=True, pad=True, origin=None):
    flags = (C.NL_HALF_FILL if half_fill else 0) | (C.NL_NAIVE_EXPR if naive else 0)
    if not want_shifts:
        flags |= C.NL_NO_SHIFTS
    if not pad:
        flags |= C.NL_NO_PAD
    if C.tracing():  # torch.compile: the mutation-annotated custom op instead of the ctypes launch (nvalchemiops/_ops.py)
        torch.ops.nvalchemiops.neighbor_search(pos, cell, pbc, batch_idx, cutoff, flags, fill_value, nm, nsh if want_shifts else None, num,
                                               origin)
        return
    n = pos.shape[0]
    qualifies = (want_shifts and pad and not half_fill and nsh is not None and nm.is_contiguous() and nsh.is_contiguous() and int(fill_value) >= n
                 and int(max_neighbors) > 0)
    emit = qualifies and (_PACKED_POLICY == "1" or (_PACKED_POLICY == "auto" and (pos.device.index, n, int(max_neighbors)) in _PACKED_WANTED))
    nbytes = int(C.lib().mi_nl_packed_bytes(n, int(max_neighbors))) if emit else 0
    if nbytes == 0:
        run(pos, cell, pbc, batch_idx, cutoff, C.NL_MATRIX, flags, nm=nm, nsh=nsh if want_shifts else None, num=num,
            max_neighbors=max_neighbors, fill_value=fill_value, origin=origin)
        _record_full_list(nm, nsh, num, fill_value, qualifies)
        if qualifies and _PACKED_POLICY == "auto":
            try:
                setattr(nm, _BUILT_ATTR, nm._version)  # "built here, full periodic-capable matrix, no companion": what `packed_companion` learns from
            except Exception:
                pass
        return
    old = getattr(nm, _PACKED_ATTR, None)  # an MD loop searches into the same buffers every step: reuse the companion's storage
    words = old.words if (old is not None and old.words.numel() == nbytes and old.words.device == pos.device) else torch.empty(nbytes, dtype=torch.uint8, device=pos.device)
    ws = workspace(n, cell.shape[0], pos.dtype, pos.device)
    # the caller announced that this list feeds dftd3 with these species (or, policy "auto", dftd3 was last run with them on a matrix of this
    # shape): also sum the coordination numbers
    ctx = getattr(nm, _D3CTX_ATTR, None)
    if ctx is None and _PACKED_POLICY == "auto":
        ctx = _D3CTX_BY_SHAPE.get((pos.device.index, n, int(max_neighbors)))
    if ctx is not None and (ctx.numbers.shape[0] != n or ctx.numbers.device != pos.device):
        ctx = None
    _written(nm, nsh, num)
    cn = None
    if ctx is not None:
        cn_bytes = int(C.lib().mi_nl_cn_bytes(n))
        cn = old.cn if (old is not None and old.cn is not None and old.cn.numel() == cn_bytes and old.cn.device == pos.device) else torch.empty(cn_bytes, dtype=torch.uint8, device=pos.device)
        req = C.MiNlCnRequest(numbers=ctx.numbers.data_ptr(), covalent_radii=ctx.rcov.data_ptr(), nz=int(ctx.rcov.shape[0]), k1=ctx.k1)
        rc = C.lib().mi_nl_neighbors_packed_cn(
            C.ptr(pos), n, C.ptr(cell), C.ptr(pbc), C.ptr(batch_idx), cell.shape[0], C.cdouble(cutoff), C.dtype_code(pos.dtype), flags,
            C.ptr(nm), C.ptr(nsh), C.ptr(num), int(max_neighbors), int(fill_value), C.ptr(origin), C.ptr(ws), ctypes.c_size_t(ws.numel()),
            C.ptr(words), ctypes.c_size_t(nbytes), ctypes.byref(req), C.ptr(cn), ctypes.c_size_t(cn_bytes), C.stream_of(pos))
        C.check(rc, "mi_nl_neighbors_packed_cn")
    else:
        rc = C.lib().mi_nl_neighbors_packed(
            C.ptr(pos), n, C.ptr(cell), C.ptr(pbc), C.ptr(batch_idx), cell.shape[0], C.cdouble(cutoff), C.dtype_code(pos.dtype), flags,
            C.ptr(nm), C.ptr(nsh), C.ptr(num), int(max_neighbors), int(fill_value), C.ptr(origin), C.ptr(ws), ctypes.c_size_t(ws.numel()),
            C.ptr(words), ctypes.c_size_t(nbytes), C.stream_of(pos))
        C.check(rc, "mi_nl_neighbors_packed")
    _record_full_list(nm, nsh, num, fill_value, qualifies)
    try:
        setattr(nm, _PACKED_ATTR, PackedCompanion(words, nm, nsh, fill_value, cn=cn))
        setattr(nsh, "_nvalchemiops_owner", weakref.ref(nm))  # so that `invalidate(shifts)` finds the matrix the companion rides on
    except Exception:  # no version counter (inference tensor): no companion
        pass


def neighbor_matrix_dual(pos, cell, pbc, batch_idx, cutoff_short, cutoff_long, fill_value, half_fill, short, long_, *, naive=True, want_shifts=True,
                         origin=None):
    """ONE sweep over the candidate pairs fills both padded matrices (`mi_nl_neighbors_dual`): `short` / `long_` = (nm, nsh | None, num)."""
    flags = (C.NL_HALF_FILL if half_fill else 0) | (C.NL_NAIVE_EXPR if naive else 0)
    if not want_shifts:
        flags |= C.NL_NO_SHIFTS
    n, nsys = pos.shape[0], cell.shape[0]
    (nm1, nsh1, num1), (nm2, nsh2, num2) = short, long_
    if C.tracing():
        torch.ops.nvalchemiops.neighbor_search_dual(pos, cell, pbc, batch_idx, cutoff_short, cutoff_long, flags, fill_value, nm1,
                                                    nsh1 if want_shifts else None, num1, nm2, nsh2 if want_shifts else None, num2, origin)
        return
    ws = workspace(n, nsys, pos.dtype, pos.device)
    _written(nm1, nsh1, num1, nm2, nsh2, num2)
    rc = C.lib().mi_nl_neighbors_dual(
        C.ptr(pos), n, C.ptr(cell), C.ptr(pbc), C.ptr(batch_idx), nsys, C.cdouble(cutoff_short), C.cdouble(cutoff_long), C.dtype_code(pos.dtype), flags,
        C.ptr(nm1), C.ptr(nsh1 if want_shifts else None), C.ptr(num1), int(nm1.shape[1]),
        C.ptr(nm2), C.ptr(nsh2 if want_shifts else None), C.ptr(num2), int(nm2.shape[1]),
        int(fill_value), C.ptr(origin), C.ptr(ws), ctypes.c_size_t(ws.numel()), C.stream_of(pos))
    C.check(rc, "mi_nl_neighbors_dual")


@torch.compiler.disable
def neighbor_csr(pos, cell, pbc, batch_idx, cutoff, half_fill, *, naive=False, want_shifts=True, max_neighbors=None, origin=None):
    """Direct COO/CSR emission: count pass -> prefix sum -> fill pass (the padded matrix is never materialised).

    Returns (neighbor_list[2,P], neighbor_ptr[N+1], shifts[P,3] | None, max_count).  The list length is read back from the device
    (one host sync), so under `torch.compile` this is an eager island: the reference's COO conversion breaks the graph at its
    `.item()` too (neighbor_utils.py:426)."""
    n = pos.shape[0]
    dev = pos.device
    flags = (C.NL_HALF_FILL if half_fill else 0) | (C.NL_NAIVE_EXPR if naive else 0)
    num = torch.empty(n, dtype=torch.int32, device=dev)
    ws = run(pos, cell, pbc, batch_idx, cutoff, C.NL_COUNT, flags, num=num, origin=origin)
    nptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    torch.cumsum(num, dim=0, out=nptr[1:])
    total, max_count = (int(v) for v in torch.stack([nptr[-1], num.max()]).tolist())  # the one host sync
    if max_neighbors is not None and max_count > max_neighbors:
        from nvalchemiops.neighborlist.neighbor_utils import NeighborOverflowError

        raise NeighborOverflowError(max_neighbors, max_count)
    lst = torch.empty((2, total), dtype=torch.int32, device=dev)
    sh = torch.empty((total, 3), dtype=torch.int32, device=dev) if want_shifts else None
    if total > 0:
        run(pos, cell, pbc, batch_idx, cutoff, C.NL_CSR, flags | C.NL_REUSE_GRID, nptr=nptr, list_ij=lst, list_sh=sh, n_pairs=total, ws=ws)
    return lst, nptr, sh, max_count

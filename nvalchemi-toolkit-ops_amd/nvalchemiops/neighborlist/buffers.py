"""Choosing the buffers a large padded neighbour matrix is written into -- by measurement (round 5; no reference counterpart).

Why this exists: on MI355X the time of a multi-GB padded-matrix search depends on WHERE the driver placed the output buffers.  For the
headline list (100 000 atoms x 2560 slots: 1.0 GB of indices, 3.1 GB of shifts, 1.0 GB of packed companion) the same kernel takes
1.04 - 1.13 ms into some buffer sets and 1.43 - 1.45 ms into others -- fixed for the life of a buffer set, both kinds present in one
process, not a function of virtual addresses or offsets (profiles/r04_probe_nl_alloc_shop.log,
profiles/r05_probe_nl_buffer_shop.log).  User code cannot see physical placement, but it can MEASURE it: an MD code allocates its
neighbour buffers once and searches into them every step, so `tuned_neighbor_buffers` allocates a few candidate sets, times a trial
search into each (HIP events, the caller's own system) and keeps the fastest.  Set-up cost: `candidates` x 3 searches; transient
memory: `candidates` x the buffer size.  Results do not depend on the choice (the buffers are plain `torch.empty` tensors).
"""
from __future__ import annotations

import statistics

import torch

from nvalchemiops.neighborlist import _engine as E


@torch.compiler.disable
def tuned_neighbor_buffers(positions: torch.Tensor, cutoff: float, cell: torch.Tensor, pbc: torch.Tensor, max_neighbors: int, *,
                           batch_idx: torch.Tensor | None = None, candidates: int = 6, trials: int = 3,
                           for_dftd3: "bool | E.D3SearchContext" = False, release_unused: bool = True, report: dict | None = None):
    """Allocate ``(neighbor_matrix[N, M], neighbor_matrix_shifts[N, M, 3], num_neighbors[N])`` for repeated
    ``cell_list(..., neighbor_matrix=..., neighbor_matrix_shifts=..., num_neighbors=...)`` / ``batch_cell_list`` calls, picking among
    `candidates` freshly allocated buffer sets the one a trial search of THIS system fills fastest.

    for_dftd3: the buffers will feed `dftd3` -- the trial searches (and every later search into these buffers) also emit the packed
    companion the D3 passes stream (`neighborlist/_engine.py`), so its placement is part of what is measured.  Pass a
    `D3SearchContext(numbers, covalent_radii, k1)` instead of True and the searches also sum the DFT-D3 coordination numbers of the list
    (the context is attached to the returned matrix: `dftd3` then skips its CN pass whenever it is called with the same atoms).
    release_unused: hand the losing candidates back to the driver (`torch.cuda.empty_cache()`), not just to torch's caching allocator.
    report: optional dict that receives the per-candidate trial times (ms) and the index chosen.
    Small lists (< 256 MiB of output) gain nothing: one set is allocated and returned."""
    from nvalchemiops.neighborlist.batch_cell_list import batch_cell_list
    from nvalchemiops.neighborlist.cell_list import cell_list

    n, m, dev = positions.shape[0], int(max_neighbors), positions.device

    ctx = for_dftd3 if isinstance(for_dftd3, E.D3SearchContext) else None

    def fresh():
        bufs = (torch.empty((n, m), dtype=torch.int32, device=dev), torch.empty((n, m, 3), dtype=torch.int32, device=dev),
                torch.empty((n,), dtype=torch.int32, device=dev))
        if ctx is not None:
            setattr(bufs[0], E._D3CTX_ATTR, ctx)
        return bufs

    def search(bufs):
        if batch_idx is None:
            cell_list(positions, cutoff, cell, pbc, neighbor_matrix=bufs[0], neighbor_matrix_shifts=bufs[1], num_neighbors=bufs[2])
        else:
            batch_cell_list(positions, cutoff, cell, pbc, batch_idx, neighbor_matrix=bufs[0], neighbor_matrix_shifts=bufs[1], num_neighbors=bufs[2])

    if for_dftd3 and E._PACKED_POLICY != "0":
        E.want_packed_companion(dev, n, m)
    if not positions.is_cuda or 16 * n * m < (256 << 20) or candidates <= 1:
        bufs = fresh()
        if report is not None:
            report.update(candidates=1, chosen=0, trial_ms=[])
        return bufs
    # never more candidates than half of the free device memory holds (a set is 16 B/slot + 4 B/slot of companion)
    try:
        free_bytes, _ = torch.cuda.mem_get_info(dev)
        candidates = max(1, min(int(candidates), int(0.5 * free_bytes // (20 * n * m + 1))))
    except Exception:
        pass
    sets, times = [], []
    for _ in range(int(candidates)):  # all candidates stay alive until the choice is made: each one is distinct memory
        bufs = fresh()
        search(bufs)  # first touch + the companion's allocation
        search(bufs)  # (and one more untimed pass: the very first search of a process also pays for code load and allocator warm-up)
        ms = []
        for _ in range(int(trials)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            search(bufs)
            b.record()
            b.synchronize()
            ms.append(a.elapsed_time(b))
        sets.append(bufs)
        times.append(statistics.median(ms))
    best = min(range(len(sets)), key=times.__getitem__)
    chosen = sets[best]
    if report is not None:
        report.update(candidates=len(sets), chosen=best, trial_ms=[round(t, 4) for t in times])
    del sets, bufs
    if release_unused:
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()
    return chosen


__all__ = ["tuned_neighbor_buffers"]

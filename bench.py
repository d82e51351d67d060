#!/usr/bin/env python
"""Headline benchmark: atom-steps/s of one full hot-path step (neighbor lists + DFT-D3(BJ) + particle-mesh Ewald)
on a 100k-atom periodic box, one process per MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" (BASELINE.json metric; BASELINE.md headline row) on synthetic data already resident in HBM:
  1. neighbor_list  rc = 9 A, padded matrix (M = 256), fp64 positions                  -> real-space PME list
  2. particle_mesh_ewald  alpha = 0.35 /A, mesh 128^3, B-spline order 5, E + F, fp64   (real + reciprocal)
  3. neighbor_list  rc = 40 Bohr (21.2 A), fp32 positions -> D3 list (~2.4k pairs/atom): padded matrix with an explicit row
     width M = 2560 (--d3-format matrix, default: the format and the explicit `max_neighbors` the reference's own D3 benchmark
     uses) or exact-size COO/CSR built in two passes (--d3-format csr)
  4. dftd3(BJ)  a1=0.4289 a2=4.4407 s8=0.7875, E + F + virial, fp32
Steps 1-2 and 3-4 are independent and are enqueued on two HIP streams (--overlap 1, default); --overlap 0 serialises them and
reports per-stage times.
System: jittered FCC box (a = 4 A, sigma = 0.05 A, seed 1234 + rank), first N sites, +-1 charges, Z in {6, 8}; D3 tables
are the reference test-suite's analytic tables extended to Z <= 94 (real Grimme tables are not in the reference repo).
A single box does not shard: for --gpus N > 1 every rank runs its own replica box (weak scaling) and the
per-system energies are exchanged with ONE RCCL all_gather per step.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches this script under `torch.distributed.run` with N ranks
(rank r on GPU r, RCCL); on a box with fewer than N GPUs the ranks share device 0 over gloo (a smoke test of the N > 1 code path,
labelled as such in `config.parallelism`).

Prints ONE JSON line (rank 0) with the driver's fields plus
  `stats`        per-step GPU time from HIP events: median / min / max (the reference protocol reports the median, benchmarks/utils.py:170-240);
                 `value` itself is the contract's wall-clock figure over exactly K steps
  `kernels`      every timed kernel: average in the timed (two-stream) region, isolated median from an untimed serial pass, SURVEY 8(d)
                 algorithmic bytes, the bytes this design moves, what bounds it, fraction of that bound
  `roofline`     the entry of the dominant kernel (largest isolated time)
  `cpu_baseline` the CPU oracle on the host: 1 thread on a bounded sample box (same density / cutoffs / spline order), and the
                 all-core OpenMP build on the FULL 100k-atom step (`--cpu-full-size` adds the 1-thread full-size leg, ~1 min)
Other workloads: `--workload c2|c3|c4` (BASELINE.json's single-GPU configurations, same JSON shape: `config.workload`, `roofline` of that
configuration's dominant kernel, `cpu_baseline`), `--workload c5` (BASELINE config 5 shard) and `--workload ref-nlist|ref-d3|ref-pme` (the reference's own published
benchmark configurations, BASELINE.md, with its warm-up / median protocol).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]

BOHR = 1.8897261246
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# VALU issue peak: 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles per SIMD at 2.4 GHz (MI355X_MICROARCH.md, wave scheduling)
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 2.0
# VALU wave-instructions per launch of the VALU-bound kernels on the headline box, from the committed SQ_INSTS_VALU pass
# (profiles/: counters cannot be read from inside the timed run)
VALU_PROFILE = next((p for p in (os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_valu.json", "r05_pmc_valu.json", "r04_pmc_valu.json", "r03_pmc_valu.json", "r02_pmc_valu.json")) if os.path.exists(p)),
                    os.path.join(ROOT, "profiles", "r04_pmc_valu.json"))
PME = dict(cutoff=9.0, alpha=0.35, mesh=(128, 128, 128), order=5, max_neighbors=256)
D3 = dict(cutoff=40.0, a1=0.4289, a2=4.4407, s8=0.7875, max_neighbors=2560)  # row width: explicit, as in the reference's own
# D3 benchmark (benchmarks/interactions/dispersion/benchmark_dftd3.py:325-347 + its yaml `max_neighbors`); the fullest row of the headline box has 2497 entries


def build_system(n_atoms: int, seed: int, device):
    from tests import systems as S  # input generators only; the oracle is imported by cpu_baseline() alone

    pos, cell, q, numbers = S.fcc_box(n_atoms, seed=seed, dtype=np.float64)
    if os.environ.get("BENCH_SHUFFLE_ATOMS") == "1":  # tuning aid, not the reported mode: the same box with its atoms in random index order
        perm = np.random.default_rng(99).permutation(n_atoms)
        pos, q, numbers = pos[perm], q[perm], numbers[perm]
    tables = S.d3_test_tables(94, seed=7)
    t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=device)  # noqa: E731
    sysd = dict(
        n=n_atoms, pos64=t(pos), cell64=t(cell), q64=t(q), numbers=t(numbers),
        pos32b=t((pos * BOHR).astype(np.float32)), cell32b=t((cell * BOHR).astype(np.float32)),
        pbc=torch.tensor([True, True, True], device=device),
        host=dict(pos=pos, cell=cell, q=q, numbers=numbers, tables=tables),
    )
    return sysd, tables


def build_batch(n_systems: int, atoms_each: int, seed: int, device, first_system: int = 0):
    """BASELINE config 5 shard: `n_systems` independent periodic boxes of `atoms_each` atoms (a = 4 A FCC, L = 32 A for 2000), systems
    [first_system, first_system + n_systems) of the global batch (the jitter seed is a function of the GLOBAL system id, so the shards of
    an N-rank run are the slices of the single-rank batch)."""
    from tests import systems as S

    parts = [S.fcc_box(atoms_each, seed=seed + 17 * (first_system + b), dtype=np.float64) for b in range(n_systems)]
    pos = np.concatenate([p[0] for p in parts])
    cell = np.stack([p[1] for p in parts])
    q = np.concatenate([p[2] for p in parts])
    numbers = np.concatenate([p[3] for p in parts])
    bi = np.repeat(np.arange(n_systems, dtype=np.int32), atoms_each)
    tables = S.d3_test_tables(94, seed=7)
    t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=device)  # noqa: E731
    sysd = dict(n=len(pos), nsys=n_systems, pos64=t(pos), cell64=t(cell), q64=t(q), numbers=t(numbers), bi=t(bi),
                pos32b=t((pos * BOHR).astype(np.float32)), cell32b=t((cell * BOHR).astype(np.float32)),
                pbc=torch.ones((n_systems, 3), dtype=torch.bool, device=device))
    return sysd, tables


def make_batch_step(sysd, tables, device, world, sizes):
    """Config-5 step on this rank's shard: batch nlist + batch PME (mesh 32^3 per system) + batch nlist + batch D3, then ONE
    all_gather of the per-system energies (D3 energy[B_local] and the per-system sum of the per-atom PME energies)."""
    from nvalchemiops.distributed import all_gather_system_values, segment_energy
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
    from nvalchemiops.neighborlist import batch_cell_list

    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    params = D3Parameters(rcov=t(tables["rcov"]), r4r2=t(tables["r4r2"]), c6ab=t(tables["c6ab"]), cn_ref=t(tables["cn_ref"]))
    n, nsys, m = sysd["n"], sysd["nsys"], PME["max_neighbors"]
    nm = torch.empty((n, m), dtype=torch.int32, device=device)
    nsh = torch.empty((n, m, 3), dtype=torch.int32, device=device)
    num = torch.empty(n, dtype=torch.int32, device=device)
    alpha = torch.full((nsys,), PME["alpha"], dtype=torch.float64, device=device)
    d3_bufs = None
    if D3_FORMAT == "matrix":  # row width checked against this shard outside the timed region (see make_step)
        md = D3["max_neighbors"]
        trial = batch_cell_list(sysd["pos32b"], D3["cutoff"], sysd["cell32b"], sysd["pbc"], sysd["bi"], max_neighbors=64)[1]
        need = int(trial.max().item())
        if need > md:
            md = D3["max_neighbors"] = (need + 63) // 64 * 64
        del trial
        d3_bufs = list_buffers(sysd["pos32b"], D3["cutoff"], sysd["cell32b"], sysd["pbc"], md, device, batch_idx=sysd["bi"],
                               for_dftd3=d3_search_context(sysd["numbers"], params), report=BUFFER_REPORT)

    def step(record=None):
        ev = []

        def mark(name):
            if record is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((name, e))

        mark("start")
        batch_cell_list(sysd["pos64"], PME["cutoff"], sysd["cell64"], sysd["pbc"], sysd["bi"], neighbor_matrix=nm,
                        neighbor_matrix_shifts=nsh, num_neighbors=num)
        mark("nlist_pme")
        e_pme, f_pme = particle_mesh_ewald(sysd["pos64"], sysd["q64"], sysd["cell64"], alpha=alpha, mesh_dimensions=(32, 32, 32),
                                           spline_order=PME["order"], batch_idx=sysd["bi"], neighbor_matrix=nm, neighbor_matrix_shifts=nsh,
                                           compute_forces=True)
        mark("pme")
        common = dict(a1=D3["a1"], a2=D3["a2"], s8=D3["s8"], d3_params=params, cell=sysd["cell32b"], batch_idx=sysd["bi"],
                      compute_virial=True, num_systems=nsys)
        if d3_bufs is not None:
            dm, dsh, nptr = d3_bufs
            batch_cell_list(sysd["pos32b"], D3["cutoff"], sysd["cell32b"], sysd["pbc"], sysd["bi"], neighbor_matrix=dm,
                            neighbor_matrix_shifts=dsh, num_neighbors=nptr)
            mark("nlist_d3")
            e_d3, f_d3, cn, vir = dftd3(sysd["pos32b"], sysd["numbers"], neighbor_matrix=dm, neighbor_matrix_shifts=dsh, fill_value=n, **common)
        else:
            lst, nptr, lsh = batch_cell_list(sysd["pos32b"], D3["cutoff"], sysd["cell32b"], sysd["pbc"], sysd["bi"], return_neighbor_list=True)
            mark("nlist_d3")
            e_d3, f_d3, cn, vir = dftd3(sysd["pos32b"], sysd["numbers"], neighbor_list=lst, neighbor_ptr=nptr, unit_shifts=lsh, **common)
        mark("d3")
        local = torch.stack([e_d3.double(), segment_energy(e_pme, sysd["bi"], nsys)], dim=1)
        if world > 1:
            local = all_gather_system_values(local, sizes)
            mark("gather")
        if record is not None:
            record.append(ev)
        return e_pme, f_pme, e_d3, f_d3, num, nptr

    step.d3_bufs = d3_bufs
    return step, {}


VIRIAL = True
BUFFER_REPORT: dict = {}  # how the 40-Bohr list's buffers were chosen (tuned_neighbor_buffers)
SEARCH_CN = False  # set after the warm-up: did the 40-Bohr search also sum the DFT-D3 coordination numbers (round 6)?
COMPANION = False  # set after the warm-up: does the 40-Bohr matrix carry the packed companion the D3 passes stream (round 5)?
D3_FORMAT = "matrix"  # D3 leg: padded neighbour matrix (default, the format the reference's D3 benchmark uses) or "csr" (exact-size COO/CSR)
OVERLAP = False


def _hip_runtime():
    """ctypes handle of the HIP runtime torch already loaded (same library object, so its streams are torch's streams)."""
    path = "libamdhip64.so"
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
    except OSError:
        pass
    return ctypes.CDLL(path)


def cu_masked_streams(device):
    """Tuning aid: CU-partitioned streams.  BENCH_CU_SIDE=k gives the side (electrostatics) stream k CUs of every XCD
    (`hipExtStreamCreateWithCUMask`; mask bits are striped over the 8 XCDs, so bits [0, 8k) are k CUs on each) and, with
    BENCH_CU_MAIN=complement, the main (dispersion) stream the other 32 - k.  Returns (main | None, side | None) as torch ExternalStreams."""
    k = int(os.environ.get("BENCH_CU_SIDE", "0"))
    if k <= 0:
        return None, None
    hip = _hip_runtime()
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    nx = 8
    layout = os.environ.get("BENCH_CU_LAYOUT", "striped")
    side_bits = [i for i in range(ncu) if ((i // nx) < k if layout == "striped" else (i % (ncu // nx)) < k)]
    words = (ncu + 31) // 32

    def make(bits):
        m = [0] * words
        for b in bits:
            m[b // 32] |= 1 << (b % 32)
        st = ctypes.c_void_p()
        arr = (ctypes.c_uint32 * words)(*m)
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), arr)
        if rc != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
        return torch.cuda.ExternalStream(st.value, device=device)

    side = make(side_bits)
    main = make([i for i in range(ncu) if i not in set(side_bits)]) if os.environ.get("BENCH_CU_MAIN", "complement") == "complement" else None
    return main, side


def d3_search_context(numbers, params):
    """What lets the searches into a D3 list's buffers also sum the DFT-D3 coordination numbers (round 6): set-up, like the
    buffers themselves.  BENCH_SEARCH_CN=0 (A/B switch): the companion only, dftd3 runs its own CN pass."""
    if os.environ.get("BENCH_SEARCH_CN", "1") == "0":
        return True
    from nvalchemiops.neighborlist import D3SearchContext

    return D3SearchContext(numbers, params.rcov, 16.0)


def list_buffers(pos, cutoff, cell, pbc, m, device, batch_idx=None, for_dftd3=False, report=None):
    """(matrix, shifts, counts) for a list the workload searches into every step: chosen among a few candidate allocations by a trial search
    (`nvalchemiops.neighborlist.tuned_neighbor_buffers`) -- set-up, outside every timed region; BENCH_TUNED_BUFFERS=0: plain torch.empty."""
    n = pos.shape[0]
    if os.environ.get("BENCH_TUNED_BUFFERS", "1") == "0":
        bufs = (torch.empty((n, m), dtype=torch.int32, device=device), torch.empty((n, m, 3), dtype=torch.int32, device=device),
                torch.empty(n, dtype=torch.int32, device=device))
        if for_dftd3 is not False and for_dftd3 is not True:  # what a caller with plain buffers does once: announce the species to the search
            from nvalchemiops.neighborlist import attach_dftd3_context

            attach_dftd3_context(bufs[0], for_dftd3.numbers, for_dftd3.rcov, for_dftd3.k1)
        return bufs
    from nvalchemiops.neighborlist import tuned_neighbor_buffers

    return tuned_neighbor_buffers(pos, cutoff, cell, pbc, m, batch_idx=batch_idx, for_dftd3=for_dftd3,
                                  candidates=int(os.environ.get("BENCH_BUFFER_CANDIDATES", "20")), report=report)


def d3_branch(sysd, params, cell_list, dftd3, n, d3_bufs, between=None):
    """40-Bohr neighbour list + DFT-D3(BJ) with forces and virial, in the selected list format; `between()` is called after the
    list is enqueued (stage timing).  Returns (energy, forces, per-atom counts [matrix] or neighbor_ptr [csr])."""
    common = dict(a1=D3["a1"], a2=D3["a2"], s8=D3["s8"], d3_params=params, cell=sysd["cell32b"].unsqueeze(0), compute_virial=VIRIAL,
                  num_systems=1)
    if D3_FORMAT == "matrix":
        dm, dsh, dnum = d3_bufs
        cell_list(sysd["pos32b"], D3["cutoff"], sysd["cell32b"], sysd["pbc"], neighbor_matrix=dm, neighbor_matrix_shifts=dsh, num_neighbors=dnum)
        if between:
            between()
        out = dftd3(sysd["pos32b"], sysd["numbers"], neighbor_matrix=dm, neighbor_matrix_shifts=dsh, fill_value=n, **common)
        return out[0], out[1], dnum
    lst, nptr, lsh = cell_list(sysd["pos32b"], D3["cutoff"], sysd["cell32b"], sysd["pbc"], return_neighbor_list=True)
    if between:
        between()
    out = dftd3(sysd["pos32b"], sysd["numbers"], neighbor_list=lst, neighbor_ptr=nptr, unit_shifts=lsh, **common)
    return out[0], out[1], nptr


def make_step(sysd, tables, device, world):
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
    from nvalchemiops.neighborlist import cell_list

    n = sysd["n"]
    t = lambda a: torch.as_tensor(a, device=device)  # noqa: E731
    params = D3Parameters(rcov=t(tables["rcov"]), r4r2=t(tables["r4r2"]), c6ab=t(tables["c6ab"]), cn_ref=t(tables["cn_ref"]))
    m = PME["max_neighbors"]
    nm = torch.empty((n, m), dtype=torch.int32, device=device)
    nsh = torch.empty((n, m, 3), dtype=torch.int32, device=device)
    num = torch.empty(n, dtype=torch.int32, device=device)
    d3_bufs = None
    if D3_FORMAT == "matrix":
        # set-up, outside the timed region: make sure the configured row width holds this rank's box (replica boxes differ in their
        # jitter seed); the counts returned by a search keep counting past the row width, so one trial build tells
        md = D3["max_neighbors"]
        trial = cell_list(sysd["pos32b"], D3["cutoff"], sysd["cell32b"], sysd["pbc"], max_neighbors=64)[1]
        need = int(trial.max().item())
        if need > md:
            md = D3["max_neighbors"] = (need + 63) // 64 * 64
        del trial
        if os.environ.get("BENCH_ARENA"):
            # tuning aid (list-fill drift): the two row buffers carved at 2 MiB-aligned offsets out of ONE large block
            gib = float(os.environ["BENCH_ARENA"])
            arena = torch.empty(int(gib * (1 << 30)), dtype=torch.uint8, device=device)
            base = (-arena.data_ptr()) % (2 << 20)
            nb_j, nb_s = n * md * 4, n * md * 12
            off_s = base + (nb_j + (2 << 20) - 1) // (2 << 20) * (2 << 20)
            d3_bufs = (arena[base:base + nb_j].view(torch.int32).view(n, md), arena[off_s:off_s + nb_s].view(torch.int32).view(n, md, 3),
                       torch.empty(n, dtype=torch.int32, device=device))
        else:
            # set-up, outside the timed region: the row buffers are chosen among a few candidate allocations by a trial search (the fill's
            # time is a property of where the driver placed them) -- what an MD code does once when it allocates its lists
            d3_bufs = list_buffers(sysd["pos32b"], D3["cutoff"], sysd["cell32b"], sysd["pbc"], md, device,
                                   for_dftd3=d3_search_context(sysd["numbers"], params), report=BUFFER_REPORT)
    gathered = [torch.zeros(2, dtype=torch.float64, device=device) for _ in range(world)] if world > 1 else None
    stage_ms = {}
    side = torch.cuda.Stream(device=device, priority=int(os.environ.get("BENCH_SIDE_PRIORITY", "0")))
    side2 = torch.cuda.Stream(device=device)  # schedule 5 only: the real-space half on a stream of its own
    cu_main, cu_side = cu_masked_streams(device)  # tuning aid, off by default
    if cu_side is not None:
        side = cu_side
        if cu_main is not None:
            torch.cuda.set_stream(cu_main)

    def step(record=None):
        ev = []

        def mark(name):
            if record is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((name, e))

        mark("start")
        if OVERLAP:
            # The electrostatics branch (9 A list + PME) and the dispersion branch (40 Bohr list + D3) are independent: they are
            # enqueued on two HIP streams so the latency-bound kernels of one fill the gaps of the other; joined before the gather.
            main = torch.cuda.current_stream()

            def pme_branch():
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    cell_list(sysd["pos64"], PME["cutoff"], sysd["cell64"], sysd["pbc"], neighbor_matrix=nm, neighbor_matrix_shifts=nsh,
                              num_neighbors=num)
                    return particle_mesh_ewald(sysd["pos64"], sysd["q64"], sysd["cell64"], alpha=PME["alpha"],
                                               mesh_dimensions=PME["mesh"], spline_order=PME["order"], neighbor_matrix=nm,
                                               neighbor_matrix_shifts=nsh, compute_forces=True)

            if OVERLAP == 1:
                e_pme, f_pme = pme_branch()
                e_d3, f_d3, nptr = d3_branch(sysd, params, cell_list, dftd3, n, d3_bufs)
            elif OVERLAP == 3:
                # schedule 3 (tuning aid): particle_mesh_ewald as its two public halves -- list + real-space sum (fp64 VALU-heavy) beside the
                # HBM-bound 40-Bohr list write, the reciprocal half (spread / FFTs / gather: light on VALU) behind an event recorded after that
                # list, i.e. beside the D3 passes
                from nvalchemiops.interactions.electrostatics import ewald_real_space, pme_reciprocal_space

                side.wait_stream(main)
                with torch.cuda.stream(side):
                    cell_list(sysd["pos64"], PME["cutoff"], sysd["cell64"], sysd["pbc"], neighbor_matrix=nm, neighbor_matrix_shifts=nsh,
                              num_neighbors=num)
                    al = torch.full((1,), PME["alpha"], dtype=torch.float64, device=device)
                    e_r, f_r = ewald_real_space(sysd["pos64"], sysd["q64"], sysd["cell64"], al, neighbor_matrix=nm, neighbor_matrix_shifts=nsh,
                                                mask_value=n, compute_forces=True)
                ev_list = torch.cuda.Event()

                def after_list():
                    ev_list.record(main)
                    with torch.cuda.stream(side):
                        side.wait_event(ev_list)
                        e_k, f_k = pme_reciprocal_space(sysd["pos64"], sysd["q64"], sysd["cell64"], PME["alpha"], mesh_dimensions=PME["mesh"],
                                                        spline_order=PME["order"], compute_forces=True)
                        box.append((e_r + e_k, f_r + f_k))

                box = []
                e_d3, f_d3, nptr = d3_branch(sysd, params, cell_list, dftd3, n, d3_bufs, after_list)
                e_pme, f_pme = box[0]
            elif OVERLAP == 4:
                # schedule 4 (tuning aid, round 6): the side stream runs the reciprocal half BEFORE the real-space sum -- the mesh solve's blocks
                # need a whole CU's LDS and sit out every long kernel of the main stream, so they should meet the gap between the 40-Bohr fill
                # and the energy pass; the real-space sum and the gather (small blocks) can share CUs with the D3 passes
                from nvalchemiops.interactions.electrostatics import ewald_real_space, pme_reciprocal_space

                side.wait_stream(main)
                with torch.cuda.stream(side):
                    cell_list(sysd["pos64"], PME["cutoff"], sysd["cell64"], sysd["pbc"], neighbor_matrix=nm, neighbor_matrix_shifts=nsh,
                              num_neighbors=num)
                    e_k, f_k = pme_reciprocal_space(sysd["pos64"], sysd["q64"], sysd["cell64"], PME["alpha"], mesh_dimensions=PME["mesh"],
                                                    spline_order=PME["order"], compute_forces=True)
                    al = torch.full((1,), PME["alpha"], dtype=torch.float64, device=device)
                    e_r, f_r = ewald_real_space(sysd["pos64"], sysd["q64"], sysd["cell64"], al, neighbor_matrix=nm, neighbor_matrix_shifts=nsh,
                                                mask_value=n, compute_forces=True)
                    e_pme, f_pme = e_r + e_k, f_r + f_k
                e_d3, f_d3, nptr = d3_branch(sysd, params, cell_list, dftd3, n, d3_bufs)
            elif OVERLAP == 5:
                # schedule 5 (round 6): three streams.  The mesh solve's plane kernels occupy a whole CU's LDS each and are only ever placed at a
                # kernel BOUNDARY of the main stream.  With the reciprocal half on a stream of its own from t = 0, its spread is done
                # long before the 40-Bohr fill ends, plane kernel A meets the fill -> energy boundary, the column kernels run beside the energy
                # pass, plane kernel C meets the energy -> chain boundary and the gather runs beside the chain pass; the 9 A list and the
                # real-space sum (fp64 VALU) keep their place beside the HBM-bound fill, on the third stream.
                from nvalchemiops.interactions.electrostatics import ewald_real_space, pme_reciprocal_space

                side.wait_stream(main)
                side2.wait_stream(main)
                with torch.cuda.stream(side):
                    e_k, f_k = pme_reciprocal_space(sysd["pos64"], sysd["q64"], sysd["cell64"], PME["alpha"], mesh_dimensions=PME["mesh"],
                                                    spline_order=PME["order"], compute_forces=True)
                with torch.cuda.stream(side2):
                    cell_list(sysd["pos64"], PME["cutoff"], sysd["cell64"], sysd["pbc"], neighbor_matrix=nm, neighbor_matrix_shifts=nsh,
                              num_neighbors=num)
                    al = torch.full((1,), PME["alpha"], dtype=torch.float64, device=device)
                    e_r, f_r = ewald_real_space(sysd["pos64"], sysd["q64"], sysd["cell64"], al, neighbor_matrix=nm, neighbor_matrix_shifts=nsh,
                                                mask_value=n, compute_forces=True)
                e_d3, f_d3, nptr = d3_branch(sysd, params, cell_list, dftd3, n, d3_bufs)
                with torch.cuda.stream(side):
                    side.wait_stream(side2)
                    e_pme, f_pme = e_r + e_k, f_r + f_k
                    for t in (e_r, f_r):
                        t.record_stream(side)
            else:  # schedule 2 (tuning aid): the PME branch is enqueued after the D3 list, next to the D3 passes only
                box = []
                e_d3, f_d3, nptr = d3_branch(sysd, params, cell_list, dftd3, n, d3_bufs, lambda: box.append(pme_branch()))
                e_pme, f_pme = box[0]
            main.wait_stream(side)
            for t in (e_pme, f_pme):
                t.record_stream(main)
            mark("pme||d3")
        else:
            cell_list(sysd["pos64"], PME["cutoff"], sysd["cell64"], sysd["pbc"], neighbor_matrix=nm, neighbor_matrix_shifts=nsh,
                      num_neighbors=num)
            mark("nlist_pme")
            e_pme, f_pme = particle_mesh_ewald(sysd["pos64"], sysd["q64"], sysd["cell64"], alpha=PME["alpha"], mesh_dimensions=PME["mesh"],
                                               spline_order=PME["order"], neighbor_matrix=nm, neighbor_matrix_shifts=nsh, compute_forces=True)
            mark("pme")
            e_d3, f_d3, nptr = d3_branch(sysd, params, cell_list, dftd3, n, d3_bufs, lambda: mark("nlist_d3"))
            mark("d3")
        if gathered is not None:
            mine = torch.stack([e_d3[0].double(), e_pme.sum()])
            torch.distributed.all_gather(gathered, mine)
            mark("gather")
        if record is not None:
            record.append(ev)
        return e_pme, f_pme, e_d3, f_d3, num, nptr

    step.d3_bufs = d3_bufs
    return step, stage_ms


def kernel_report():
    """{kernel: (launches, total_ms, median_ms, min_ms, max_ms)} from the library's HIP-event records (cleared by the call)."""
    from nvalchemiops import _capi as C

    buf = ctypes.create_string_buffer(1 << 16)
    C.lib().mi_timing_report_stats(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot, med, lo, hi = line.rsplit(" ", 5)
        out[name] = (int(cnt), float(tot), float(med), float(lo), float(hi))
    return out


def profile_lookup(path: str, kernel: str, field: str, atoms: int, workload: str):
    """A per-launch figure of `kernel` from a committed rocprofv3 PMC summary of this very workload (counters cannot be read from
    inside the timed run): (value, "profiles/<file>") or (None, None)."""
    if workload != "headline" or atoms != 100000 or not os.path.exists(path):
        return None, None
    try:
        v = json.load(open(path)).get("kernels", {}).get(kernel, {}).get(field)
    except Exception:
        return None, None
    return (v, os.path.relpath(path, ROOT)) if v is not None else (None, None)


def traffic_profile():
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            return path
    return ""


def kernel_accounting(kernel: str, n: int, pairs_d3: int):
    """(bound, algorithmic bytes per launch by SURVEY.md 8(d), bytes this design moves per launch, note).

    8(d): nlist padded  N(3s+4) + 16 N M;   D3 per pass  16 x entries + N(3s+4) in + N(12+4) out;   real-space  16 N M + N(4s) + N(4s);
    spread  N 4s + mesh s;   gather  4 mesh s + N 4s + N 4s;   convolve  spec 2s (1 read + 4 writes).   s = bytes per position scalar.
    `design` differs where the kernel deliberately moves other bytes than the formula's: the D3 CN pass also WRITES a 4 B/slot packed
    copy of a periodic padded list, which is all the energy and chain passes stream afterwards."""
    m, mesh = PME["max_neighbors"], float(np.prod(PME["mesh"]))
    packed = D3_FORMAT == "matrix" and os.environ.get("NVALCHEMIOPS_D3_PACKED_LIST", "1") != "0"
    if kernel in ("d3_energy", "d3_cn", "d3_chain"):
        entries = float(n) * D3["max_neighbors"] if D3_FORMAT == "matrix" else float(pairs_d3)
        algo = 16.0 * entries + 40.0 * n
        cn_bytes = 4.0 if COMPANION else 20.0  # with the search's companion the CN pass streams 4 B/slot and writes nothing
        if kernel == "d3_cn" and SEARCH_CN:
            # the neighbour search summed the coordination numbers while it wrote the list: what is left of this pass is one small launch
            # that verifies the fingerprint of the inputs and copies N floats (no list walk; its bytes are priced with the search kernel)
            return "latency", None, 12.0 * n, "adopts the coordination numbers the neighbour search summed (fingerprint check + N-float copy); no list walk"
        design = ((cn_bytes if kernel == "d3_cn" else 4.0) * entries + 40.0 * n) if packed else algo
        if kernel == "d3_energy":
            return "valu", algo, design, "C6 contraction + BJ damping per directed pair: VALU-issue-bound, not HBM-bound (DESIGN.md 3.1)"
        if COMPANION and packed:
            return "hbm", algo, design, "streams the 4 B/slot companion the neighbour search wrote, gathers one 16 B record per neighbour"
        return "hbm", algo, design, ("streams the caller's 16 B/slot list, gathers one 16 B record per neighbour" +
                                      (", writes the 4 B/slot packed copy" if packed and kernel == "d3_cn" else ""))
    if kernel == "nl_query_csr":
        return "hbm", n * (3 * 4 + 8) + 20.0 * pairs_d3, None, ""
    if kernel == "nl_query_count":
        return "valu", n * (3 * 4 + 4.0), None, "distance tests only"
    if kernel == "nl_query_matrix_f64":
        return "hbm", n * (3 * 8 + 4) + 16.0 * n * m, None, ""
    if kernel == "nl_query_matrix_f32":
        algo = n * (3 * 4 + 4) + 16.0 * n * D3["max_neighbors"]
        moved = (algo + 4.0 * n * D3["max_neighbors"]) if COMPANION else None
        if SEARCH_CN:
            # the launch does the work of TWO rows of SURVEY 8(d): the padded-matrix search and the first D3 pass (`_cn_kernel_nm`), whose
            # algorithmic bytes -- 16 B per slot + 40 B per atom -- it never moves because the pairs are still in registers.  `algorithmic_bytes`
            # is the sum of the two rows; `algorithmic_bytes_list_only` keeps the search's own figure beside it
            algo += 16.0 * n * D3["max_neighbors"] + 40.0 * n
        return ("hbm", algo, moved,
                "HBM writes + 6.3e8 distance tests" + ("; also writes the 4 B/slot packed companion for the D3 passes" if COMPANION else "") +
                ("; also sums the DFT-D3 coordination numbers over its hits (the D3 CN pass's work, none of its bytes)" if SEARCH_CN else ""))
    if kernel == "ewald_real":
        return "hbm", 16.0 * n * m + n * (3 * 8 + 8) + n * (8 + 3 * 8), None, ""
    if kernel == "spline_spread":
        return "latency", n * 4 * 8 + mesh * 8, None, "binning + LDS-tile accumulation, 4 small launches"
    if kernel == "pme_gather_finish":
        return "hbm", 4 * mesh * 8 + n * 4 * 8 + n * 4 * 8, None, "L2/MALL-resident mesh gather"
    if kernel == "pme_convolve":
        return "hbm", (mesh / 2) * 16 * 5, None, ""
    # fused mesh solve (mi_pme_solve): complex half-spectra of 16 B per point; every kernel reads and writes its planes / columns once
    if kernel == "pme_solve_fwd":
        return "hbm", mesh * 8 + (mesh / 2) * 16, None, "real planes in, (y,z)-transformed half spectrum out"
    if kernel == "pme_solve_cols":
        return "hbm", (mesh / 2) * 16 * (2 + 1 + 4), None, "x transform + Green function in place, then 4 channels back along x"
    if kernel == "pme_solve_inv":
        return "hbm", 4 * ((mesh / 2) * 16 + mesh * 8), None, "4 channels: half spectrum in, real planes out"
    return "latency", None, None, ""


def kernel_table(kernels, isolated, n, pairs_d3, workload):
    """One row per timed kernel (see the module docstring)."""
    rows = {}
    tpath = traffic_profile()
    for name, (cnt, tot, med, lo, hi) in sorted(kernels.items()):
        bound, algo, design, note = kernel_accounting(name, n, pairs_d3)
        iso = isolated.get(name)
        t_ms = iso[2] if iso else med  # the isolated median is the reproducible figure (profiles/*_serial.csv); fall back to the timed median
        row = {"launches": cnt, "avg_ms_timed_region": tot / cnt, "median_ms_timed_region": med,
               "isolated_median_ms": iso[2] if iso else None, "isolated_min_ms": iso[3] if iso else None, "isolated_max_ms": iso[4] if iso else None,
               "bound": bound, "algorithmic_bytes": algo, "design_bytes": design}
        if name == "nl_query_matrix_f32" and SEARCH_CN and algo:
            row["algorithmic_bytes_list_only"] = n * (3 * 4 + 4) + 16.0 * n * D3["max_neighbors"]
            row["frac_of_hbm_peak_list_only"] = row["algorithmic_bytes_list_only"] / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if algo:
            row["algorithmic_GBps"] = algo / (t_ms * 1e-3) / 1e9
            row["frac_of_hbm_peak"] = row["algorithmic_GBps"] / HBM_PEAK_GBS
        if design:  # the bytes this design really streams (energy / chain read 4 B/slot, not the 16 the formula prices)
            row["design_GBps"] = design / (t_ms * 1e-3) / 1e9
            row["design_frac_of_hbm_peak"] = row["design_GBps"] / HBM_PEAK_GBS
        traffic, src = profile_lookup(tpath, name, "hbm_bytes_per_launch", n, workload)
        row["traffic_bytes"], row["traffic_from_profile"] = traffic, src
        if bound == "valu":
            insts, vsrc = profile_lookup(VALU_PROFILE, name, "valu_wave_insts_per_launch", n, workload)
            if insts:
                row["valu_wave_insts"] = insts
                row["valu_insts_from_profile"] = vsrc
                row["valu_Ginstr_per_s"] = insts / (t_ms * 1e-3) / 1e9
                row["frac_of_valu_issue_peak"] = row["valu_Ginstr_per_s"] / VALU_PEAK_GINSTR
                if name.startswith("d3_") and pairs_d3:
                    row["valu_insts_per_64_pairs"] = insts / (pairs_d3 / 64.0)
        if note:
            row["note"] = note
        rows[name] = row
    return rows


def _in_step_ms(r):
    return r["avg_ms_timed_region"]


def roofline_of(rows):
    """The contract's `roofline` object for the dominant kernel = the kernel with the most WORK: the largest isolated launch duration (untimed
    serial pass), falling back to the in-step duration where no isolated figure exists (config workloads run on one stream: the same thing).
    Not simply the longest in-step duration: in the two-stream step the electrostatics branch's kernels are starved behind the dispersion
    branch's persistent blocks and stretch 3 - 15 x (the 0.23 ms fp64 list shows 1.0 - 1.3 ms in-step) without being what the step waits
    for; they are listed under `stretched_side_stream`.  `achieved` / `frac` use the dominant kernel's in-step average (algorithmic bytes per
    launch / average launch duration over the timed region, HIP events on the kernel's own stream); `achieved_isolated` / `frac_isolated` are
    the same bytes over the isolated median, the figure that is comparable across runs and with profiles/*_serial.csv."""
    if not rows:
        return None
    name, r = max(rows.items(), key=lambda kv: (kv[1].get("isolated_median_ms") or _in_step_ms(kv[1])))
    t_ms, iso_ms = _in_step_ms(r), r["isolated_median_ms"]
    if r["bound"] == "valu" and r.get("valu_wave_insts"):
        rate = r["valu_wave_insts"] / (t_ms * 1e-3) / 1e9
        out = {"bound": "valu", "kernel": name, "achieved": rate, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s", "frac": rate / VALU_PEAK_GINSTR,
               "achieved_isolated": r.get("valu_Ginstr_per_s"), "frac_isolated": r.get("frac_of_valu_issue_peak"),
               "hbm_frac": (r["algorithmic_bytes"] / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if r.get("algorithmic_bytes") else None}
    else:
        ach = (r["algorithmic_bytes"] / (t_ms * 1e-3) / 1e9) if r.get("algorithmic_bytes") else None
        out = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS if ach else None,
               "achieved_isolated": r.get("algorithmic_GBps") if iso_ms else None, "frac_isolated": r.get("frac_of_hbm_peak") if iso_ms else None}
    stretched = {k: {"in_step_ms": _in_step_ms(v), "isolated_median_ms": v["isolated_median_ms"]} for k, v in rows.items()
                 if v.get("isolated_median_ms") and _in_step_ms(v) > max(2.0 * v["isolated_median_ms"], t_ms * 0.5) and k != name}
    if stretched:
        out["stretched_side_stream"] = stretched
    out.update({"traffic": r["traffic_bytes"], "traffic_from_profile": r["traffic_from_profile"], "launch_ms": t_ms,
                "launch_ms_kind": "average launch duration inside the timed region (HIP events on the kernel's stream; the only bracketed kernel there)",
                "dominant_by": "largest isolated launch duration (the kernel with the most work)",
                "launch_ms_isolated_median": iso_ms, "launches": r["launches"],
                "algorithmic_bytes_per_launch": r["algorithmic_bytes"], "design_bytes_per_launch": r["design_bytes"]})
    if r.get("algorithmic_bytes_list_only"):
        # the fused search + CN launch: the same durations against the search's own 8(d) row alone (what rounds 1-5 reported for this kernel)
        out["list_only"] = {"algorithmic_bytes_per_launch": r["algorithmic_bytes_list_only"],
                            "achieved": r["algorithmic_bytes_list_only"] / (t_ms * 1e-3) / 1e9, "frac": r["algorithmic_bytes_list_only"] / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "frac_isolated": (r["algorithmic_bytes_list_only"] / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if iso_ms else None,
                            "note": "algorithmic bytes of the padded-matrix search alone; `achieved` / `frac` above add the D3 CN pass's 8(d) bytes, "
                                    "whose work this launch does without moving them"}
    # the dominant HBM-bound kernel beside it when the dominant kernel is VALU-bound
    hb = {k: v for k, v in rows.items() if v["bound"] == "hbm" and v.get("algorithmic_bytes")}
    if out["bound"] != "hbm" and hb:
        k, v = max(hb.items(), key=lambda kv: _in_step_ms(kv[1]))
        ach = v["algorithmic_bytes"] / (_in_step_ms(v) * 1e-3) / 1e9
        out["dominant_hbm_kernel"] = {"kernel": k, "achieved": ach, "frac": ach / HBM_PEAK_GBS, "launch_ms": _in_step_ms(v),
                                      "frac_isolated": v.get("frac_of_hbm_peak"), "launch_ms_isolated_median": v["isolated_median_ms"],
                                      "algorithmic_bytes_per_launch": v["algorithmic_bytes"], "design_bytes_per_launch": v["design_bytes"],
                                      "traffic": v["traffic_bytes"]}
    return out


# ---- in-run HBM calibration (make a slow box distinguishable from a slow build) -------------------------------------------
def hbm_calibration(device, gib: float = 4.0, reps: int = 5):
    """Copy and fill rates of THIS box, measured right before the timed region with the library's plain 16-byte-per-lane kernels
    (csrc/calib.hip) over `gib` GiB: `copy_GBps` counts read + written bytes, `fill_GBps` written bytes (median of `reps`)."""
    from nvalchemiops import _capi as C

    nbytes = int(gib * (1 << 30)) // 16 * 16
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=device)
    src.zero_()
    st = C.stream_of(src)
    L = C.lib()

    def med(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        return statistics.median(ts)

    sink = torch.zeros(1, dtype=torch.float32, device=device)
    t_copy = med(lambda: C.check(L.mi_calibrate_copy(C.ptr(src), C.ptr(dst), ctypes.c_size_t(nbytes), st), "mi_calibrate_copy"))
    t_fill = med(lambda: C.check(L.mi_calibrate_fill(C.ptr(dst), ctypes.c_size_t(nbytes), ctypes.c_float(1.0), st), "mi_calibrate_fill"))
    t_read = med(lambda: C.check(L.mi_calibrate_read(C.ptr(src), ctypes.c_size_t(nbytes), C.ptr(sink), st), "mi_calibrate_read"))
    del src, dst
    torch.cuda.empty_cache()  # hand the two calibration blocks back: they must not shape the allocator state of the timed region
    return {"copy_GBps": 2.0 * nbytes / t_copy / 1e6, "fill_GBps": nbytes / t_fill / 1e6, "read_GBps": nbytes / t_read / 1e6, "bytes": nbytes,
            "copy_ms": t_copy, "fill_ms": t_fill, "read_ms": t_read,
            "note": "float4 read / fill / copy streams over this many bytes on this box, before the timed region (csrc/calib.hip); copy counts read + written bytes"}


# ---- CPU baseline (the oracle; reported, not the target) ----------------------------------------------------------------------------
def _oracle_step(O, pos, cell, q, numbers, tables, mesh, order):
    """One full step of the oracle on host arrays, the workload of the GPU step: list 9 A (M = 256) -> PME (E + F) -> list 40 Bohr
    (padded matrix) -> D3 (E + F + virial).  Returns the four stage times."""
    pb, cb = (pos * BOHR).astype(np.float32), (cell * BOHR).astype(np.float32)
    t0 = time.perf_counter()
    nm, num, sh = O.cell_list(pos, PME["cutoff"], cell, [True] * 3, max_neighbors=PME["max_neighbors"])
    t1 = time.perf_counter()
    with O.extended_splines():  # order 5 as in the GPU step ("beyond reference" mode of the oracle: its reference mode is zero there)
        ep = O.particle_mesh_ewald(pos, q, cell, PME["alpha"], mesh, order, neighbor_matrix=nm, neighbor_matrix_shifts=sh, compute_forces=True)[0]
    t2 = time.perf_counter()
    nm2, num2, sh2 = O.cell_list(pb, D3["cutoff"], cb, [True] * 3, max_neighbors=D3["max_neighbors"])
    t3 = time.perf_counter()
    ed = O.dftd3(pb, numbers, tables, D3["a1"], D3["a2"], D3["s8"], neighbor_matrix=nm2, neighbor_matrix_shifts=sh2, cell=cb, compute_virial=True)[0]
    t4 = time.perf_counter()
    _oracle_step.energies = {"e_d3_Ha": float(np.asarray(ed).reshape(-1)[0]), "e_pme": float(np.asarray(ep, dtype=np.float64).sum())}
    _oracle_step.d3_inputs = (pb, numbers, nm2, sh2, cb)  # for the wide-sum check of the parity object (dropped right after)
    return np.array([t1 - t0, t2 - t1, t3 - t2, t4 - t3])


def cpu_baseline(sample_atoms: int, full_size_1thread: bool, budget_s: float = 12.0):
    """The CPU oracle on this host.  `value`: 1 thread (what Warp's CPU backend does per launch: SURVEY F9) on a bounded sample -- full
    steps of the same workload (same density, cutoffs, alpha, spline order; mesh spacing ~0.94 A) on a smaller periodic box whose edge
    is >= 2 x the D3 cutoff, so the per-atom pair counts are those of the 100k-atom box.  `full_size`: the all-core OpenMP build of the
    same source on the FULL 100k-atom step (skipped when the sample predicts > 60 s), plus the 1-thread full-size leg on request."""
    from oracle import oracle as O
    from tests import systems as S

    tables = S.d3_test_tables(94, seed=7)
    pos, cell, q, numbers = S.fcc_box(sample_atoms, seed=1234, dtype=np.float64)
    edge = float(cell[0, 0])
    mesh = (int(2 * round(edge / 0.9375 / 2)),) * 3
    stages, steps, t_begin = np.zeros(4), 0, time.perf_counter()
    while True:
        stages += _oracle_step(O, pos, cell, q, numbers, tables, mesh, PME["order"])
        steps += 1
        if time.perf_counter() - t_begin >= budget_s or steps >= 40:
            break
    total, per = float(stages.sum()), stages / steps
    out = {
        "value": sample_atoms * steps / total, "unit": "atom-steps/s", "cores": 1, "kind": "port",
        "sample": f"{steps} steps on a {sample_atoms}-atom periodic box ({edge:.0f} A edge) at the same density/cutoffs (mesh {mesh[0]}^3, spline order "
                  f"{PME['order']} in the oracle's extended mode); per step: nlist9A {per[0]:.3f}s, PME {per[1]:.3f}s, nlist40Bohr {per[2]:.3f}s, D3 {per[3]:.3f}s",
        "seconds": total, "host_cores": os.cpu_count(),
    }
    full = {}
    cores = os.cpu_count() or 1
    predicted_1t = 100000.0 / out["value"]
    fpos, fcell, fq, fnum = S.fcc_box(100000, seed=1234, dtype=np.float64)
    if cores >= 2 and predicted_1t / (0.5 * cores) < 60.0:
        with O.openmp(cores) as om:
            st = _oracle_step(O, fpos, fcell, fq, fnum, tables, PME["mesh"], PME["order"])
            full["all_cores"] = {"value": 100000.0 / float(st.sum()), "unit": "atom-steps/s", "cores": om.threads, "seconds": float(st.sum()),
                                 "stage_s": {"nlist9A": st[0], "pme": st[1], "nlist40Bohr": st[2], "d3": st[3]},
                                 "what": "one full 100k-atom step, OpenMP build of the oracle (atom loops shared between threads, scipy.fft workers)"}
    else:
        full["all_cores"] = {"skipped": f"predicted {predicted_1t / max(0.5 * cores, 1):.0f} s on {cores} cores"}
    if full_size_1thread or predicted_1t < 75.0:
        # the 1-thread leg at FULL size whenever the sample predicts about a minute or less (the driver line's own
        # number should not be an extrapolation): `value` becomes this measurement, the sample-box figure stays beside it
        st = _oracle_step(O, fpos, fcell, fq, fnum, tables, PME["mesh"], PME["order"])
        full["one_thread"] = {"value": 100000.0 / float(st.sum()), "unit": "atom-steps/s", "cores": 1, "seconds": float(st.sum()),
                              "stage_s": {"nlist9A": st[0], "pme": st[1], "nlist40Bohr": st[2], "d3": st[3]},
                              "what": "one full 100k-atom step, serial oracle",
                              "energies": dict(getattr(_oracle_step, "energies", {}))}  # the oracle's answer for the box rank 0 times
        # the same D3 sum with every fp32 accumulation of the reference carried in double (`O.d3_wide_sums`: the pair arithmetic without the
        # summation-order noise of a 2.4e8-term fp32 sum) -- the figure the GPU's fp64 lane partials are compared with; untimed, all cores
        try:
            pb, zz, nm2, sh2, cb = _oracle_step.d3_inputs
            with O.openmp(cores), O.d3_wide_sums():
                ew = O.dftd3(pb, zz, tables, D3["a1"], D3["a2"], D3["s8"], neighbor_matrix=nm2, neighbor_matrix_shifts=sh2, cell=cb, compute_virial=True)[0]
            full["one_thread"]["energies"]["e_d3_Ha_wide_sums"] = float(np.asarray(ew).reshape(-1)[0])
        except Exception as exc:
            full["one_thread"]["energies"]["e_d3_wide_sums_error"] = f"{type(exc).__name__}: {exc}"[:200]
        out["sample_box"] = {"value": out["value"], "sample": out["sample"], "seconds": out["seconds"]}
        out["value"], out["seconds"] = full["one_thread"]["value"], full["one_thread"]["seconds"]
        out["sample"] = ("ONE full step of the headline workload itself (100 000-atom box, mesh 128^3, spline order 5 in the oracle's extended mode), "
                         f"serial oracle, 1 thread: nlist9A {st[0]:.2f}s, PME {st[1]:.2f}s, nlist40Bohr {st[2]:.2f}s, D3 {st[3]:.2f}s")
    else:
        out["sample"] = "SAMPLE BOX, not the 100k box (1-thread full-size step predicted > 75 s; --cpu-full-size runs it): " + out["sample"]
    out["full_size"] = full
    _oracle_step.d3_inputs = None
    return out


def add_parity(res):
    """`parity` of the bench line: the timed step's own energies against the oracle's for the SAME 100k-atom box (the 1-thread full-size CPU
    leg computes them anyway).  north_star's "|dE| < 1e-6 Ha" is met RELATIVELY (D3 energies are fp32 outputs, as in the reference: one ulp of
    a -2e4 Ha total is 2e-3 Ha); both figures are printed so nobody has to take that on trust."""
    o = ((res.get("cpu_baseline") or {}).get("full_size") or {}).get("one_thread", {}).get("energies")
    g = res.get("energies")
    if not o or not g or res.get("config", {}).get("atoms_per_gpu") != 100000:
        return
    wide = o.get("e_d3_Ha_wide_sums", o["e_d3_Ha"])  # the checker: the oracle with the reference's fp32 sums carried in double
    d3_abs = abs(g["e_d3_Ha"] - wide)
    pme_abs = abs(g["e_pme"] - o["e_pme"])
    res["parity"] = {
        "d3_energy_gpu_Ha": g["e_d3_Ha"], "d3_energy_oracle_Ha": wide, "d3_abs_dE_Ha": d3_abs, "d3_rel_dE": d3_abs / max(abs(wide), 1e-300),
        "d3_energy_oracle_reference_order_fp32_Ha": o["e_d3_Ha"], "d3_reference_order_minus_wide_Ha": o["e_d3_Ha"] - wide,
        "pme_energy_gpu": g["e_pme"], "pme_energy_oracle": o["e_pme"], "pme_abs_dE": pme_abs, "pme_rel_dE": pme_abs / max(abs(o["e_pme"]), 1e-300),
        "bar": "relative: |dE| <= 1e-6 |E| for the fp32 D3 energy (the reference's own CPU-vs-GPU rtol, test_dftd3.py:477-489), 1e-9 |E| for fp64 PME; "
               "the absolute D3 figure is bounded below by fp32 output rounding of the total (ulp(|E|) ~ 6e-8 |E|)",
        "oracle": "CPU restatement (oracle/), same box, its own lists; D3: the reference's pair arithmetic with its fp32 accumulations carried in double "
                  "(the reference-order fp32 sum of 2.4e8 terms -- also listed -- carries ~1e-6 relative summation noise of its own); neighbour indices are compared "
                  "bit-exactly in tests/test_nlist_gpu.py::test_headline_list_100k_40bohr_full_size_matches_oracle",
        "d3_within_bar": bool(d3_abs <= 1e-6 * abs(wide) + 1e-6), "pme_within_bar": bool(pme_abs <= 1e-9 * abs(o["e_pme"]) + 1e-9)}


def compact_configs(budget_cpu_s: float = 3.0):
    """BASELINE.json configs 2 / 3 / 4 for the driver's record: one short pass of `--workload cN` each, in its own
    process AFTER the headline's timed region, reduced to {ms, value, roofline kernel / frac, cpu_baseline}.  ~10 s of GPU work in total."""
    out = {}
    env = dict(os.environ, BENCH_CPU_BUDGET_S=str(budget_cpu_s), BENCH_CALIB_GIB="1")
    for name in ("c2", "c3", "c4"):
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", "50", "--warmup", "5", "--processes", "1"],
                           capture_output=True, text=True, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            out[name] = {"error": (r.stderr or "no output")[-300:]}
            continue
        d = json.loads(lines[-1])
        roof, cb = d.get("roofline") or {}, d.get("cpu_baseline") or {}
        out[name] = {"workload": d["config"]["workload"], "ms": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "dtype": d["dtype"],
                     "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launch_ms", "algorithmic_bytes_per_launch")},
                     "kernels_ms": {k: round(v["avg_ms_timed_region"], 4) for k, v in d.get("kernels", {}).items()},
                     "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")},
                     "wall_s": time.perf_counter() - t0}
    out.update(extra_records(env))
    return out


def _child(argv, env, timeout=240):
    """One child run of this script; (parsed last JSON line | None, error text)."""
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True, env=env, timeout=timeout)
    except subprocess.TimeoutExpired:
        return None, f"timeout after {timeout} s"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return None, (r.stderr or "no output")[-300:]
    return json.loads(lines[-1]), ""


def c5_cpu_sample(n_systems: int = 2, budget_s: float = 6.0):
    """CPU leg of BASELINE config 5 on a bounded sample: `n_systems` of the shard's 2000-atom periodic boxes (L = 32 A < 2 rc: every atom sees
    several images of the same neighbour), one after the other through the serial oracle -- systems of a batch are independent, so the
    per-atom work is the batch's.  Mesh 32^3, spline order 5 (extended mode), M sized to the fullest row as the GPU shard's is."""
    from oracle import oracle as O
    from tests import systems as S

    tables = S.d3_test_tables(94, seed=7)
    boxes = [S.fcc_box(2000, seed=1234 + 17 * b, dtype=np.float64) for b in range(n_systems)]
    saved = D3["max_neighbors"]
    t0, steps = time.perf_counter(), 0
    try:
        D3["max_neighbors"] = 2560
        while True:
            for pos, cell, q, numbers in boxes:
                _oracle_step(O, pos, cell, q, numbers, tables, (32, 32, 32), PME["order"])
            steps += 1
            if time.perf_counter() - t0 >= budget_s or steps >= 10:
                break
    finally:
        D3["max_neighbors"] = saved
        _oracle_step.d3_inputs = None
    sec = time.perf_counter() - t0
    return {"value": 2000.0 * n_systems * steps / sec, "unit": "atom-steps/s", "cores": 1, "kind": "port", "seconds": sec,
            "sample": f"{steps} passes of the serial oracle over {n_systems} of the shard's 2000-atom boxes (nlist 9 A + PME 32^3 order 5 + nlist 40 Bohr + D3 with virial each)"}


def extra_records(env):
    """What else the driver's line carries since round 6, each from its own short child process after the headline's
    timed region: BASELINE config 5 (one 128-system shard), the reference's three published benchmark rows with its own warm-up / median
    protocol (benchmarks/utils.py:133-240), and the headline step two more ways -- into plain `torch.empty` list buffers (what a caller who
    does not pick buffers by measurement gets) and replayed from a HIP graph (no launch gaps; per-kernel events impossible, hence not the
    reported mode)."""
    out = {}
    t0 = time.perf_counter()
    d, err = _child(["--workload", "c5", "--steps", "20", "--warmup", "3", "--processes", "1"], env)
    if d is None:
        out["c5"] = {"error": err}
    else:
        roof = d.get("roofline") or {}
        out["c5"] = {"workload": d["config"]["workload"], "ms": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "dtype": d["dtype"],
                     "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launch_ms", "algorithmic_bytes_per_launch")},
                     "kernels_ms": {k: round(v["avg_ms_timed_region"], 4) for k, v in d.get("kernels", {}).items()},
                     "wall_s": time.perf_counter() - t0}
        try:
            out["c5"]["cpu_baseline"] = c5_cpu_sample(budget_s=float(env.get("BENCH_CPU_BUDGET_S", "3")) * 2)
        except Exception as exc:
            out["c5"]["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    for name in ("ref-nlist", "ref-d3", "ref-pme"):
        t0 = time.perf_counter()
        d, err = _child(["--workload", name], env)
        out[name.replace("-", "_")] = {"error": err} if d is None else {**{k: v for k, v in d.items() if k not in ("data", "n_gpus")}, "wall_s": time.perf_counter() - t0}
    for key, extra_env, what in (("headline_plain_buffers", {"BENCH_TUNED_BUFFERS": "0"},
                                  "the headline step into plain torch.empty list buffers (no trial-search selection): one fresh process"),
                                 ("headline_graph_replay", {"BENCH_GRAPH": "1", "BENCH_CALIB": "0"},
                                  "the headline step captured once into a HIP graph (both streams) and replayed: one fresh process")):
        t0 = time.perf_counter()
        d, err = _child(["--steps", "100", "--warmup", "10", "--processes", "1", "--cpu-sample", "0"], dict(env, BENCH_CONFIGS="0", **extra_env))
        out[key] = ({"error": err} if d is None else
                    {"ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "what": what,
                     "list_fill_40bohr_isolated_median_ms": (d.get("kernels", {}).get("nl_query_matrix_f32") or {}).get("isolated_median_ms"),
                     "wall_s": time.perf_counter() - t0})
    return out


# ---- the reference's published benchmark configurations ---------------------------------------------------------------
def _median_ms(fn, warmup, iters):
    """benchmarks/utils.py:133-240: warm-up, then `iters` calls each bracketed by events on the op's stream; median."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    times = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        times.append(a.elapsed_time(b))
    return statistics.median(times), min(times), max(times)


def _lattice(n_atoms, basis, a, dtype):
    """First n_atoms sites of a cubic lattice with the given basis (benchmarks/systems.py:874-984 recipe: no jitter)."""
    per = len(basis)
    nc = int(np.ceil((n_atoms / per) ** (1.0 / 3.0)))
    ijk = np.stack(np.meshgrid(np.arange(nc), np.arange(nc), np.arange(nc), indexing="ij"), -1).reshape(-1, 3)
    sites = (ijk[:, None, :] + np.asarray(basis)[None, :, :]).reshape(-1, 3)[:n_atoms] * a
    return sites.astype(dtype), (np.eye(3) * nc * a).astype(dtype), np.tile(np.arange(per), len(ijk))[:n_atoms]


def ref_nlist(device):
    """benchmarks/neighborlist/benchmark_neighborlist.py (cell_list rows): perfect FCC a = 4 A, rc = 5 A, fp32, outputs and cache
    pre-allocated, M = estimate_max_neighbors(5, 0.35, safety_factor=1.0), 10 warm-up + 100 timed calls, median."""
    from nvalchemiops.neighborlist import neighbor_list
    from nvalchemiops.neighborlist.neighbor_utils import allocate_cell_list, estimate_max_neighbors
    from nvalchemiops.neighborlist.cell_list import estimate_cell_list_sizes

    h100 = {32768: 0.878, 131072: 6.71, 262144: 9.82, 524288: 18.44}
    rows = []
    for n in (32768, 131072, 262144, 524288):
        pos, cell, _ = _lattice(n, [[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]], 4.0, np.float32)
        tp, tc = torch.as_tensor(pos, device=device), torch.as_tensor(cell, device=device).reshape(1, 3, 3)
        pbc = torch.ones((1, 3), dtype=torch.bool, device=device)
        m = estimate_max_neighbors(5.0, atomic_density=0.35, safety_factor=1.0)
        nm = torch.full((n, m), n, dtype=torch.int32, device=device)
        sh = torch.zeros((n, m, 3), dtype=torch.int32, device=device)
        num = torch.zeros(n, dtype=torch.int32, device=device)
        max_cells, radius = estimate_cell_list_sizes(tc, pbc, 5.0)
        cache = allocate_cell_list(n, max_cells, radius, device)
        kw = dict(zip(("cells_per_dimension", "neighbor_search_radius", "atom_periodic_shifts", "atom_to_cell_mapping", "atoms_per_cell_count",
                       "cell_atom_start_indices", "cell_atom_list"), cache))
        fn = lambda: neighbor_list(tp, 5.0, cell=tc, pbc=pbc, method="cell_list", neighbor_matrix=nm, neighbor_matrix_shifts=sh,  # noqa: E731
                                   num_neighbors=num, **kw)
        med, lo, hi = _median_ms(fn, 10, 100)
        pairs = int(num.sum().item())
        rows.append({"atoms": n, "max_neighbors": m, "directed_pairs": pairs, "median_ms": med, "min_ms": lo, "max_ms": hi,
                     "atoms_per_s": n / med * 1e3, "reference_h100_median_ms": h100[n], "speedup_vs_reference_h100": h100[n] / med})
    return {"metric": "cell_list neighbor_list median ms (reference benchmark configuration)", "unit": "ms", "rows": rows,
            "source": "BASELINE.md / docs/benchmarks/benchmark_results/neighbor_list_benchmark_cell-list_h100-80gb-hbm3.csv"}


def ref_d3(device):
    """benchmarks/interactions/dispersion/benchmark_dftd3.py: CsCl a = 4.14 A supercells of 30^3 and 35^3 cells (54 000 / 85 750 atoms),
    list = cell_list(rc 21.2 A, max_neighbors 1200) on the Angstrom coordinates, dftd3 on the Bohr coordinates with the padded matrix only
    (no cell / shifts: the benchmark passes none), PBE-BJ parameters with the 35/40 smoothing window, the script's simplified Cs/Cl
    tables (data values, :103-130), fp32, 3 warm-up + 10 timed calls, median.  The list build is not timed (as in the reference)."""
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
    from nvalchemiops.neighborlist import neighbor_list

    rcov, r4r2 = torch.zeros(56), torch.zeros(56)
    rcov[17], rcov[55], r4r2[17], r4r2[55] = 1.88, 4.91, 8.0, 18.0
    c6ab, cn_ref = torch.zeros(56, 56, 5, 5), torch.zeros(56, 56, 5, 5)
    c6ab[17, 17], c6ab[17, 55], c6ab[55, 17], c6ab[55, 55] = 50.0, 200.0, 200.0, 800.0
    cn_ref[:, :] = (torch.arange(5.0) * 0.5)[:, None].expand(5, 5)
    params = D3Parameters(rcov=rcov.to(device), r4r2=r4r2.to(device), c6ab=c6ab.to(device), cn_ref=cn_ref.to(device))
    h100 = {54000: 11.67, 85750: 16.45}
    rows = []
    for size in (30, 35):
        n = 2 * size ** 3
        pos, cell, which = _lattice(n, [[0, 0, 0], [.5, .5, .5]], 4.14, np.float32)
        numbers = torch.as_tensor(np.where(which == 0, 55, 17).astype(np.int32), device=device)
        coord, tc = torch.as_tensor(pos, device=device), torch.as_tensor(cell, device=device)
        pbc = torch.tensor([True, True, True], device=device)
        nm, num, _ = neighbor_list(coord, 21.2, cell=tc, pbc=pbc, method="cell_list", max_neighbors=1200)
        positions = coord * BOHR
        fn = lambda: dftd3(positions=positions, numbers=numbers, d3_params=params, neighbor_matrix=nm, fill_value=n, a1=0.4289, a2=4.4407,  # noqa: E731
                           s6=1.0, s8=0.7875, k1=16.0, k3=-4.0, s5_smoothing_on=35.0, s5_smoothing_off=40.0)
        med, lo, hi = _median_ms(fn, 3, 10)
        pairs = int(num.sum().item())
        rows.append({"atoms": n, "max_neighbors": 1200, "directed_pairs": pairs, "median_ms": med, "min_ms": lo, "max_ms": hi,
                     "atoms_per_s": n / med * 1e3, "pairs_per_s": pairs / med * 1e3, "reference_h100_median_ms": h100[n],
                     "speedup_vs_reference_h100": h100[n] / med})
    return {"metric": "dftd3 median ms (reference benchmark configuration, neighbour list excluded)", "unit": "ms", "rows": rows,
            "source": "BASELINE.md / docs/benchmarks/benchmark_results/dftd3_benchmark_nvalchemiops_h100-80gb-hbm3.csv"}


def ref_pme(device):
    """benchmarks/interactions/electrostatics/benchmark_electrostatics.py, single-system rows: BCC ("CsCl") a = 4.14 A, +-1 charges,
    fp32, estimate_pme_parameters(accuracy 1e-6), spline order 4, reciprocal space only, energies only, k-vectors precomputed,
    3 warm-up + 10 timed calls, median."""
    from nvalchemiops.interactions.electrostatics import estimate_pme_parameters, generate_k_vectors_pme, pme_reciprocal_space

    h100 = {54000: 0.79, 85750: 0.808}
    rows = []
    for size in (30, 35):
        n = 2 * size ** 3
        pos, cell, which = _lattice(n, [[0, 0, 0], [.5, .5, .5]], 4.14, np.float32)
        tp, tc = torch.as_tensor(pos, device=device), torch.as_tensor(cell, device=device)
        q = torch.as_tensor(np.where(which == 0, 1.0, -1.0).astype(np.float32), device=device)
        pp = estimate_pme_parameters(tp, tc, accuracy=1e-6)
        mesh = tuple(int(v) for v in pp.mesh_dimensions)
        kv, k2 = generate_k_vectors_pme(tc, mesh)
        fn = lambda: pme_reciprocal_space(positions=tp, charges=q, cell=tc, alpha=pp.alpha, mesh_dimensions=mesh, spline_order=4,  # noqa: E731
                                          compute_forces=False, k_vectors=kv, k_squared=k2)
        if not rows:
            # process warm-up ahead of the first row only (code-object load, plan creation + self-test, allocator, clocks of a GPU that was
            # idle while this child started): the reference's benchmark script has run other configurations in the same process by the
            # time it reaches these rows; without it the first row's 10-call median is bimodal (0.23 / 0.45 ms in one run)
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        med, lo, hi = _median_ms(fn, 3, 10)
        fused = lambda: pme_reciprocal_space(positions=tp, charges=q, cell=tc, alpha=pp.alpha, mesh_dimensions=mesh, spline_order=4,  # noqa: E731
                                             compute_forces=False)
        med_f, _, _ = _median_ms(fused, 3, 10)
        rows.append({"atoms": n, "mesh": list(mesh), "alpha": float(torch.as_tensor(pp.alpha).reshape(-1)[0]), "median_ms": med, "min_ms": lo,
                     "max_ms": hi, "median_ms_without_k_arrays": med_f, "atoms_per_s": n / med * 1e3, "reference_h100_median_ms": h100[n],
                     "speedup_vs_reference_h100": h100[n] / med})
    return {"metric": "pme_reciprocal_space median ms (reference benchmark configuration: energies only, order 4, fp32)", "unit": "ms", "rows": rows,
            "source": "BASELINE.md / docs/benchmarks/benchmark_results/electrostatics_benchmark_pme_nvalchemiops_h100-80gb-hbm3.csv"}


# ---- the other BASELINE.json configurations as driver-parseable lines ---------------------------------------------
def _timed_steps(step, steps, warmup):
    """The contract's timing of a single-GPU workload, in the headline's lean form (round 6): `warmup` untimed steps; an INSTRUMENTED pass
    (every library kernel bracketed by HIP events on its own stream + one event per step: the per-kernel table and `stats`; an event record
    is a packet of its own, ~5 us of bubble on either side of a kernel, ~13 brackets per step of config 4 = 0.06 of its 0.80 ms); then THE
    timed region: exactly `steps` steps between two synchronisations, wall clock, with the dominant kernel's launches the only bracketed ones.
    Returns (last outputs, elapsed seconds of the timed region, per-step ms of the instrumented pass, kernel records, names timed in-region)."""
    from nvalchemiops import _capi as C

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    C.lib().mi_timing_select(None)
    C.lib().mi_timing_enable(1)
    ev = []
    for _ in range(max(5, min(steps, 20))):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append(e)
        out = step()
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    ev.append(e)
    torch.cuda.synchronize()
    C.lib().mi_timing_enable(0)
    instrumented = kernel_report()
    dominant = max(instrumented.items(), key=lambda kv: kv[1][2])[0] if instrumented else None  # largest median launch duration
    C.lib().mi_timing_select(dominant.encode() if dominant else None)
    C.lib().mi_timing_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    C.lib().mi_timing_enable(0)
    C.lib().mi_timing_select(None)
    timed_only = kernel_report()
    kernels = dict(instrumented)
    kernels.update(timed_only)
    return out, elapsed, [a.elapsed_time(b) for a, b in zip(ev[:-1], ev[1:])], kernels, set(timed_only), len(ev) - 1


def _graph_replay(step, steps):
    """BENCH_GRAPH=1: the same step captured once in a HIP graph (torch.cuda.CUDAGraph: the library launches on torch's current stream, which
    is the capturing one) and replayed `steps` times -- what the eager number pays in launch gaps.  Reported beside `ms_per_step`, never as it."""
    try:
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()  # shapes / tables / plans exist before the capture
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                step()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        return {"captured": True, "ms_per_step": (time.perf_counter() - t0) / steps * 1e3, "steps": steps}
    except Exception as exc:  # a host read inside the step (an allocation size, an overflow check) cannot be captured
        return {"captured": False, "error": f"{type(exc).__name__}: {exc}"[:300]}


def _config_rows(kernels, acct, workload=None):
    """Kernel table of a config workload: every timed kernel with its in-step average and, where `acct` prices it, SURVEY 8(d) bytes;
    `traffic_bytes` from the committed PMC summary of the same workload (profiles/r05_pmc_traffic_<workload>.json) when there is one."""
    rows = {}
    tfile = next((p for p in (os.path.join(ROOT, "profiles", f"{r}_pmc_traffic_{workload}.json") for r in ("r06", "r05")) if os.path.exists(p)), "") if workload else ""
    traffic = {}
    if tfile and os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get("kernels", {})
        except Exception:
            traffic = {}
    for name, (cnt, tot, med, lo, hi) in sorted(kernels.items()):
        bound, algo, note = acct.get(name, ("latency", None, ""))
        row = {"launches": cnt, "avg_ms_timed_region": tot / cnt, "median_ms_timed_region": med, "isolated_median_ms": None, "bound": bound,
               "algorithmic_bytes": algo, "design_bytes": None, "traffic_bytes": None, "traffic_from_profile": None}
        if algo:
            row["algorithmic_GBps"] = algo / (tot / cnt * 1e-3) / 1e9
            row["frac_of_hbm_peak"] = row["algorithmic_GBps"] / HBM_PEAK_GBS
        if note:
            row["note"] = note
        if name in traffic and traffic[name].get("hbm_bytes_per_launch") is not None:
            row["traffic_bytes"], row["traffic_from_profile"] = traffic[name]["hbm_bytes_per_launch"], os.path.relpath(tfile, ROOT)
        rows[name] = row
    return rows


CPU_BUDGET_S = float(os.environ.get("BENCH_CPU_BUDGET_S", "10"))  # CPU-oracle legs of the config workloads (at least one pass each)


def _bounded(fn, budget_s, max_reps=20):
    """Repeat fn() for about budget_s seconds (at least once): (reps, seconds)."""
    reps, t0 = 0, time.perf_counter()
    while True:
        fn()
        reps += 1
        if time.perf_counter() - t0 >= budget_s or reps >= max_reps:
            return reps, time.perf_counter() - t0


def config_c2(device, args):
    """BASELINE config 2: 50 000-atom periodic box (jittered FCC, a = 4 A), `cell_list` rc = 5 A, fp32, padded-matrix output into
    pre-allocated buffers at the API's default row width M = estimate_max_neighbors(5 A) = 928 (what `cell_list` allocates when the caller
    names no width; the reference benchmark's protocol, benchmarks/neighborlist/benchmark_config.yaml + utils.py:133-240, pre-allocates too)."""
    from nvalchemiops.neighborlist import cell_list, estimate_max_neighbors
    from tests import systems as S

    n, rc = 50000, 5.0
    pos, cell, _, _ = S.fcc_box(n, seed=1234, dtype=np.float32)
    tp, tc = torch.as_tensor(pos, device=device), torch.as_tensor(cell, device=device)
    pbc = torch.tensor([True] * 3, device=device)
    m = estimate_max_neighbors(rc)
    nm, sh, num = list_buffers(tp, rc, tc, pbc, m, device, report=BUFFER_REPORT)

    def step():
        cell_list(tp, rc, tc, pbc, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
        return num

    acct = {"nl_query_matrix_f32": ("hbm", n * (3 * 4 + 4) + 16.0 * n * m, "owner-written rows: hits, then padding (fill value N, zero shifts)")}

    def cpu():
        from oracle import oracle as O

        reps, sec = _bounded(lambda: O.cell_list(pos, rc, cell, [True] * 3, max_neighbors=m), CPU_BUDGET_S)
        return {"value": n * reps / sec, "unit": "atom-steps/s", "cores": 1, "kind": "port", "seconds": sec,
                "sample": f"{reps} full-size passes of the oracle's cell_list restatement on the same 50 000-atom box (serial, 1 thread)"}

    what = f"config 2: {n}-atom periodic FCC box, cell_list rc = 5 A, fp32, padded matrix M = {m} (default width) into pre-allocated outputs"
    return dict(step=step, atoms=n, acct=acct, cpu=cpu, workload=what, dtype="f32", extra=lambda out: {"directed_pairs": int(out.sum().item()),
                                                                                                      "neighbors_max": int(out.max().item())})


def config_c3(device, args):
    """BASELINE config 3: batch of 256 x 512-atom molecules (free space: every molecule in its own non-periodic bounding cell),
    batch `neighbor_list` rc = 40 Bohr (padded matrix M = 512 + zero shifts) + `dftd3`(BJ) energies / forces / virials, fp32."""
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
    from nvalchemiops.neighborlist import batch_cell_list
    from tests import systems as S

    nmol, per, rc, m = 256, 512, 40.0, 512
    mols = [S.molecule(per, seed=2000 + s) for s in range(4)]
    pos = np.concatenate([(mols[s % 4][0] * BOHR).astype(np.float32) for s in range(nmol)])
    z = np.concatenate([mols[s % 4][1] for s in range(nmol)])
    side = float(max(mm[2] for mm in mols)) * BOHR + 1.0
    cell = np.broadcast_to(np.eye(3, dtype=np.float32) * side, (nmol, 3, 3)).copy()
    bi = np.repeat(np.arange(nmol, dtype=np.int32), per)
    n = nmol * per
    tables = S.d3_test_tables(94, seed=7)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)  # noqa: E731
    params = D3Parameters(rcov=t(tables["rcov"]), r4r2=t(tables["r4r2"]), c6ab=t(tables["c6ab"]), cn_ref=t(tables["cn_ref"]))
    tp, tz, tb, tc = t(pos), t(z), t(bi), t(cell)
    pbc = torch.zeros((nmol, 3), dtype=torch.bool, device=device)
    nm, sh, num = list_buffers(tp, rc, tc, pbc, m, device, batch_idx=tb, for_dftd3=d3_search_context(tz, params), report=BUFFER_REPORT)
    bj = dict(a1=D3["a1"], a2=D3["a2"], s8=D3["s8"])

    def step():
        batch_cell_list(tp, rc, tc, pbc, tb, neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
        e, f, cn, vir = dftd3(tp, tz, d3_params=params, neighbor_matrix=nm, neighbor_matrix_shifts=sh, cell=tc, batch_idx=tb, num_systems=nmol,
                              fill_value=n, compute_virial=True, **bj)
        return num, e, f, vir

    per_pass = 16.0 * n * m + 40.0 * n
    search_cn = os.environ.get("BENCH_SEARCH_CN", "1") != "0"  # the batch search also sums the coordination numbers
    acct = {"nl_query_matrix_f32": ("hbm", n * (3 * 4 + 4) + 16.0 * n * m + (per_pass if search_cn else 0.0),
                                    "search + the D3 CN pass's sums in one launch: both SURVEY 8(d) rows" if search_cn else ""),
            "d3_cn": (("latency", None, "adopts the search's coordination numbers (fingerprint check + N-float copy); no list walk") if search_cn else
                      ("hbm", per_pass, "streams 16 B/slot + one 16 B gather per neighbour")),
            "d3_energy": ("valu", per_pass, "C6 contraction + BJ damping per directed pair (VALU-issue-bound, DESIGN 3.1)"),
            "d3_chain": ("hbm", per_pass, "")}

    def cpu():
        from oracle import oracle as O

        k = 8  # bounded sample: the first 8 molecules (the batch is 64 copies of 4 molecules: identical per-atom work)
        sp, sz, sb, sc = pos[:k * per], z[:k * per], bi[:k * per], cell[:k]

        def one():
            onm, onum, osh = O.cell_list(sp, rc, sc, np.zeros((k, 3), bool), batch_idx=sb, max_neighbors=m)
            O.dftd3(sp, sz, tables, D3["a1"], D3["a2"], D3["s8"], neighbor_matrix=onm, neighbor_matrix_shifts=osh, cell=sc, batch_idx=sb,
                    num_systems=k, compute_virial=True)

        reps, sec = _bounded(one, CPU_BUDGET_S)
        return {"value": k * per * reps / sec, "unit": "atom-steps/s", "cores": 1, "kind": "port", "seconds": sec,
                "sample": f"{reps} passes of the oracle (batch cell list + D3 with virial, serial) over the first {k} of the 256 molecules "
                          "(the batch repeats 4 distinct molecules: per-atom work identical)"}

    what = (f"config 3: {nmol} x {per}-atom molecules (free space, non-periodic cells), batch neighbor_list rc = 40 Bohr (padded M = {m}) + "
            "DFT-D3(BJ) E + F + virial, fp32")
    return dict(step=step, atoms=n, acct=acct, cpu=cpu, workload=what, dtype="f32",
                extra=lambda out: {"directed_pairs": int(out[0].sum().item()), "neighbors_max": int(out[0].max().item()),
                                   "e_d3_Ha_first_molecule": float(out[1][0].item())})


def config_c4(device, args):
    """BASELINE config 4: 100 000-atom periodic box with charges, `neighbor_list` rc = 9 A (padded M = 256) + `particle_mesh_ewald`
    (real + reciprocal, alpha 0.35, mesh 128^3, B-spline order 5, energies + forces), fp64 -- the electrostatics branch of the headline step."""
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
    from nvalchemiops.neighborlist import cell_list

    n = 100000
    sysd, _ = build_system(n, 1234, device)
    m, mesh = PME["max_neighbors"], float(np.prod(PME["mesh"]))
    nm, sh, num = list_buffers(sysd["pos64"], PME["cutoff"], sysd["cell64"], sysd["pbc"], m, device, report=BUFFER_REPORT)

    def step():
        cell_list(sysd["pos64"], PME["cutoff"], sysd["cell64"], sysd["pbc"], neighbor_matrix=nm, neighbor_matrix_shifts=sh, num_neighbors=num)
        e, f = particle_mesh_ewald(sysd["pos64"], sysd["q64"], sysd["cell64"], alpha=PME["alpha"], mesh_dimensions=PME["mesh"],
                                   spline_order=PME["order"], neighbor_matrix=nm, neighbor_matrix_shifts=sh, compute_forces=True)
        return num, e, f

    acct = {"nl_query_matrix_f64": ("hbm", n * (3 * 8 + 4) + 16.0 * n * m, ""),
            "ewald_real": ("hbm", 16.0 * n * m + n * (3 * 8 + 8) + n * (8 + 3 * 8), "fp64 erfc pair sum over the padded rows"),
            "spline_spread": ("latency", n * 4 * 8 + mesh * 8, "binning + LDS-tile accumulation"),
            "pme_gather_finish": ("hbm", 4 * mesh * 8 + n * 4 * 8 + n * 4 * 8, "L2/MALL-resident mesh gather"),
            "pme_convolve": ("hbm", (mesh / 2) * 16 * 5, ""),
            "pme_solve_fwd": ("hbm", mesh * 8 + (mesh / 2) * 16, "real planes in, (y,z)-transformed half spectrum out"),
            "pme_solve_cols": ("hbm", (mesh / 2) * 16 * (2 + 1 + 4), "x transform + Green function in place, then 4 channels back along x"),
            "pme_solve_inv": ("hbm", 4 * ((mesh / 2) * 16 + mesh * 8), "4 channels: half spectrum in, real planes out")}
    host = sysd["host"]

    def cpu():
        from oracle import oracle as O

        def one():
            onm, onum, osh = O.cell_list(host["pos"], PME["cutoff"], host["cell"], [True] * 3, max_neighbors=m)
            with O.extended_splines():  # order 5 as true B-splines, like the GPU step (the oracle's reference mode is zero there)
                O.particle_mesh_ewald(host["pos"], host["q"], host["cell"], PME["alpha"], PME["mesh"], PME["order"], neighbor_matrix=onm,
                                      neighbor_matrix_shifts=osh, compute_forces=True)

        reps, sec = _bounded(one, CPU_BUDGET_S, max_reps=3)
        return {"value": n * reps / sec, "unit": "atom-steps/s", "cores": 1, "kind": "port", "seconds": sec,
                "sample": f"{reps} FULL-size step(s) of the oracle (cell list 9 A + PME mesh 128^3 order 5 in its extended mode, E + F), serial, numpy FFTs"}

    what = (f"config 4: {n}-atom periodic FCC box with +-1 charges, nlist(9 A, padded M = {m}) + particle_mesh_ewald(real + reciprocal, alpha 0.35, "
            "mesh 128^3, spline order 5 = true B-spline (the reference evaluates order 5 as 0), E + F), fp64")
    return dict(step=step, atoms=n, acct=acct, cpu=cpu, workload=what, dtype="f64",
                extra=lambda out: {"pme_neighbors_max": int(out[0].max().item()), "e_pme": float(out[1].sum().item())})


def run_config(name, device, args):
    """`--workload c2|c3|c4`: one BASELINE.json configuration as a line of the same shape as the headline's (metric per config: atom-steps/s of
    ONE pass of that configuration's hot path; `roofline` = the dominant kernel of that configuration, in-step; `cpu_baseline` = the oracle)."""
    cfg = {"c2": config_c2, "c3": config_c3, "c4": config_c4}[name](device, args)
    calibration = hbm_calibration(device, gib=1.0) if os.environ.get("BENCH_CALIB", "1") != "0" else None
    out, elapsed, step_ms, kernels, timed_names, n_instr = _timed_steps(cfg["step"], args.steps, args.warmup)
    rows = _config_rows(kernels, cfg["acct"], name)
    for kname, r in rows.items():
        in_timed = kname in timed_names
        r["in_step_source"] = "timed region" if in_timed else "instrumented pass before the timed region (every kernel bracketed)"
        r["launches_per_step"] = r["launches"] / (args.steps if in_timed else n_instr)
    roof = roofline_of({k: v for k, v in rows.items() if v.get("algorithmic_bytes")} or rows)
    if roof and calibration and roof.get("achieved") and roof["bound"] == "hbm":
        roof["frac_of_box_fill"] = roof["achieved"] / calibration["fill_GBps"]
        roof["frac_of_box_copy"] = roof["achieved"] / calibration["copy_GBps"]
    n = cfg["atoms"]
    res = {"metric": f"atom-steps/sec of BASELINE config {name[1]} (one pass of its hot path per step)", "value": n * args.steps / elapsed,
           "unit": "atom-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
           "config": {"workload": cfg["workload"], "atoms_per_gpu": n, **cfg["extra"](out),
                      "list_buffers": ({"selection": "tuned_neighbor_buffers (fastest candidate by a trial search at set-up, untimed)", **BUFFER_REPORT}
                                       if BUFFER_REPORT else "torch.empty")},
           "stats": {"step_ms_median": statistics.median(step_ms), "step_ms_min": min(step_ms), "step_ms_max": max(step_ms), "timed_region_s": elapsed,
                     "step_ms_source": "instrumented pass before the timed region (one event per step, every kernel bracketed)"},
           "roofline": roof, "calibration": calibration, "kernels": rows}
    if os.environ.get("BENCH_GRAPH", "0") == "1":
        res["hip_graph"] = _graph_replay(cfg["step"], args.steps)
    if args.cpu_sample > 0:
        res["cpu_baseline"] = cfg["cpu"]()
    print(json.dumps(res), flush=True)


# ---- launcher -------------------------------------------------------------------------------------------------------------------------
def pme_train(device, atoms: int = 100000, iters: int = 20):
    """Training path of the electrostatics leg (reference protocol test_pme.py:1458,1571 / autograd.py:525-665):
    `particle_mesh_ewald` on the headline box (9 A padded list built once, alpha 0.35, mesh 128^3, fp64), forward + backward through the
    registered `alchemiops::*` ops and their hand-written adjoint kernels.  Two losses per spline order (4 = formula-identical to the
    reference, 5 = the headline order):
      energy        L = sum_i E_i                   backward = the forces by autograd (first-order adjoints)
      energy+force  L = sum_i E_i + sum_i W_i.F_i   force matching: backward runs the second-order adjoints
    Reports event-bracketed medians (ms): inference forward (no grad: the fused path), forward under grad, backward, and the adjoint kernels."""
    from nvalchemiops import _capi as C
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
    from nvalchemiops.neighborlist import cell_list
    from tests import systems as S

    pos, cell, q, _ = S.fcc_box(atoms, seed=1234, dtype=np.float64)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)  # noqa: E731
    tp, tc, tq = t(pos), t(cell), t(q)
    pbc = torch.tensor([True] * 3, device=device)
    nm, num, nsh = cell_list(tp, PME["cutoff"], tc, pbc, max_neighbors=PME["max_neighbors"])
    g = torch.Generator(device=device).manual_seed(3)
    W = torch.randn((atoms, 3), dtype=torch.float64, device=device, generator=g)
    rows = []
    for order in (4, 5):
        kw = dict(alpha=PME["alpha"], mesh_dimensions=PME["mesh"], spline_order=order, neighbor_matrix=nm, neighbor_matrix_shifts=nsh)
        infer_e = _median_ms(lambda: particle_mesh_ewald(tp, tq, tc, **kw), 3, iters)[0]
        infer_ef = _median_ms(lambda: particle_mesh_ewald(tp, tq, tc, compute_forces=True, **kw), 3, iters)[0]
        for loss_kind in ("energy", "energy+force"):
            def forward(ev=None):
                p = tp.detach().clone().requires_grad_(True)
                if ev:
                    ev[0].record()  # the forward figure brackets the particle_mesh_ewald call alone, like the inference figure
                if loss_kind == "energy":
                    e = particle_mesh_ewald(p, tq, tc, **kw)
                    if ev:
                        ev[1].record()
                    return p, e.sum()
                e, f = particle_mesh_ewald(p, tq, tc, compute_forces=True, **kw)
                if ev:
                    ev[1].record()
                return p, e.sum() + (W * f).sum()

            for _ in range(3):
                p, loss = forward()
                loss.backward()
            torch.cuda.synchronize()
            C.lib().mi_timing_enable(1)
            fwd, bwd = [], []
            for _ in range(iters):
                a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                p, loss = forward((a, b))
                loss.backward()  # (the loss's own two or three reductions are counted with the backward)
                c.record()
                c.synchronize()
                fwd.append(a.elapsed_time(b))
                bwd.append(b.elapsed_time(c))
            C.lib().mi_timing_enable(0)
            ker = {k: round(v[2], 4) for k, v in kernel_report().items()}
            f_ms, b_ms = statistics.median(fwd), statistics.median(bwd)
            rows.append({"spline_order": order, "loss": loss_kind, "forward_inference_ms": infer_e if loss_kind == "energy" else infer_ef,
                         "forward_ms": f_ms, "backward_ms": b_ms, "backward_over_forward": b_ms / f_ms,
                         "backward_over_inference_forward": b_ms / (infer_e if loss_kind == "energy" else infer_ef),
                         "kernel_median_ms": ker, "grad_norm": float(p.grad.norm().item())})
    return {"metric": "ms per training step of particle_mesh_ewald (forward + backward) on the 100k-atom box", "unit": "ms", "rows": rows,
            "config": {"workload": f"{atoms}-atom periodic FCC box, nlist 9 A (M=256) built once, PME alpha 0.35, mesh 128^3, fp64; orders 4 and 5 "
                                   "(5 = true B-spline: the reference evaluates order 5 as 0)"},
            "dtype": "f64"}


def _free_port() -> int:
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run with N ranks on this node, rank r on GPU r
    over RCCL.  With fewer than N GPUs the ranks share device 0 and use gloo (RCCL refuses two ranks on one device): a functional
    check of the N > 1 path, labelled as such in the JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < n:
        env["BENCH_SHARE_DEVICE"], env["BENCH_BACKEND"] = "1", "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def fresh_processes(args) -> int:
    """The headline line as the MEDIAN of `args.processes` fresh processes.  Every child is this script with
    `--processes 1 --cpu-sample 0`: W warm-up + exactly K timed steps, barrier / synchronize on both sides, one JSON line.  This parent never
    touches the GPU; it prints the child line with the median ms_per_step, adds `processes` (every child's step time, list-fill state and
    calibration) and runs the CPU baseline once."""
    argv = [a for a in sys.argv[1:]]
    for flag in ("--processes", "--cpu-sample"):
        while flag in argv:
            k = argv.index(flag)
            del argv[k:k + 2]
        argv = [a for a in argv if not a.startswith(flag + "=")]
    lines, t_children = [], time.perf_counter()
    for k in range(args.processes):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv + ["--processes", "1", "--cpu-sample", "0"], capture_output=True, text=True)
        out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not out:
            sys.stderr.write(r.stderr[-4000:])
            return r.returncode or 1
        lines.append(json.loads(out[-1]))
    order = sorted(range(len(lines)), key=lambda k: lines[k]["ms_per_step"])
    pick = order[len(order) // 2]
    res = lines[pick]

    def state(d):
        k = d["kernels"].get("nl_query_matrix_f32", {})
        return {"ms_per_step": d["ms_per_step"], "value": d["value"], "list_fill_40bohr_isolated_median_ms": k.get("isolated_median_ms"),
                "list_fill_40bohr_in_step_ms": k.get("avg_ms_timed_region"), "d3_cn_in_step_ms": d["kernels"].get("d3_cn", {}).get("avg_ms_timed_region"),
                "calibration_fill_GBps": (d.get("calibration") or {}).get("fill_GBps")}

    res["processes"] = {"count": len(lines), "reported": f"process {pick + 1} of {len(lines)} (median ms_per_step)", "seconds_all": time.perf_counter() - t_children,
                        "each": [state(d) for d in lines],
                        "note": "fresh processes run one after the other; the list fill is in one of two states for the life of a process "
                                "(placement of its row buffers by the driver: DESIGN.md 3.3), so the line is the median process, not a coin flip"}
    if args.cpu_sample > 0:
        res["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_full_size)
        add_parity(res)
    if args.workload == "headline" and os.environ.get("BENCH_CONFIGS", "1") != "0":
        res["configs"] = compact_configs()
    print(json.dumps(res), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250, help="timed steps (default 250: > 1 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--atoms", type=int, default=100000)
    ap.add_argument("--cpu-sample", type=int, default=6912, help="atoms in the CPU-baseline sample box (0 = skip the CPU baseline)")
    ap.add_argument("--cpu-full-size", action="store_true", help="also time ONE full 100k-atom step of the serial oracle (~1 min)")
    ap.add_argument("--workload", default="headline", choices=["headline", "c2", "c3", "c4", "c5", "ref-nlist", "ref-d3", "ref-pme", "pme-train"],
                    help="headline: one 100k-atom box per GPU (default, the BASELINE metric); c5: BASELINE config 5, --systems x 2000-atom "
                         "boxes per GPU sharded at system granularity; ref-*: the reference's published benchmark rows (BASELINE.md)")
    ap.add_argument("--systems", type=int, default=128, help="systems per GPU for --workload c5")
    ap.add_argument("--no-virial", action="store_true", help="experiment switch: D3 without the virial (the headline includes it)")
    ap.add_argument("--d3-format", default="matrix", choices=["matrix", "csr"],
                    help="neighbour-list format of the D3 leg: padded matrix with explicit row width (default; what the reference's own D3 benchmark "
                         "uses) or exact-size COO/CSR (two-pass build)")
    ap.add_argument("--overlap", type=int, default=1, choices=[0, 1, 2, 3, 4, 5], help="1 (default): PME and D3 branches on two HIP streams; 0: one stream, per-stage times; 2 - 4: tuning aids; 5: three streams (reciprocal half, real-space half, dispersion)")
    ap.add_argument("--processes", type=int, default=0,
                    help="headline, 1 GPU: run the timed region in this many FRESH processes one after the other and report the one with the median "
                         "ms_per_step (default 3; 1 = time in this process).  The 40-Bohr list fill has two states that are fixed for the life of a "
                         "process (0.82 / 1.13 ms: a property of where the driver places the row buffers, DESIGN.md 3.3): one process is a coin flip")
    args = ap.parse_args()
    global VIRIAL, OVERLAP, D3_FORMAT
    VIRIAL = not args.no_virial
    D3_FORMAT = args.d3_format
    OVERLAP = int(args.overlap)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.processes == 0:
        args.processes = 3 if (args.workload == "headline" and args.gpus == 1 and "WORLD_SIZE" not in os.environ
                               and os.environ.get("BENCH_GRAPH") != "1") else 1
    if args.processes > 1:
        sys.exit(fresh_processes(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    shared = os.environ.get("BENCH_SHARE_DEVICE") == "1"  # all ranks on cuda:0 (gloo): N > 1 smoke test on a 1-GPU box
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if os.environ.get("BENCH_MAIN_PRIORITY"):  # tuning aid: run the main (dispersion) branch on a prioritised HIP stream
        torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ["BENCH_MAIN_PRIORITY"])))
    device = torch.device("cuda", local_rank)
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        kw = {"device_id": device} if backend == "nccl" else {}
        torch.distributed.init_process_group(backend=backend, **kw)
        # self-validating scaling record: a real N-GPU run is RCCL ("nccl") over N DISTINCT devices (asserted, and written into the line)
        ident = [None] * world
        props = torch.cuda.get_device_properties(device)
        torch.distributed.all_gather_object(ident, (local_rank, str(getattr(props, "uuid", "")), props.name))
        if not shared:
            assert torch.distributed.get_backend() == "nccl", f"multi-GPU bench must run over RCCL, got {torch.distributed.get_backend()}"
            assert len({(i[0], i[1]) for i in ident}) == world, f"{world} ranks must sit on {world} distinct devices, saw {ident}"

    from nvalchemiops import _capi as C

    if args.workload.startswith("ref-") or args.workload == "pme-train":
        if world > 1:
            raise SystemExit("ref-* / pme-train workloads are single-GPU (the reference's benchmarks are)")
        res = {"ref-nlist": ref_nlist, "ref-d3": ref_d3, "ref-pme": ref_pme, "pme-train": pme_train}[args.workload](device)
        res.update({"n_gpus": 1, "data": "synthetic", "higher_is_better": False, "workload": args.workload,
                    "protocol": "median of event-bracketed calls after warm-up (reference: benchmarks/utils.py:133-240)"})
        print(json.dumps(res), flush=True)
        return

    if args.workload in ("c2", "c3", "c4"):
        if world > 1:
            raise SystemExit("c2 / c3 / c4 are BASELINE.json's single-GPU configurations")
        if args.steps == 250:
            args.steps = 100
        run_config(args.workload, device, args)
        return

    if args.workload == "c5":
        # BASELINE config 5: world x --systems periodic 2000-atom boxes, sharded at system granularity (distributed.shard_batch);
        # weak scaling: --systems per GPU (128 = one eighth of the 1024-system batch)
        from nvalchemiops.distributed import partition_systems

        args.atoms = args.systems * 2000
        total = args.systems * world
        s0, s1 = partition_systems([2000] * total, world)[rank]
        sysd, tables = build_batch(s1 - s0, 2000, 1234, device, first_system=s0)
        step, _ = make_batch_step(sysd, tables, device, world, [b - a for a, b in partition_systems([2000] * total, world)])
        args.cpu_sample = 0
    else:
        sysd, tables = build_system(args.atoms, 1234 + rank, device)
        step, _ = make_step(sysd, tables, device, world)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # this box's own HBM rates, measured before the timed region (not part of it)
    calibration = hbm_calibration(device, gib=float(os.environ.get("BENCH_CALIB_GIB", "4"))) if os.environ.get("BENCH_CALIB", "1") != "0" else None
    for _ in range(args.warmup):
        out = step()
    barrier()
    global COMPANION, SEARCH_CN
    from nvalchemiops.neighborlist import _engine as NE

    bufs = getattr(step, "d3_bufs", None)
    COMPANION = bool(bufs) and hasattr(bufs[0], NE._PACKED_ATTR)
    SEARCH_CN = COMPANION and getattr(bufs[0], NE._PACKED_ATTR).cn is not None
    records = []
    step_events = []
    graph_mode = os.environ.get("BENCH_GRAPH") == "1"
    if graph_mode:
        # tuning aid, not the reported mode: the whole step (both streams) captured once into a hipGraph and replayed.  Per-kernel
        # HIP-event timing is impossible inside a graph, so `roofline` cannot be measured live in this mode.
        graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            out = step()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=cap):
                out = step()
        torch.cuda.current_stream().wait_stream(cap)
        graph.replay()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        barrier()
        elapsed = time.perf_counter() - t0
    else:
        # (1) instrumented pass, NOT the timed region: the same two-stream step with every kernel bracketed by HIP events (the per-kernel
        # in-step table) and one event per step (`stats`).  An event record is a packet of its own in the stream -- ~5 us of bubble on either
        # side of a kernel, ~25 brackets per step -- so the timed region below carries ONE bracket, the dominant kernel's (round 6).
        C.lib().mi_timing_select(None)
        C.lib().mi_timing_enable(1)
        for _ in range(max(5, min(args.steps, 50))):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            step_events.append(e)
            out = step(records if len(records) < 32 else None)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        step_events.append(e)
        barrier()
        C.lib().mi_timing_enable(0)
        instrumented = kernel_report()
        # (2) which kernel is the dominant one?  the one with the most work: largest median duration in a short serialised pass
        saved, OVERLAP = OVERLAP, 0
        C.lib().mi_timing_enable(1)
        for _ in range(3):
            step()
        barrier()
        C.lib().mi_timing_enable(0)
        OVERLAP = saved
        probe = kernel_report()
        dominant = max(probe.items(), key=lambda kv: kv[1][2])[0] if probe else None
        # (3) THE timed region: exactly `steps` steps between two barriers, wall clock; the dominant kernel's launches are the only ones
        # bracketed (its average duration inside this region is `roofline.achieved`'s denominator)
        C.lib().mi_timing_select(dominant.encode() if dominant else None)
        C.lib().mi_timing_enable(1)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        elapsed = time.perf_counter() - t0
        C.lib().mi_timing_enable(0)
        C.lib().mi_timing_select(None)
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        tt = mine.clone()
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernels = kernel_report()  # the timed region's records: the dominant kernel only
    if not graph_mode:
        timed_only = kernels
        kernels = dict(instrumented)
        kernels.update(timed_only)  # the dominant kernel's row carries its launches INSIDE the timed region
    step_ms = [a.elapsed_time(b) for a, b in zip(step_events[:-1], step_events[1:])]
    # Untimed extra pass: the same step with the two branches serialised, to time every kernel in isolation.  In the timed region
    # the bandwidth-bound kernels run beside the other stream's work, so their HIP-event durations include that contention (and
    # move when a profiler changes the overlap); the isolated figures are the ones comparable across runs and with
    # profiles/*kernel_stats*_serial.csv.  They do not enter `value`.
    isolated, serial_ms = {}, []
    if not graph_mode:
        saved, OVERLAP = OVERLAP, 0
        C.lib().mi_timing_enable(1)
        sev = []
        for _ in range(max(5, min(args.steps, 20))):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            sev.append(e)
            step()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        sev.append(e)
        barrier()
        C.lib().mi_timing_enable(0)
        isolated = kernel_report()
        serial_ms = [a.elapsed_time(b) for a, b in zip(sev[:-1], sev[1:])]
        OVERLAP = saved

    e_pme, f_pme, e_d3, f_d3, num, nptr = out
    matrix_d3 = D3_FORMAT == "matrix"
    pairs_d3 = int(nptr.sum().item()) if matrix_d3 else int(nptr[-1].item())  # matrix format: `nptr` holds num_neighbors
    if matrix_d3 and int(nptr.max().item()) > D3["max_neighbors"]:
        raise RuntimeError(f"D3 neighbour matrix overflow: {int(nptr.max().item())} > {D3['max_neighbors']}")
    stage_ms, gather_us = {}, []
    for ev in records:
        for (_, a), (name, b) in zip(ev[:-1], ev[1:]):
            stage_ms[name] = stage_ms.get(name, 0.0) + a.elapsed_time(b) / len(records)
            if name == "gather":
                gather_us.append(a.elapsed_time(b) * 1e3)

    if rank == 0:
        total_atoms = args.atoms * world
        value = total_atoms * args.steps / elapsed
        rows = kernel_table(kernels, isolated, args.atoms, pairs_d3, args.workload)
        n_instr = max(len(step_events) - 1, 1)
        for name, r in rows.items():  # where each row's in-step figures come from, and its launches per step
            in_timed = graph_mode or name in timed_only
            r["in_step_source"] = "timed region" if in_timed else "instrumented pass before the timed region (every kernel bracketed)"
            r["launches_per_step"] = r["launches"] / (args.steps if in_timed else n_instr)
        par = "single GPU"
        if world > 1:
            par = ("replica box" if args.workload == "headline" else "system-granular shard") + f" per rank, {world} ranks, 1 all_gather of per-system energies per step over "
            par += "RCCL" if backend == "nccl" else f"{backend} (ranks share ONE device: functional check of the N > 1 path, not a scaling number)"
        result = {
            "metric": "atom-steps/sec (nlist+D3+PME) on 100k-atom PBC box",
            "value": value, "unit": "atom-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (PME) + f32 (D3)", "data": "synthetic",
            "config": {"workload": (f"{args.atoms}-atom periodic FCC box per GPU: nlist(9 A, padded M=256) + PME(alpha 0.35, mesh 128^3, "
                                    "spline order 5 = true B-spline (the reference evaluates order 5 as 0), E+F, fp64) + nlist(40 Bohr, " + (f"padded M={D3['max_neighbors']}" if D3_FORMAT == "matrix" else "CSR") + ") + DFT-D3(BJ, E+F+virial, fp32)")
                       if args.workload == "headline" else
                       (f"config 5: {args.systems} x 2000-atom periodic boxes per GPU (batch): nlist(9 A) + PME(mesh 32^3 per system, order 5, "
                        "fp64) + nlist(40 Bohr, " + (f"padded M={D3['max_neighbors']}" if D3_FORMAT == "matrix" else "CSR") + ") + DFT-D3(BJ), one all_gather of per-system energies"),
                       "atoms_per_gpu": args.atoms, "d3_directed_pairs": pairs_d3, "pme_neighbors_max": int(num.max().item()), "d3_neighbors_max": int((nptr if matrix_d3 else (nptr[1:] - nptr[:-1])).max().item()),
                       "parallelism": par, "ranks": world, "backend": backend,
                       "devices": ([f"{i[2]} #{i[0]} {i[1]}" for i in ident] if world > 1 else [torch.cuda.get_device_name(device)]),
                       "real_space_divide": "hardware rcp/rsq + Newton (EW_IEEE_DIV=0 build of csrc/ewald.hip)"},
            "stats": ({"step_ms_median": statistics.median(step_ms), "step_ms_min": min(step_ms), "step_ms_max": max(step_ms),
                       "value_at_median": total_atoms / statistics.median(step_ms) * 1e3, "timed_region_s": elapsed,
                       "overlap": OVERLAP,
                       "step_ms_median_serial_untimed": statistics.median(serial_ms) if serial_ms else None,
                       "note": "per-step GPU time between HIP events on rank 0's main stream in the INSTRUMENTED pass that precedes the timed region (every kernel "
                               "bracketed: ~0.1 ms of event bubbles per step); `value` is the wall-clock figure over exactly `steps` steps with one bracket per step"}
                      if step_ms else None),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "ranks": {"ms_per_step": [round(x, 4) for x in rank_ms], "imbalance_max_over_min": max(rank_ms) / max(min(rank_ms), 1e-9),
                      "all_gather_us_median": statistics.median(gather_us) if gather_us else None,
                      "note": "wall-clock per step of every rank over the timed region (`value` uses the slowest); the all_gather figure is rank 0's "
                              "GPU time between the events around the one collective of a step"},
            "energies": {"e_d3_Ha": float(e_d3[0].item()), "e_pme": float(e_pme.sum().item())},
            "roofline": roofline_of(rows),
            "calibration": calibration,
            "kernels": rows,
        }
        result["config"]["d3_list_buffers"] = (
            {"selection": "nvalchemiops.neighborlist.tuned_neighbor_buffers: fastest of the candidate allocations by a trial search at set-up (untimed)",
             **BUFFER_REPORT} if BUFFER_REPORT else "torch.empty")
        result["config"]["d3_list_companion"] = (
            "on: the 40-Bohr search also writes a 4 B/slot packed companion (policy 'auto', learned from the warm-up's first dftd3 call), "
            "all three D3 passes stream it; outputs bit-identical to the plain path" if COMPANION else "off")
        result["config"]["d3_search_cn"] = (
            "on: the 40-Bohr search also sums the DFT-D3 coordination numbers over its hits (D3SearchContext attached to the list buffers at set-up); "
            "dftd3 adopts them after a device-side fingerprint check of positions / species / cell / k1 and skips its CN pass" if SEARCH_CN else "off")
        # bytes the step's timed kernels really move per step (design bytes where they differ from the 8(d) formula, else the formula's)
        # against the ~6.3 TB/s the guide gives as achievable: the whole step's distance from an HBM floor, not one kernel's
        moved = sum((r.get("design_bytes") or r.get("algorithmic_bytes") or 0.0) * r["launches_per_step"] for r in rows.values())
        step_s = elapsed / args.steps
        result["step_traffic"] = {"moved_bytes_per_step": moved, "moved_TBps": moved / step_s / 1e12, "frac_of_achievable_6p3_TBps": moved / step_s / 6.3e12,
                                  "frac_of_hbm_peak": moved / step_s / (HBM_PEAK_GBS * 1e9),
                                  "floor_ms_at_6p3_TBps": moved / 6.3e12 * 1e3,
                                  "note": "sum over the timed kernels of the bytes each streams per launch (design bytes) x launches per step; the VALU-bound "
                                          "energy pass (0.9 ms for 1 GB) is why the step cannot sit on this floor"}
        if calibration and result["roofline"] and result["roofline"].get("bound") == "hbm" and result["roofline"].get("achieved"):
            # the same achieved figure against what a plain copy reaches on THIS box (read + written bytes per second)
            result["roofline"]["frac_of_box_copy"] = result["roofline"]["achieved"] / calibration["copy_GBps"]
            result["roofline"]["frac_of_box_fill"] = result["roofline"]["achieved"] / calibration["fill_GBps"]
            result["roofline"]["frac_of_box_read"] = result["roofline"]["achieved"] / calibration["read_GBps"]
            if result["roofline"].get("design_bytes_per_launch") and result["roofline"].get("launch_ms"):
                # where the kernel deliberately moves other bytes than the 8(d) formula's (the CN pass also writes the packed copy): the
                # bytes it moves per second against the same box figures
                moved = result["roofline"]["design_bytes_per_launch"] / (result["roofline"]["launch_ms"] * 1e-3) / 1e9
                result["roofline"]["moved_GBps"] = moved
                result["roofline"]["moved_frac_of_box_copy"] = moved / calibration["copy_GBps"]
                result["roofline"]["moved_frac_of_box_fill"] = moved / calibration["fill_GBps"]  # the list kernels only write
        if world == 1 and args.cpu_sample > 0:
            result["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_full_size)
            add_parity(result)
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""N > 1 path ON KERNELS: two ranks share the one device of the test box (gloo -- RCCL refuses two ranks on one GPU),
each runs the full batch step (batch_cell_list -> particle_mesh_ewald -> batch_cell_list 40 Bohr -> dftd3) on its shard of a
periodic batch cut by `distributed.shard_batch`, ONE all_gather of the per-system energies; the gathered [B, 2] table must equal the
single-rank batch run.  Also: `python bench.py --gpus 2` launches its two ranks by itself."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOHR = 1.8897261246


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(dev):
    from tests import systems as S

    counts = [500, 864, 256, 700, 500, 864]  # ragged: the partition is by atom count
    parts = [S.fcc_box(n, seed=50 + b, dtype=np.float64) for b, n in enumerate(counts)]
    t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)  # noqa: E731
    pos = np.concatenate([p[0] for p in parts])
    d = dict(pos=t(pos), cell=t(np.stack([p[1] for p in parts])), q=t(np.concatenate([p[2] for p in parts])),
             numbers=t(np.concatenate([p[3] for p in parts])), pos_b=t((pos * BOHR).astype(np.float32)),
             cell_b=t((np.stack([p[1] for p in parts]) * BOHR).astype(np.float32)),
             ptr=torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32))
    return d, counts, S.d3_test_tables(17)


def _step(pos, cell, q, numbers, pos_b, cell_b, bi, nsys, tables, dev):
    from nvalchemiops.distributed import segment_energy
    from nvalchemiops.interactions.dispersion import D3Parameters, dftd3
    from nvalchemiops.interactions.electrostatics import particle_mesh_ewald
    from nvalchemiops.neighborlist import batch_cell_list

    t = lambda a: torch.as_tensor(a, device=dev)  # noqa: E731
    params = D3Parameters(rcov=t(tables["rcov"]), r4r2=t(tables["r4r2"]), c6ab=t(tables["c6ab"]), cn_ref=t(tables["cn_ref"]))
    pbc = torch.ones((nsys, 3), dtype=torch.bool, device=dev)
    nm, num, sh = batch_cell_list(pos, 6.0, cell, pbc, bi, max_neighbors=128)
    assert int(num.max()) <= 128
    e_pme, _ = particle_mesh_ewald(pos, q, cell, alpha=torch.full((nsys,), 0.4, dtype=torch.float64, device=dev), mesh_dimensions=(24, 24, 24),
                                   spline_order=5, batch_idx=bi, neighbor_matrix=nm, neighbor_matrix_shifts=sh, compute_forces=True)
    dm, dnum, dsh = batch_cell_list(pos_b, 20.0, cell_b, pbc, bi, max_neighbors=1024)
    assert int(dnum.max()) <= 1024
    e_d3, f_d3, cn, vir = dftd3(pos_b, numbers, a1=0.4289, a2=4.4407, s8=0.7875, d3_params=params, neighbor_matrix=dm, neighbor_matrix_shifts=dsh,
                                cell=cell_b, batch_idx=bi, compute_virial=True, num_systems=nsys, fill_value=pos.shape[0])
    return torch.stack([e_d3.double(), segment_energy(e_pme, bi, nsys)], dim=1)


def _worker(rank, world, port, out, backend="gloo", own_device=False):
    import torch.distributed as dist

    sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
    from nvalchemiops.distributed import all_gather_system_values, partition_systems, shard_batch

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank if own_device else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    d, counts, tables = _batch(dev)
    s0, s1, a0, a1, bi, (pos, q, numbers, pos_b), (cell, cell_b) = shard_batch(
        d["ptr"], rank, world, d["pos"], d["q"], d["numbers"], d["pos_b"], per_system=(d["cell"], d["cell_b"]))
    local = _step(pos.contiguous(), cell.contiguous(), q.contiguous(), numbers.contiguous(), pos_b.contiguous(), cell_b.contiguous(), bi, s1 - s0,
                  tables, dev)
    sizes = [b - a for a, b in partition_systems(counts, world)]
    full = all_gather_system_values(local, sizes)
    if backend == "nccl":  # the same gather through the C ABI's own communicator (mi_comm_*): the id travels over the process group once
        from nvalchemiops.distributed import NativeCommunicator

        with NativeCommunicator(rank, world) as comm:
            again = all_gather_system_values(local, sizes, comm=comm)
            torch.cuda.synchronize()
        assert torch.equal(again, full)
    if rank == 0:
        bi_all = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int32, device=dev), torch.tensor(counts, device=dev))
        single = _step(d["pos"], d["cell"], d["q"], d["numbers"], d["pos_b"], d["cell_b"], bi_all, len(counts), tables, dev)
        out["gathered"], out["single"], out["sizes"] = full.cpu().numpy(), single.cpu().numpy(), sizes
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharded_batch_step_equals_single_rank():
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    g, s = out["gathered"], out["single"]
    assert g.shape == s.shape == (6, 2) and sum(out["sizes"]) == 6 and min(out["sizes"]) >= 1
    # D3 energies: fp32, a system's sum does not depend on which other systems share the launch (rtol = atol = 1e-6, the reference's
    # batch-vs-individual bar, test_dftd3.py:2386-2391); PME: fp64, FFT batch composition only
    np.testing.assert_allclose(g[:, 0], s[:, 0], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(g[:, 1], s[:, 1], rtol=1e-10, atol=1e-10)
    assert np.all(np.abs(s) > 1e-3)


def test_two_ranks_over_rccl_one_gpu_each():
    """The same sharded step with `backend="nccl"` (RCCL on ROCm), one rank per GPU: runs wherever the box has >= 2 GPUs, so that the
    first multi-GPU driver run is not the first execution of the RCCL branch (`all_gather_system_values` on device tensors)."""
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs >= 2 GPUs for an RCCL process group (this box has {torch.cuda.device_count()}); the gloo variant above covers the "
                    "same code path on one device")
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out, "nccl", True), nprocs=2, join=True)
    g, s = out["gathered"], out["single"]
    np.testing.assert_allclose(g[:, 0], s[:, 0], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(g[:, 1], s[:, 1], rtol=1e-10, atol=1e-10)


def test_native_communicator_single_rank():
    """`mi_comm_*` (csrc/comm.cpp) on the one GPU of the test box: an RCCL communicator of one rank -- id, init, fp32 / fp64 all-gather on the
    current stream, ragged per-system gather, in-place form, destroy.  The two-rank form runs in the RCCL test above on boxes with 2 GPUs."""
    import ctypes

    sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
    from nvalchemiops import _capi as C
    from nvalchemiops.distributed import NativeCommunicator, all_gather_system_values

    dev = torch.device("cuda:0")
    ver = ctypes.c_int(0)
    C.check(C.lib().mi_comm_library_version(ctypes.byref(ver)), "mi_comm_library_version")
    assert ver.value > 20000  # NCCL_VERSION_CODE of an RCCL 2.x
    with NativeCommunicator(0, 1, device=dev) as comm:
        n_ranks, rank = ctypes.c_int(-1), ctypes.c_int(-1)
        C.check(C.lib().mi_comm_size(comm._comm, ctypes.byref(n_ranks), ctypes.byref(rank)), "mi_comm_size")
        assert (n_ranks.value, rank.value) == (1, 0)
        for dt in (torch.float32, torch.float64):
            local = torch.randn(7, 2, dtype=dt, device=dev)
            (got,) = comm.all_gather(local)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # the collective is enqueued on the caller's current stream
                (got2,) = comm.all_gather(local * 2)
            torch.cuda.current_stream().wait_stream(side)
            assert torch.equal(got, local) and torch.equal(got2, local * 2)
            assert torch.equal(all_gather_system_values(local, [7], comm=comm), local)
            buf = local.clone().reshape(-1)  # in place: send == recv + rank * count
            C.check((C.lib().mi_comm_allgather_f32 if dt == torch.float32 else C.lib().mi_comm_allgather_f64)(
                comm._comm, C.ptr(buf), C.ptr(buf), ctypes.c_size_t(buf.numel()), C.stream_of(buf)), "mi_comm_allgather (in place)")
            assert torch.equal(buf, local.reshape(-1))
        with pytest.raises(ValueError):
            comm.all_gather(torch.zeros(3, dtype=torch.int32, device=dev))
        assert comm.all_gather(torch.zeros(0, dtype=torch.float32, device=dev))[0].numel() == 0
    with pytest.raises(RuntimeError):
        comm.all_gather(torch.zeros(3, device=dev))


@pytest.mark.parametrize("workload", ["headline", "c5"])
def test_bench_launches_its_own_ranks(workload):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: two ranks (sharing the device over gloo on a 1-GPU box), n_gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    extra = ["--atoms", "20000"] if workload == "headline" else ["--workload", "c5", "--systems", "4"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-sample", "0"] + extra,
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["ranks"] == 2 and res["value"] > 0
    assert res["config"]["atoms_per_gpu"] == (20000 if workload == "headline" else 8000)
    per_rank = res["ranks"]
    assert len(per_rank["ms_per_step"]) == 2 and per_rank["imbalance_max_over_min"] >= 1.0
    if workload == "c5":
        assert per_rank["all_gather_us_median"] is not None and per_rank["all_gather_us_median"] > 0

"""Batch (system-level) data parallelism over the GPUs of one node -- the one parallel axis of this path.

The reference has no multi-GPU code at all.  Every kernel of the hot path touches per-system
data only through ``batch_idx[atom]``, so a batch of independent systems shards embarrassingly: contiguous ranges of
systems per rank (balanced by atom count), rank-local neighbour lists / D3 / PME with no data-path communication, and
ONE collective per step -- an RCCL all_gather (``backend="nccl"`` is RCCL on ROCm; ``gloo`` works for CPU tests) of the
per-system energies, <= 4 KiB for 1024 systems, latency-bound over xGMI.  Forces and neighbour lists stay sharded.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def partition_systems(atoms_per_system, world_size: int) -> list[tuple[int, int]]:
    """Contiguous system ranges [s0, s1) per rank with near-equal atom counts (greedy prefix split)."""
    counts = [int(c) for c in atoms_per_system]
    total = sum(counts)
    bounds, acc, s = [0], 0, 0
    for r in range(1, world_size):
        target = total * r / world_size
        while s < len(counts) and acc + counts[s] / 2.0 <= target:
            acc += counts[s]
            s += 1
        bounds.append(s)
    bounds.append(len(counts))
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def shard_batch(batch_ptr: torch.Tensor, rank: int, world_size: int, *per_atom: torch.Tensor, per_system: tuple[torch.Tensor, ...] = ()):
    """Slice per-atom and per-system tensors for `rank`.  Returns (s0, s1, a0, a1, local_batch_idx, per_atom_slices,
    per_system_slices); `local_batch_idx` restarts at 0 and neighbour indices computed from the slices are rank-local."""
    ptr = batch_ptr.tolist()
    counts = [ptr[i + 1] - ptr[i] for i in range(len(ptr) - 1)]
    s0, s1 = partition_systems(counts, world_size)[rank]
    a0, a1 = ptr[s0], ptr[s1]
    dev = per_atom[0].device if per_atom else batch_ptr.device
    local_counts = torch.tensor(counts[s0:s1], dtype=torch.long, device=dev)
    local_idx = torch.repeat_interleave(torch.arange(s1 - s0, dtype=torch.int32, device=dev), local_counts)
    return s0, s1, a0, a1, local_idx, tuple(t[a0:a1] for t in per_atom), tuple(t[s0:s1] for t in per_system)


def all_gather_system_values(local: torch.Tensor, systems_per_rank: list[int], group=None, comm: "NativeCommunicator | None" = None) -> torch.Tensor:
    """One all_gather of per-system values ([B_local, ...] -> [B, ...]); ragged shards are padded to the largest.  `comm`: gather through the
    library's own RCCL communicator (`NativeCommunicator`, the `mi_comm_*` entry points of the C ABI) instead of torch.distributed."""
    width = max(systems_per_rank)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if comm is not None:
        out = comm.all_gather(pad)
    else:
        world = dist.get_world_size(group)
        out = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(out, pad, group=group)
    return torch.cat([o[:n] for o, n in zip(out, systems_per_rank)], dim=0)


class NativeCommunicator:
    """The collective of this path behind the C ABI: `mi_comm_unique_id` / `mi_comm_init` / `mi_comm_allgather_f32|f64` / `mi_comm_destroy`
    (include/nvalchemiops_hip.h, csrc/comm.cpp) -- RCCL over xGMI, one rank per GPU, no torch.distributed on the data path.

    The 128-byte id rank 0 draws has to reach every rank over some host channel: `exchange(id_or_None) -> id` does that (rank 0 is handed
    the id and returns it, the others are handed None and return what rank 0 sent).  Default: `broadcast_object_list` of the initialised
    torch.distributed default group (any backend -- it is used once, for these 128 bytes).  Call with this rank's GPU current."""

    ID_BYTES = 128

    def __init__(self, rank: int, world_size: int, exchange=None, device: torch.device | None = None):
        import ctypes

        from nvalchemiops import _capi as C

        self._C, self._ct = C, ctypes
        self.rank, self.world_size = int(rank), int(world_size)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._comm = ctypes.c_void_p()
        ident = None
        if self.rank == 0:
            buf = ctypes.create_string_buffer(self.ID_BYTES)
            C.check(C.lib().mi_comm_unique_id(buf, ctypes.c_size_t(self.ID_BYTES)), "mi_comm_unique_id")
            ident = buf.raw
        if exchange is None:
            exchange = self._exchange_through_torch
        ident = exchange(ident) if self.world_size > 1 or ident is None else ident
        if not isinstance(ident, (bytes, bytearray)) or len(ident) != self.ID_BYTES:
            raise ValueError(f"NativeCommunicator: the exchange must return rank 0's {self.ID_BYTES}-byte id")
        with torch.cuda.device(self.device):
            C.check(C.lib().mi_comm_init(bytes(ident), ctypes.c_size_t(self.ID_BYTES), self.world_size, self.rank, ctypes.byref(self._comm)),
                    "mi_comm_init")

    @staticmethod
    def _exchange_through_torch(ident):
        box = [ident]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def all_gather(self, local: torch.Tensor) -> list[torch.Tensor]:
        """[world_size] tensors shaped like `local` (same shape and dtype on every rank, float32 or float64), enqueued on the current stream."""
        C, ctypes = self._C, self._ct
        if self._comm.value is None:
            raise RuntimeError("NativeCommunicator: closed")
        C.require_device(local)
        if local.dtype not in (torch.float32, torch.float64):
            raise ValueError(f"NativeCommunicator.all_gather: float32 / float64 values, got {local.dtype}")
        src = local.detach().contiguous()
        out = torch.empty((self.world_size,) + tuple(src.shape), dtype=src.dtype, device=src.device)
        fn = C.lib().mi_comm_allgather_f32 if src.dtype == torch.float32 else C.lib().mi_comm_allgather_f64
        with torch.cuda.device(src.device):
            C.check(fn(self._comm, C.ptr(src), C.ptr(out), ctypes.c_size_t(src.numel()), C.stream_of(src)), "mi_comm_allgather")
        return list(out.unbind(0))

    def close(self) -> None:
        if self._comm.value is not None:
            torch.cuda.synchronize(self.device)
            comm, self._comm = self._comm, self._ct.c_void_p()
            self._C.check(self._C.lib().mi_comm_destroy(comm), "mi_comm_destroy")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def segment_energy(per_atom_energy: torch.Tensor, local_batch_idx: torch.Tensor, num_local_systems: int) -> torch.Tensor:
    """Per-system sum of per-atom energies (PME returns energies per atom, D3 per system: SURVEY F10)."""
    if per_atom_energy.is_cuda and per_atom_energy.dtype in (torch.float32, torch.float64) and per_atom_energy.dim() == 1:
        # one wave-aggregated atomic per wave and system (`mi_segment_sum`) instead of torch's index_add, which issues one same-address
        # atomic per ATOM (70 us per step for the 128 x 2000-atom shard of config 5)
        from nvalchemiops import _capi as C

        out = torch.zeros(num_local_systems, dtype=per_atom_energy.dtype, device=per_atom_energy.device)
        v, bi = per_atom_energy.detach().contiguous(), C.i32(local_batch_idx)
        C.check(C.lib().mi_segment_sum(C.ptr(v), C.ptr(bi), v.shape[0], C.dtype_code(v.dtype), C.ptr(out), C.stream_of(v)), "mi_segment_sum")
        return out
    out = torch.zeros(num_local_systems, dtype=per_atom_energy.dtype, device=per_atom_energy.device)
    return out.index_add_(0, local_batch_idx.long(), per_atom_energy)


__all__ = ["partition_systems", "shard_batch", "all_gather_system_values", "segment_energy", "NativeCommunicator"]

"""Multi-GPU: partition SUBJECTS across ranks, replicate the support points.

Every (subject, support point) pair is independent — the reference's loop nest
(src/simulator/likelihood/matrix.rs:79-98) has no cross-iteration state — so the
path shards with NO data-path collective: one process per GPU, each rank simulates
its contiguous block of subjects against the full theta grid and owns the matching
rows of the prediction tensor.

The partition rule and the one optional exchange live under the C ABI (include/pmx.h
"sharding across GPUs": ``pmx_shard_bounds`` / ``pmx_shard_rows`` /
``pmx_population_create_shard`` / ``pmx_comm_*`` / ``pmx_allgather_predictions``);
this module is their Python face.  ``all_gather_predictions`` is only for a caller
that wants the full prediction tensor on every device: each rank writes its rows
straight into its block of the full tensor and the blocks are exchanged IN PLACE
(RCCL over xGMI through the library's own communicator on GPUs; per-owner broadcasts
over ``torch.distributed`` for CPU rehearsals) — no padding, no concatenation.
NPAG-style callers consume per-subject rows and never need it.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import _ffi
from .flatten import FlatPopulation


def shard_bounds(flat: FlatPopulation, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous subject ranges, balanced by events per subject (= subject-event-steps per support point):
    ``pmx_shard_bounds``."""
    d = flat.desc()
    b = np.zeros(world_size + 1, dtype=np.int64)
    _ffi.check(_ffi.lib().pmx_shard_bounds(C.byref(d), int(world_size), b.ctypes.data))
    return [(int(b[r]), int(b[r + 1])) for r in range(world_size)]


def shard_rows(flat: FlatPopulation, bounds: List[Tuple[int, int]]) -> List[Tuple[int, int]]:
    """First / one-past-last prediction row of every shard: ``pmx_shard_rows``."""
    d = flat.desc()
    n = len(bounds)
    b = np.array([s0 for s0, _ in bounds] + [bounds[-1][1]], dtype=np.int64)
    rows = np.zeros(n + 1, dtype=np.int64)
    _ffi.check(_ffi.lib().pmx_shard_rows(C.byref(d), n, b.ctypes.data, rows.ctypes.data))
    return [(int(rows[r]), int(rows[r + 1])) for r in range(n)]


class ShardedPopulation:
    """The rank-local view of a sharded population."""

    def __init__(self, flat: FlatPopulation, rank: int, world_size: int):
        self.world_size = world_size
        self.rank = rank
        self.flat = flat
        self.bounds = shard_bounds(flat, world_size)
        self.rows = shard_rows(flat, self.bounds)
        self.n_observations_total = self.rows[-1][1]
        s0, s1 = self.bounds[rank]
        self.local = flat.subject_slice(s0, s1)
        self.local_steps_per_support = int(flat.events_per_subject()[s0:s1].sum())

    @property
    def local_rows(self) -> Tuple[int, int]:
        return self.rows[self.rank]

    def device_population(self, device: int = 0):
        """This rank's shard on its GPU (``pmx_population_create_shard``)."""
        from . import runtime

        return runtime.DevicePopulation(self.flat, device, subjects=self.bounds[self.rank])


class Communicator:
    """``pmx_comm``: one RCCL rank per GPU.  The 128-byte id is made by rank 0 and shipped to the other ranks over the
    ``torch.distributed`` group the caller already has (any backend); a C / Rust caller uses its own transport."""

    def __init__(self, device: int, group=None):
        import torch
        import torch.distributed as dist

        L = _ffi.lib()
        self.rank, self.world_size, self.device = dist.get_rank(group), dist.get_world_size(group), int(device)
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            _ffi.check(L.pmx_comm_unique_id(buf))
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        on_gpu = dist.get_backend(group) == "nccl"
        if on_gpu:
            ident = ident.to(torch.device("cuda", self.device))
        dist.broadcast(ident, src=0, group=group)
        raw = (C.c_uint8 * 128)(*ident.cpu().tolist())
        h = C.c_void_p()
        _ffi.check(L.pmx_comm_create(raw, self.world_size, self.rank, self.device, C.byref(h)))
        self.handle = h

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and _ffi is not None and _ffi._lib is not None:
            _ffi._lib.pmx_comm_destroy(h)
            self.handle = None


def full_prediction_tensor(sharded: ShardedPopulation, n_support: int, device=None, dtype=None):
    """``(full, local)``: the ``[n_observations_total, P]`` tensor and the view of this rank's rows inside it - hand
    ``local`` to ``runtime.predict(..., pred=local)`` so the kernel writes where the exchange expects the block."""
    import torch

    full = torch.empty((sharded.n_observations_total, n_support), dtype=dtype or torch.float64, device=device)
    r0, r1 = sharded.local_rows
    return full, full[r0:r1]


def all_gather_predictions(full, sharded: ShardedPopulation, comm: Optional[Communicator] = None, group=None):
    """In-place exchange of the row blocks of ``full`` (see ``full_prediction_tensor``): afterwards every rank holds every
    rank's rows.  CUDA tensors go through ``pmx_allgather_predictions`` (RCCL; ``comm`` = a ``Communicator``), enqueued
    on torch's current stream and not synchronised; CPU tensors (gloo rehearsals) through one ``dist.broadcast`` per
    owner into the same views."""
    import torch
    import torch.distributed as dist

    assert full.dim() == 2 and full.shape[0] == sharded.n_observations_total and full.stride(1) == 1
    if full.is_cuda:
        assert comm is not None, "CUDA tensors need a Communicator (pmx_comm)"
        rows = np.array([r0 for r0, _ in sharded.rows] + [sharded.rows[-1][1]], dtype=np.int64)
        stream = torch.cuda.current_stream(full.device).cuda_stream
        _ffi.check(_ffi.lib().pmx_allgather_predictions(comm.handle, full.data_ptr(), rows.ctypes.data, int(full.stride(0)), stream))
        return full
    for r, (r0, r1) in enumerate(sharded.rows):
        if r1 > r0:
            dist.broadcast(full[r0:r1], src=r, group=group)
    return full

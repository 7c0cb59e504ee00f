"""Multi-GPU: partition SUBJECTS across ranks, replicate the support points.

Every (subject, support point) pair is independent — the reference's loop nest
(src/simulator/likelihood/matrix.rs:79-98) has no cross-iteration state — so the
path shards with NO data-path collective: one process per GPU, each rank simulates
its contiguous block of subjects against the full theta grid and owns the matching
rows of the prediction tensor.

``all_gather_predictions`` is the one optional exchange (RCCL all-gather over xGMI
when the backend is "nccl"): only for a caller that wants the full prediction
tensor on every device.  NPAG-style callers consume per-subject rows and never
need it.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np

from .flatten import FlatPopulation


def shard_bounds(flat: FlatPopulation, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous subject ranges, balanced by events per subject (= subject-event-steps per support point)."""
    S = flat.n_subjects
    w = flat.events_per_subject().astype(np.float64)
    if S == 0 or w.sum() == 0:
        w = np.ones(max(S, 1))
    csum = np.concatenate([[0.0], np.cumsum(w)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        cut = int(np.searchsorted(csum, target, side="left"))
        cuts.append(min(max(cut, cuts[-1]), S))
    cuts.append(S)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


class ShardedPopulation:
    """The rank-local view of a sharded population."""

    def __init__(self, flat: FlatPopulation, rank: int, world_size: int):
        self.world_size = world_size
        self.rank = rank
        self.bounds = shard_bounds(flat, world_size)
        obs_off = flat.observation_offsets()
        self.rows = [(int(obs_off[s0]), int(obs_off[s1])) for (s0, s1) in self.bounds]
        self.n_observations_total = int(obs_off[-1])
        s0, s1 = self.bounds[rank]
        self.local = flat.subject_slice(s0, s1)
        self.local_steps_per_support = int(flat.events_per_subject()[s0:s1].sum())

    @property
    def local_rows(self) -> Tuple[int, int]:
        return self.rows[self.rank]


def all_gather_predictions(pred_local, sharded: ShardedPopulation, group=None):
    """All-gather the per-rank prediction blocks [rows_r, P] into the full [n_obs_total, P] tensor on
    every rank (``torch.distributed``; backend "nccl" == RCCL over xGMI on ROCm, "gloo" on CPU).
    Blocks are padded to the largest shard so one fixed-size collective moves everything."""
    import torch
    import torch.distributed as dist

    world = sharded.world_size
    P = pred_local.shape[1]
    max_rows = max(r1 - r0 for (r0, r1) in sharded.rows)
    if pred_local.shape[0] == max_rows:
        send = pred_local.contiguous()
    else:
        send = torch.zeros((max_rows, P), dtype=pred_local.dtype, device=pred_local.device)
        send[: pred_local.shape[0]] = pred_local
    gathered = torch.empty((world * max_rows, P), dtype=pred_local.dtype, device=pred_local.device)
    dist.all_gather_into_tensor(gathered, send, group=group)
    if all((r1 - r0) == max_rows for (r0, r1) in sharded.rows):
        return gathered
    parts = [gathered[r * max_rows: r * max_rows + (r1 - r0)] for r, (r0, r1) in enumerate(sharded.rows)]
    return torch.cat(parts, dim=0)

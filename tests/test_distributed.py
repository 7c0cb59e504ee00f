"""N>1 path on CPU: world_size-2 gloo processes.  Each rank takes its subject shard and the full theta
grid; the compute stand-in is the CPU oracle (tests may use it; the product path is HIP-only), so this
covers exactly the sharding and gather logic the GPU ranks run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from pharmsol_amd import Data, synth
from pharmsol_amd.distributed import ShardedPopulation, all_gather_predictions, shard_bounds
from tests import models


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ragged_population():
    rng = np.random.default_rng(21)
    subjects = [models.random_subject(rng, multi_occasion=True) for _ in range(37)]
    m = models.handwritten_analytical("two_compartments", 0, 4)
    return m, m.flatten(Data(subjects)), synth.theta_c3(5)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, flat, theta = _ragged_population()
        sh = ShardedPopulation(flat, rank, world)
        local, _ = oracle.predict(m, sh.local, theta, nthreads=1)
        r0, r1 = sh.local_rows
        assert local.shape == (r1 - r0, theta.shape[0])
        full = all_gather_predictions(torch.from_numpy(local), sh).numpy()
        # whole-job step count: sum over ranks (what bench.py reports as `value` numerator)
        steps = torch.tensor([sh.local_steps_per_support], dtype=torch.int64)
        dist.all_reduce(steps)
        q.put((rank, full, int(steps.item())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    m, flat, theta = _ragged_population()
    want, _ = oracle.predict(m, flat, theta, nthreads=1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, full, steps in results:
        np.testing.assert_array_equal(full, want)  # every rank holds the full tensor, rows in subject order
        assert steps == flat.n_events


def test_shard_bounds_cover_and_balance():
    m, flat, _ = synth.config_c3(1000, 2)
    for world in (1, 2, 3, 4, 8):
        b = shard_bounds(flat, world)
        assert b[0][0] == 0 and b[-1][1] == 1000 and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        sizes = [s1 - s0 for s0, s1 in b]
        assert max(sizes) - min(sizes) <= 1
    # ragged: balanced by events, not by subject count
    m, flat, _ = synth.config_c4(2000)
    ev = flat.events_per_subject()
    b = shard_bounds(flat, 4)
    loads = [ev[s0:s1].sum() for s0, s1 in b]
    assert max(loads) - min(loads) <= 2 * ev.max()


def test_more_ranks_than_subjects():
    m, flat, theta = synth.config_c3(3, 2)
    b = shard_bounds(flat, 8)
    assert sum(s1 - s0 for s0, s1 in b) == 3
    sh = ShardedPopulation(flat, 7, 8)
    assert sh.local.n_subjects in (0, 1)


def test_subject_slices_concatenate_to_the_whole():
    m, flat, theta = _ragged_population()
    want, _ = oracle.predict(m, flat, theta, nthreads=1)
    parts = []
    for r in range(3):
        sh = ShardedPopulation(flat, r, 3)
        p, _ = oracle.predict(m, sh.local, theta, nthreads=1)
        parts.append(p)
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), want)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_launches_its_own_ranks_from_the_plain_command_line(scaling):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts torch.distributed.run itself (before
    anything touches a GPU) and passes the children's exit code on.  --dry-run = the plumbing without kernels (this box
    has no GPU): process group, subject sharding, step-count / time reductions, the one JSON line from rank 0."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--scaling", scaling,
                        "--subjects", "1000", "--support", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None and d["scaling"] == scaling
    total_subjects = 1000 if scaling == "strong" else 2000
    assert d["config"]["steps_per_pass"] == total_subjects * 8 * 8  # all ranks' subject-event-steps per pass
    assert d["config"]["subjects_per_gpu"] == (500 if scaling == "strong" else 1000)

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
from pharmsol_amd.distributed import ShardedPopulation, all_gather_predictions, full_prediction_tensor, shard_bounds, shard_rows
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
        # the rank's rows go straight into its block of the full tensor; the blocks (of unequal size here) are exchanged
        # in place: no padding, no concatenation
        full_t, mine = full_prediction_tensor(sh, theta.shape[0])
        full_t.fill_(float("nan"))
        mine.copy_(torch.from_numpy(local))
        full = all_gather_predictions(full_t, sh).numpy()
        assert full_t.data_ptr() == torch.from_numpy(full).data_ptr()
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


def test_c_abi_shard_bounds_follow_the_events_balanced_rule():
    """pmx_shard_bounds / pmx_shard_rows against a numpy restatement of the rule: rank r starts at the first subject whose
    predecessors hold at least r/n of all events (a searchsorted on the running event count)."""
    rng = np.random.default_rng(3)
    for flat in (_ragged_population()[1], synth.config_c4(777)[1], synth.population_c5(100), synth.population_c23(5)):
        ev = flat.events_per_subject().astype(np.float64)
        csum = np.concatenate([[0.0], np.cumsum(ev)])
        off = flat.observation_offsets()
        for world in (1, 2, 3, 5, 8, 16):
            cuts = [0]
            for r in range(1, world):
                cuts.append(min(max(int(np.searchsorted(csum, csum[-1] * r / world, side="left")), cuts[-1]), flat.n_subjects))
            cuts.append(flat.n_subjects)
            b = shard_bounds(flat, world)
            assert b == [(cuts[r], cuts[r + 1]) for r in range(world)]
            assert shard_rows(flat, b) == [(int(off[s0]), int(off[s1])) for s0, s1 in b]
    # a population without events is split by head count; bad arguments are refused
    import ctypes as C

    from pharmsol_amd import _abi, _ffi
    empty = synth.population_c23(0)
    assert shard_bounds(empty, 3) == [(0, 0)] * 3
    d = flat.desc()
    out = np.zeros(4, dtype=np.int64)
    assert _ffi.lib().pmx_shard_bounds(C.byref(d), 0, out.ctypes.data) == _abi.PMX_ERR_INVALID_ARGUMENT
    bad = np.array([0, 3, 2, flat.n_subjects], dtype=np.int64)
    assert _ffi.lib().pmx_shard_rows(C.byref(d), 3, bad.ctypes.data, out.ctypes.data) == _abi.PMX_ERR_INVALID_ARGUMENT


def test_c_abi_collective_entry_points_fail_loudly_without_a_device():
    """pmx_comm_* / pmx_population_create_shard on a box without a GPU: argument checks first, then PMX_ERR_NO_DEVICE - no
    silent CPU path."""
    import ctypes as C

    from pharmsol_amd import _abi, _ffi
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    L = _ffi.lib()
    flat = synth.population_c23(10)
    d = flat.desc()
    h = C.c_void_p()
    assert L.pmx_population_create_shard(C.byref(d), 4, 2, 0, C.byref(h)) == _abi.PMX_ERR_INVALID_ARGUMENT
    assert L.pmx_population_create_shard(C.byref(d), 0, 11, 0, C.byref(h)) == _abi.PMX_ERR_INVALID_ARGUMENT
    assert L.pmx_population_create_shard(C.byref(d), 2, 7, 0, C.byref(h)) == _abi.PMX_ERR_NO_DEVICE
    ident = (C.c_uint8 * 128)()
    assert L.pmx_comm_create(ident, 2, 2, 0, C.byref(h)) == _abi.PMX_ERR_INVALID_ARGUMENT  # rank out of range
    assert L.pmx_comm_create(ident, 2, 1, 0, C.byref(h)) == _abi.PMX_ERR_NO_DEVICE
    rows = np.array([0, 5], dtype=np.int64)
    assert L.pmx_allgather_predictions(None, None, rows.ctypes.data, 4, None) == _abi.PMX_ERR_INVALID_ARGUMENT


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


# --------------------------------------------------------------------------- GPU (one device)
@pytest.mark.gpu
def test_shard_populations_write_the_rows_of_the_whole_population():
    """pmx_population_create_shard for three events-balanced shards of a ragged, multi-occasion, covariate-free population
    and of C5 (covariate knots re-based): each shard's kernel writes its row block of one full tensor; the result equals
    the whole population's pass bit for bit."""
    from pharmsol_amd import runtime

    for m, flat, theta in (_ragged_population(), (synth.model_three_cpt_abs_wt(), synth.population_c5(300), synth.theta_c5(40))):
        theta = np.ascontiguousarray(theta[:40] if theta.shape[0] > 40 else theta)
        whole, _ = runtime.predict(m, runtime.DevicePopulation(flat, 0), theta)
        n = 3
        b = shard_bounds(flat, n)
        rows = shard_rows(flat, b)
        full = torch.full_like(whole, float("nan"))
        for r in range(n):
            pop = runtime.DevicePopulation(flat, 0, subjects=b[r])
            assert pop.n_subjects == b[r][1] - b[r][0] and pop.n_observations == rows[r][1] - rows[r][0]
            if pop.n_observations:
                runtime.predict(m, pop, theta, pred=full[rows[r][0]:rows[r][1]])
        torch.cuda.synchronize()
        assert torch.equal(full, whole)


@pytest.mark.gpu
@pytest.mark.parametrize("force_broadcast", [False, True])
def test_rccl_communicator_and_in_place_allgather_on_one_rank(force_broadcast, monkeypatch):
    """The RCCL calls under the C ABI with a world of ONE rank (all a single-GPU box can host): ncclGetUniqueId,
    ncclCommInitRank, the in-place ncclAllGather and - forced - the grouped ncclBroadcast form, stream-ordered behind the
    kernel that wrote the block.  (Ranks > 1 first meet hardware in the driver's multi-GPU run.)"""
    import ctypes as C

    from pharmsol_amd import _ffi, runtime

    if force_broadcast:
        monkeypatch.setenv("PMX_DEBUG_ALLGATHER_BROADCAST", "1")
    L = _ffi.lib()
    ident = (C.c_uint8 * 128)()
    _ffi.check(L.pmx_comm_unique_id(ident))
    assert any(ident)
    h = C.c_void_p()
    _ffi.check(L.pmx_comm_create(ident, 1, 0, 0, C.byref(h)))
    try:
        assert L.pmx_comm_size(h) == 1 and L.pmx_comm_rank(h) == 0
        m, flat, theta = synth.config_c3(500, 64)
        pop = runtime.DevicePopulation(flat, 0)
        want, _ = runtime.predict(m, pop, theta)
        full = torch.full((pop.n_observations, 64), float("nan"), dtype=torch.float64, device="cuda:0")
        runtime.predict(m, pop, theta, pred=full)
        rows = np.array([0, pop.n_observations], dtype=np.int64)
        _ffi.check(L.pmx_allgather_predictions(h, full.data_ptr(), rows.ctypes.data, 64, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert torch.equal(full, want)
        bad = np.array([1, pop.n_observations], dtype=np.int64)
        from pharmsol_amd import _abi
        assert L.pmx_allgather_predictions(h, full.data_ptr(), bad.ctypes.data, 64, None) == _abi.PMX_ERR_INVALID_ARGUMENT
    finally:
        L.pmx_comm_destroy(h)

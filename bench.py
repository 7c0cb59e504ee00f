#!/usr/bin/env python3
"""bench.py — subject-event-steps/s of the HIP prediction path on BASELINE.json's headline workload.

A "step" of this benchmark = one pass of the hot path (every subject x every support point, one
kernel launch) over one resident batch of synthetic input.  Workload at N=1 = C3, the configuration
the north-star target is quoted on: two-compartment IV analytical, 100k subjects x 1000 support
points, 8 events per subject (8e8 subject-event-steps per pass).

N>1: one process per GPU.  `python bench.py --gpus N` launches itself under torch.distributed.run when
it was not started by it (the parent touches no GPU and exits with the children's code).  Subjects are
sharded across ranks, theta is replicated; there is no data-path collective (the reference's loop nest,
likelihood/matrix.rs:79-98, has no exchange step).  `--scaling weak` (default): every rank DRAWS its own
100k-subject shard of an N x 100k population (no rank ever builds the global one); `--scaling strong`: ONE
100k x 1000 population (BASELINE configs[2]) split N ways by pmx_shard_bounds.  `--gather` additionally times
passes that end in the optional in-place RCCL all-gather of the prediction blocks (pmx_allgather_predictions over
the library's own communicator; reported beside `value` as `gather.value_with_gather`, never as `value`).

Contract: W untimed warm-up passes, then exactly K timed passes bracketed by barrier +
torch.cuda.synchronize() on both sides; MAX over ranks; rank 0 prints ONE JSON line.
Inputs (population, theta) are resident in HBM before the timed region starts.

Extra objects on the line:
  roofline      dominant kernel vs the roofline that bounds it.  "hbm" (C3, C2): achieved = algorithmic bytes
                per launch (8*S*O*P + 8*P*k + 26*S*E) / mean launch duration measured with HIP events
                on the launch stream inside the timed region; peak = 8 TB/s.  "fp64_valu" (C5, C4, --ragged:
                compute-bound kernels): achieved = FP64-rate vector wave-instructions per launch (PMC SQ_INSTS_VALU,
                profiles/kernel_counters.json) / launch duration; peak = 1024 SIMDs x 2.4 GHz / 4 cycles per
                wave-instruction.  `traffic` / `valu_insts` come from committed rocprofv3 --pmc runs of the same
                command (separate passes); `*_source` names the file.
  cpu_baseline  the CPU oracle (a port of the reference algorithm, OpenMP over subjects like the
                reference's rayon loop) on a bounded sample of the same workload, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# FP64 vector issue: one wave64 instruction per 4 cycles per SIMD; 256 CUs x 4 SIMDs; 2.4 GHz max clock (same guide)
FP64_VALU_PEAK_GINST = 1024 * 2.4 / 4.0


def algorithmic_bytes(S, O, P, k, E, cov_segments=0):
    """each array touched once."""
    return 8 * S * O * P + 8 * P * k + 26 * S * E + 24 * cov_segments


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--subjects", type=int, default=100_000, help="subjects per GPU (weak) / in total (strong); C3: 100000")
    ap.add_argument("--support", type=int, default=1000, help="support points (C3: 1000)")
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c4", "c5", "user"],
                    help="c2..c5: BASELINE.json's configs; user: the reference's covariate parity model "
                         "(tests/analytical_macro_lowering.rs:225-260, closures compiled at run time), 50k subjects x 256")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = --subjects per GPU; strong = one population of --subjects split N ways (BASELINE configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle sample")
    ap.add_argument("--gather", action="store_true", help="N>1: also time the optional RCCL all-gather")
    ap.add_argument("--no-status", action="store_true", help="do not write the per-pair status bytes")
    ap.add_argument("--loglik", action="store_true",
                    help="time the fused log-likelihood entry (pmx_loglik_device) instead of predictions: output S x P")
    ap.add_argument("--spin-up-ms", type=float, default=80.0,
                    help="untimed back-to-back passes before the warm-up, to bring the device clocks up after set-up")
    ap.add_argument("--place-gib", type=float, default=96.0,
                    help="size limit of the arena searched for a fast window for the prediction matrix "
                         "(runtime.place_predictions; capped at 75 %% of the free device memory); 0 = plain first allocation")
    ap.add_argument("--alloc-tries", type=int, default=8,
                    help="plain allocations timed as placement candidates beside the arena's windows (the first one is "
                         "reported as frac_first_allocation)")
    ap.add_argument("--ragged", action="store_true",
                    help="C3 with per-subject jittered sampling times (no shared design, no related step lengths)")
    ap.add_argument("--constant-cov", action="store_true",
                    help="c5: one wt value per subject instead of 2-4 interpolation knots")
    ap.add_argument("--ld", type=int, default=0,
                    help="row pitch of the prediction matrix in doubles (>= support points).  0 = the library's recommendation "
                         "(pmx_recommended_ld: rows start on 128-byte boundaries; 1000 -> 1008), -1 = dense rows + the arena search")
    ap.add_argument("--no-class", action="store_true",
                    help="A/B: disable the classed kernel (shared-design propagator reuse); every subject walks the generic kernel")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing rehearsal without a GPU: launch, process group (gloo), sharding, reductions and the JSON "
                         "line run, the kernel passes do not; the line carries \"dry_run\": true and value null")
    return ap.parse_args(argv)


def launch_children(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run BEFORE anything
    touches the GPU (this parent never does) and pass their exit code on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_children(args))
    if args.no_class:
        os.environ["PMX_DISABLE_CLASSING"] = "1"

    import numpy as np
    import torch
    import torch.distributed as dist

    from pharmsol_amd import synth
    from pharmsol_amd.distributed import shard_bounds

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
        sys.exit(2)
    dry = args.dry_run
    if not dry:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists; --dry-run rehearses the plumbing only)"
    backend = "gloo" if dry else os.environ.get("PMX_BENCH_BACKEND", "nccl")
    dev = None
    device_index = 0
    if not dry:
        n_dev = torch.cuda.device_count()
        device_index = local_rank % n_dev  # (rehearsal on a 1-GPU box maps every rank to cuda:0)
        torch.cuda.set_device(device_index)
        dev = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = dev if (world > 1 and backend == "nccl") else torch.device("cpu")

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[device_index]) if backend == "nccl" else dist.barrier()

    def sync():
        if not dry:
            torch.cuda.synchronize()

    # ---------------------------------------------------------------- workload (synthetic, seeded)
    strong = args.scaling == "strong"
    P = args.support
    batch = False
    bound = "hbm"
    if args.workload in ("c3", "c2"):
        S_arg = args.subjects
        if args.workload == "c2":
            S_arg, P = (10_000 if args.subjects == 100_000 else args.subjects), 1
        S_global = S_arg if strong else S_arg * world
        model = synth.model_two_cpt_iv()
        theta = synth.theta_c3(P) if args.workload == "c3" else synth.theta_c2()
        gen = lambda n, shard: synth.population_c23(n, ragged=args.ragged, shard=shard)
        label = f"C{'3' if args.workload == 'c3' else '2'}: two_compartments analytical, %s x {P} support points, 8 events/subject"
        counters_key = "c3_ragged" if args.ragged else args.workload
        if args.ragged:
            bound = "fp64_valu"
            label += " (jittered sampling times)"
        if args.no_class or args.loglik:  # the generic walker / the fused fold are instruction-bound, not write-bound
            bound = "fp64_valu"
            counters_key += "_generic" if args.no_class else "_loglik"
        dtype_tol = 1e-6
    elif args.workload == "c4":
        S_arg = 50_000 if args.subjects == 100_000 else args.subjects
        S_global = S_arg if strong else S_arg * world
        model = synth.model_one_cmt_iv_ode()
        theta_all = None
        gen = lambda n, shard: synth.config_c4(n, shard=shard)[1:]  # (population, one theta row per subject)
        batch, P = True, 1
        bound, counters_key = "fp64_valu", "c4"
        label = "C4: ode one_cmt_iv RK4 h<=0.02, %s, irregular schedules, one theta per subject (ODE parity is against the RK4 oracle and closed forms: the reference's diffsol BDF is unpinnable here)"
        dtype_tol = 1e-4
    elif args.workload == "user":
        S_arg = 50_000 if args.subjects == 100_000 else args.subjects
        S_global = S_arg if strong else S_arg * world
        P = 256 if args.support == 1000 else args.support
        model = synth.model_user_covariates()
        theta = synth.theta_user(P)
        gen = lambda n, shard: synth.population_user(n, shard=shard)
        bound, counters_key = "fp64_valu", "user"
        label = f"user closures: one_compartment_with_absorption, lag / fa / init / volume as functions of (theta, t, wt, renal), %s x {P} support points"
        dtype_tol = 1e-6
    else:
        S_arg = 200_000 if args.subjects == 100_000 else args.subjects
        S_global = S_arg if strong else S_arg * world
        P = 512 if args.support == 1000 else args.support
        model = synth.model_three_cpt_abs_wt()
        theta = synth.theta_c5(P)
        gen = lambda n, shard: synth.population_c5(n, constant_wt=args.constant_cov, shard=shard)
        bound, counters_key = "fp64_valu", "c5"
        label = f"C5: three_compartments_with_absorption + {'subject-constant' if args.constant_cov else 'time-varying'} wt covariate, %s x {P} support points"
        dtype_tol = 1e-6
    label = label % (f"{S_global} subjects split over {world} GPU(s)" if strong else f"{S_arg} subjects/GPU")
    # This rank's subjects.  weak: its OWN draw of S_arg subjects (shard `rank` of the N x S_arg population; nobody builds
    # the global one).  strong: ONE population of S_arg subjects (a single GPU's normal load, cheap to build on every
    # rank), cut by the library's events-balanced rule (pmx_shard_bounds) - the rank keeps its slice only.
    if strong and world > 1:
        g = gen(S_arg, 0)
        flat_global, theta_all = g if batch else (g, None)
        s0, s1 = shard_bounds(flat_global, world)[rank]
        flat = flat_global.subject_slice(s0, s1)
        if batch:
            theta = theta_all[s0:s1]
        del flat_global, g
    else:
        g = gen(S_arg, rank)
        flat, theta = g if batch else (g, theta)
    k = theta.shape[1]
    steps_per_pass_local = flat.n_events * (1 if batch else P)

    kernel_name, placed = "", "first allocation"
    row_pitch = None
    pass_ms, first_alloc_ms = [], None
    elapsed = 0.0
    pred = None
    em = None
    attainable_gbs = None
    if dry:
        barrier()
        t0 = time.perf_counter()
        barrier()
        elapsed = time.perf_counter() - t0
        n_obs = flat.n_observations
    else:
        from pharmsol_amd import runtime

        if args.loglik:
            assert not batch, "--loglik is defined for the matrix shape (subjects x support points)"
            from pharmsol_amd import AssayErrorModel, AssayErrorModels, ErrorPoly, _abi

            # 'measured' values: this rank's predictions at the first support point x deterministic lognormal noise
            pop0 = runtime.DevicePopulation(flat, device_index)
            p0, _ = runtime.predict(model, pop0, theta[:1])
            torch.cuda.synchronize()
            rng = synth.SplitMix64(synth.SEED ^ 0x11 ^ rank)
            noise = np.exp(0.2 * (rng.uniform(pop0.n_observations) - 0.5))
            vals = np.abs(p0.cpu().numpy()[:, 0]) * noise + 0.05
            flat.ev_value = flat.ev_value.copy()
            flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION] = vals
            del pop0, p0
            em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
            label += " -> fused log-likelihood (additive error, sigma from the observation)"
        pop = runtime.DevicePopulation(flat, device_index)
        d_theta = torch.as_tensor(np.ascontiguousarray(theta), device=dev)
        n_obs = pop.n_observations
        out_shape = (n_obs,) if batch else ((pop.n_subjects, P) if args.loglik else (n_obs, P))
        ld = None
        if not batch and not args.loglik and args.ld >= 0:
            ld = max(args.ld, P) if args.ld else runtime.recommended_ld(P)
            if ld == P:
                ld = None  # (already a multiple of 16: the dense path)
        # (N > 1: no placement search - eight ranks each timing 100 GiB of arena windows would say more about the search
        # than about the path; every rank writes into a plain allocation, what a caller's own buffer gets)
        want_placement = (world == 1 and not batch and not args.loglik and args.place_gib > 0 and ld is None and
                          n_obs * P * 8 > (1 << 28))
        want_gather = world > 1 and args.gather and not batch and not args.loglik
        full = comm = rows_all = None
        if want_gather:
            # every rank's row count -> the row blocks of the full tensor; this rank's kernel writes straight into its block
            from pharmsol_amd.distributed import Communicator

            cnt = torch.tensor([n_obs], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            allc = [torch.zeros_like(cnt) for _ in range(world)]
            dist.all_gather(allc, cnt)
            starts_ = np.concatenate([[0], np.cumsum([int(c.item()) for c in allc])]).astype(np.int64)
            rows_all = [(int(starts_[r]), int(starts_[r + 1])) for r in range(world)]
            comm = Communicator(device_index)
        em_c = em.to_c(model) if em is not None else None
        status = None if args.no_status else torch.zeros((pop.n_subjects,) if batch else (pop.n_subjects, P),
                                                         dtype=torch.uint8, device=dev)
        pred = None

        def one_pass(out=None):
            if args.loglik:
                runtime.loglik(model, pop, em_c, d_theta, ll=pred, status=status, want_status=not args.no_status)
            else:
                runtime.predict(model, pop, d_theta, pred=pred if out is None else out, status=status, batch=batch,
                                want_status=not args.no_status)

        def spin_up(out=None):
            # untimed back-to-back passes: after the set-up phase (allocations, frees, host work) the device sits idle for
            # a while and comes back with reduced clocks (profiles/r01/dispatch_ramp.txt)
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < args.spin_up_ms * 1e-3:
                for _ in range(8):
                    one_pass(out)
                torch.cuda.synchronize()

        def ms_into(out, reps=10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                one_pass(out)
            e0.record()
            for _ in range(reps):
                one_pass(out)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        row_pitch = (ld if (ld is not None and not want_gather) else P) if not (batch or args.loglik) else None
        if ld is not None and not want_gather:
            # rows padded to a pitch (ld_pred of pmx_predict_device; the same bytes are written, the padding never is).
            # The first plain allocation of the process is what a caller's own buffer gets (`frac_first_allocation`);
            # --alloc-tries - 1 more are timed with the real kernel during set-up, all alive at once so that each sits on
            # different memory, and the fastest is kept (where the matrix lands in HBM moves the row-strided stream by up to
            # 25 %)
            pred = torch.empty((n_obs, ld), dtype=torch.float64, device=dev)[:, :P]
            placed = "one plain allocation, rows padded to %d doubles" % ld
            if n_obs * P * 8 > (1 << 28) and args.alloc_tries > 1 and args.place_gib > 0:
                spin_up(pred)
                first_alloc_ms = best_ms = ms_into(pred)
                held, n_plain = [pred], 1
                for _ in range(args.alloc_tries - 1):
                    try:
                        cand = torch.empty((n_obs, ld), dtype=torch.float64, device=dev)[:, :P]
                    except RuntimeError:
                        break
                    held.append(cand)
                    n_plain += 1
                    ms_c = ms_into(cand, reps=6)
                    if ms_c < best_ms:
                        pred, best_ms = cand, ms_c
                del held
                cand = None
                torch.cuda.empty_cache()
                placed = "best of %d plain allocations, rows padded to %d doubles (pmx_recommended_ld)" % (n_plain, ld)
                # ... and, at N = 1, the best window of an arena the library maps chunk by chunk and times window by window
                # (pmx_prediction_buffer_create_pitched, exhaustive form): boxes exist where none of the plain allocations
                # is of the fast kind (profiles/r03/bench_c3_box_without_a_fast_plain_allocation.json).  (N > 1: ranks time
                # their few plain candidates only - eight ranks each timing 100 GiB of arena windows would say more about the
                # search than about the path -)
                free_b, _tot = torch.cuda.mem_get_info(dev)
                gib = min(args.place_gib, 0.75 * free_b / (1 << 30))
                try:
                    if world > 1:
                        # ... unless the best of them is clearly of the slow kind: a flat fill of the same buffer gives this
                        # device's write ceiling (pmx_measure_write_ceiling, ~5 ms), and a rank whose kernel stays below
                        # 92 % of it runs the library's quick search (it stops inside the first fast plateau, ~0.1 s)
                        import ctypes as C

                        from pharmsol_amd import _ffi
                        gbs = C.c_double()
                        base = pred._base if getattr(pred, "_base", None) is not None else pred
                        _ffi.check(_ffi.lib().pmx_measure_write_ceiling(base.data_ptr(), int(base.numel()), 3,
                                                                        torch.cuda.current_stream(dev).cuda_stream, C.byref(gbs)))
                        if 8.0 * n_obs * P / (best_ms * 1e-3) / 1e9 >= 0.92 * gbs.value:
                            raise StopIteration
                    cand = runtime.place_predictions(model, pop, d_theta, search_gib=gib, exhaustive=(world == 1), ld=ld)
                    if ms_into(cand) < best_ms:
                        pred = cand
                        placed = "best window of a %.0f GiB arena, rows padded to %d doubles (%.3f ms during the search)" % (
                            gib, ld, cand._pmx_owner.ms_per_pass)
                except StopIteration:
                    pass
                except Exception as e:  # no virtual-memory API / not enough memory
                    placed += " (arena search failed: %s)" % type(e).__name__
                cand = None
                torch.cuda.empty_cache()
        elif want_placement:
            # Where the matrix lands in HBM changes the write rate of the row-strided stream by up to 25 %, and which memory
            # is the fast kind differs from box to box.  Candidates, all timed with the real kernel
            # during set-up (outside the timed region; the chosen buffer is then reused by every pass):
            #   1. the plain first allocation of the process (what a caller who hands pmx_predict_device his own buffer
            #      gets: `frac_first_allocation`) and --alloc-tries - 1 more plain allocations;
            #   2. the best window of an arena the library maps chunk by chunk and times window by window
            #      (pmx_prediction_buffer_create, exhaustive form).
            # (PMX_TUNE_PLACE_WINDOW = a traced re-run into the window an earlier run chose: no other candidate, so that
            # the trace holds passes into that window only; tools/profile_round.sh)
            forced = os.environ.get("PMX_TUNE_PLACE_WINDOW") is not None
            plain, plain_ms, n_plain = None, None, 0
            if not forced:
                plain = torch.empty(out_shape, dtype=torch.float64, device=dev)
                spin_up(plain)
                first_alloc_ms = plain_ms = ms_into(plain)
                n_plain = 1
                # ... and a few more plain allocations, all alive at once so that each sits on different memory
                held, cand = [plain], None
                for _ in range(max(0, args.alloc_tries - 1)):
                    try:
                        cand = torch.empty(out_shape, dtype=torch.float64, device=dev)
                    except RuntimeError:
                        break
                    held.append(cand)
                    n_plain += 1
                    ms_c = ms_into(cand, reps=6)
                    if ms_c < plain_ms:
                        plain, plain_ms = cand, ms_c
                del held, cand
                torch.cuda.empty_cache()
            free_b, _tot = torch.cuda.mem_get_info(dev)
            gib = min(args.place_gib, 0.75 * free_b / (1 << 30))
            try:
                pred = runtime.place_predictions(model, pop, d_theta, search_gib=gib, exhaustive=True)
                placed = "best window of a %.0f GiB arena (%.3f ms during the search)" % (gib, pred._pmx_owner.ms_per_pass)
                if plain is not None and ms_into(pred) > plain_ms:  # (boxes exist whose arenas hold no fast window at all)
                    pred = plain
                    placed = "best of %d plain allocations (no window of a %.0f GiB arena was faster)" % (n_plain, gib)
            except Exception as e:  # no virtual-memory API / not enough memory: best of a few plain allocations
                alloc_log = []
                pred = runtime.alloc_predictions(model, pop, d_theta, tries=args.alloc_tries, log=alloc_log)
                placed = "best of %d candidate allocations (%s)" % (len(alloc_log), type(e).__name__)
            del plain
            torch.cuda.empty_cache()
        elif want_gather:
            full = torch.empty((rows_all[-1][1], P), dtype=torch.float64, device=dev)
            pred = full[rows_all[rank][0]:rows_all[rank][1]]
            placed = "this rank's row block of the full [%d x %d] tensor" % (rows_all[-1][1], P)
        else:
            pred = torch.empty(out_shape, dtype=torch.float64, device=dev)
        spin_up()
        for _ in range(args.warmup):
            one_pass()
        torch.cuda.synchronize()

        # ---------------------------------------------------------------- timed region
        # HIP events on torch's current stream == the stream the kernels are enqueued on.  One pair per pass gives the
        # spread over the K passes; for launch-bound shapes (C2: 11 us per pass) the marker packets would double the pass
        # time, so those get a single pair around all K.
        per_pass_events = steps_per_pass_local >= 50_000_000
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1 if per_pass_events else 2)]
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(args.steps):
            one_pass()
            if per_pass_events:
                evs[i + 1].record()
        if not per_pass_events:
            evs[1].record()
        barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if per_pass_events:
            pass_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
        else:
            pass_ms = [evs[0].elapsed_time(evs[1]) / args.steps]
        kernel_name = runtime.last_kernel_name()
        # the device's write ceiling as measured on THIS box in THIS process (a linear streaming fill of the output buffer,
        # outside the timed region): what "attainable" means beside the 8 TB/s datasheet peak
        if pred is not None and pred.numel() >= (1 << 22):
            import ctypes as C

            from pharmsol_amd import _ffi
            gbs = C.c_double()
            flat_out = pred if pred.is_contiguous() else None
            if flat_out is None and getattr(pred, "_base", None) is not None and pred._base.is_contiguous():
                flat_out = pred._base  # (rows padded to a pitch: the whole allocation, padding included)
            if flat_out is not None:
                # (the fill overwrites the buffer: one more kernel pass below restores what the parity sample reads)
                _ffi.check(_ffi.lib().pmx_measure_write_ceiling(flat_out.data_ptr(), int(flat_out.numel()), 5,
                                                                torch.cuda.current_stream(dev).cuda_stream, C.byref(gbs)))
                attainable_gbs = gbs.value
                one_pass()
                torch.cuda.synchronize()

    kms = float(np.mean(pass_ms)) if pass_ms else 0.0
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    t_steps = torch.tensor([steps_per_pass_local], dtype=torch.int64, device=red_dev)
    t_kms = torch.tensor([kms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_steps, op=dist.ReduceOp.SUM)
        dist.all_reduce(t_kms, op=dist.ReduceOp.MAX)
    elapsed_max = float(t_el.item())
    steps_per_pass = int(t_steps.item())
    kernel_ms_mean = float(t_kms.item())

    # ---------------------------------------------------------------- optional all-gather (outside `value`)
    gather = None
    if not dry and world > 1 and args.gather and full is not None:
        try:
            from pharmsol_amd.distributed import all_gather_predictions

            class _Rows:  # what all_gather_predictions reads of a ShardedPopulation
                rows, n_observations_total = rows_all, rows_all[-1][1]

            def gather_pass():
                one_pass()
                all_gather_predictions(full, _Rows, comm=comm)

            for _ in range(max(1, args.warmup)):
                gather_pass()
            torch.cuda.synchronize()
            barrier()
            g0 = time.perf_counter()
            for _ in range(args.steps):
                gather_pass()
            barrier()
            torch.cuda.synchronize()
            t_g = torch.tensor([time.perf_counter() - g0], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t_g, op=dist.ReduceOp.MAX)
            gsec = float(t_g.item())
            gather = {"ms_per_step_with_gather": gsec / args.steps * 1e3,
                      "value_with_gather": steps_per_pass * args.steps / gsec,
                      "GB_received_per_rank": (full.numel() - pred.numel()) * 8 / 1e9,
                      "collective": "pmx_allgather_predictions: in-place ncclAllGather (equal blocks) / grouped ncclBroadcast (unequal)",
                      "rccl_ranks": world}
        except Exception as e:  # never let the optional leg kill the measured line
            gather = {"error": repr(e)[:300]}

    # ---------------------------------------------------------------- parity sample + CPU baseline (rank 0)
    cpu_baseline = None
    max_rel_err = None
    parity_ok = None
    if rank == 0 and not dry:
        import oracle  # test infrastructure: the checker / the timed CPU baseline, never the product path

        cores = oracle.max_threads()
        if args.loglik:
            run = lambda f, th: oracle.loglik(model, f, em, th)
        else:
            run = (lambda f, th: oracle.predict_batch(model, f, th)) if batch else (lambda f, th: oracle.predict(model, f, th))
        per_subject = max(flat.n_events / max(flat.n_subjects, 1) * (1 if batch else P), 1.0)

        def timed(n, first=0):
            f = flat.subject_slice(first, first + n)
            th = theta[first:first + n] if batch else theta
            c0 = time.perf_counter()
            w, _ = run(f, th)
            return w, time.perf_counter() - c0, f.n_events * (1 if batch else P)

        # parity sample: subjects from the head, the middle and the tail of this rank's shard, GPU vs oracle
        n_probe = min(flat.n_subjects, max(64, 4 * cores))
        obs_off = flat.observation_offsets()
        max_rel_err = 0.0
        probe_s = probe_steps = 0
        starts = sorted({0, max(0, flat.n_subjects // 2 - n_probe // 2), max(0, flat.n_subjects - n_probe)})
        for first in starts:
            want, secs, steps = timed(n_probe, first)
            if first == 0:
                probe_s, probe_steps = secs, steps
            if args.loglik:
                got = pred[first:first + want.shape[0]].cpu().numpy()
            else:
                r0 = int(obs_off[first])
                got = pred[r0:r0 + want.shape[0]].cpu().numpy()
            scale = max(float(np.nanmax(np.abs(want))), 1e-300)
            err = np.abs(got - want) / np.maximum(np.abs(want), 1e-12 * scale)
            e = float(np.max(err)) if err.size else 0.0
            max_rel_err = e if not (e <= max_rel_err) else max_rel_err  # (NaN sticks)
        parity_ok = bool(max_rel_err <= dtype_tol)
        if world == 1 and not args.no_cpu_baseline:
            # grow the sample until one run takes >= 1 s, then size the timed run for ~cpu_seconds of wall time
            n, secs, steps = n_probe, probe_s, probe_steps
            while secs < 1.0 and n < flat.n_subjects:
                n = min(flat.n_subjects, n * 4)
                _, secs, steps = timed(n)
            rate = steps / secs
            n_sample = int(min(flat.n_subjects, max(n, args.cpu_seconds * rate / per_subject)))
            if n_sample > n:
                _, secs, steps = timed(n_sample)
            else:
                n_sample = n
            cpu_baseline = {
                "value": steps / secs, "unit": "subject-event-steps/s", "cores": cores, "kind": "port",
                "sample": f"first {n_sample} subjects of the same population x {'own theta' if batch else str(P) + ' support points'} "
                          f"({steps} steps, {secs:.1f} s wall on {cores} threads); oracle/pmx_oracle.c, OpenMP over subjects "
                          f"(rayon-shaped loop nest of likelihood/matrix.rs:79-98), f64",
            }

    rc = 0
    if rank == 0:
        E_tot = flat.n_events
        O_tot = n_obs
        # per-launch algorithmic bytes of THIS rank's kernel (S*O = n_obs rows, S*E = n_events)
        b_alg = 8 * O_tot * (1 if batch else P) + 8 * theta.shape[0] * k + 26 * E_tot
        if args.loglik:  # S x P sums out, 24 B of {value, const, weight} per observation in
            b_alg = 8 * flat.n_subjects * P + 8 * theta.shape[0] * k + 26 * E_tot + 24 * O_tot
        # PMC counters of the same command, collected in separate rocprofv3 --pmc passes and committed (tools/pmc_run.sh,
        # tools/kernel_counters.py): per-launch HBM traffic and vector wave-instructions of the dominant kernel
        counters, counters_src = {}, None
        cpath = os.path.join(ROOT, "profiles", "kernel_counters.json")
        full_size = (world == 1 or not strong) and \
            S_arg == {"c3": 100_000, "c2": 10_000, "c4": 50_000, "c5": 200_000, "user": 50_000}[args.workload] and \
            P == {"c3": 1000, "c2": 1, "c4": 1, "c5": 512, "user": 256}[args.workload]
        if full_size and os.path.exists(cpath):
            try:
                counters = json.load(open(cpath)).get(counters_key, {})
                counters_src = "profiles/kernel_counters.json[%s] (%s)" % (counters_key, counters.get("source", "rocprofv3 --pmc"))
            except Exception:
                counters = {}
        t_k = max(kernel_ms_mean, 1e-9) * 1e-3
        if bound == "hbm":
            achieved = b_alg / t_k / 1e9
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
            valu = counters.get("valu_wave_insts_per_launch")
            if valu:  # the secondary ceiling: share of the FP64 vector issue slots at the 2.4 GHz peak clock
                roofline["valu_frac"] = valu / t_k / 1e9 / FP64_VALU_PEAK_GINST
                roofline["valu_insts"] = valu
            if first_alloc_ms:
                roofline["frac_first_allocation"] = b_alg / (first_alloc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                roofline["kernel_ms_first_allocation"] = first_alloc_ms
        else:
            valu = counters.get("valu_wave_insts_per_launch")
            achieved = (valu / t_k / 1e9) if valu else None
            roofline = {"bound": "fp64_valu", "achieved": achieved, "peak": FP64_VALU_PEAK_GINST, "unit": "Gwave-inst/s",
                        "frac": (achieved / FP64_VALU_PEAK_GINST) if achieved else None,
                        "valu_insts": valu, "valu_source": counters_src if valu else None,
                        "hbm_frac": b_alg / t_k / 1e9 / HBM_PEAK_GBS,
                        "note": "compute-bound: FP64-rate vector instructions issue once per 4 cycles per SIMD; "
                                "frac = share of the chip's issue slots at the 2.4 GHz peak clock"}
        roofline["attainable"] = attainable_gbs if not dry else None  # measured write ceiling of this device (GB/s): linear fill
        if attainable_gbs and roofline.get("bound") == "hbm":
            roofline["frac_of_attainable"] = roofline["achieved"] / attainable_gbs
        roofline.update({"traffic": counters.get("hbm_bytes_per_launch"),
                         "traffic_source": counters_src if counters.get("hbm_bytes_per_launch") else None,
                         "kernel": kernel_name, "kernel_ms": kernel_ms_mean, "algorithmic_bytes": b_alg})
        if pass_ms and len(pass_ms) > 1:
            srt = sorted(pass_ms)
            roofline["kernel_ms_spread"] = {"min": srt[0], "median": srt[len(srt) // 2], "max": srt[-1]}
        value = (steps_per_pass * args.steps / elapsed_max) if not dry else None
        line = {
            "metric": "subject_event_steps_per_sec", "value": value, "unit": "subject-event-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": label, "subjects_per_gpu": flat.n_subjects if strong else S_arg, "support_points": P,
                       "steps_per_pass": steps_per_pass, "kernel": kernel_name,
                       "status_bytes_written": not args.no_status,
                       "prediction_buffer": placed, "row_pitch_doubles": row_pitch, "sharding": f"subjects x{world}, no data-path collective",
                       "backend": backend if world > 1 else None,
                       "rccl_ranks": (dist.get_world_size() if (world > 1 and backend == "nccl") else None)},
            "max_rel_err_vs_cpu_ref": max_rel_err, "rel_err_tolerance": dtype_tol, "parity_ok": parity_ok,
            "roofline": roofline if not dry else None,
            "cpu_baseline": cpu_baseline,
        }
        if dry:
            line["dry_run"] = True
        if gather is not None:
            line["gather"] = gather

        def clean(o):  # NaN / inf are not JSON: a non-finite number on the line becomes null (parity_ok says why)
            if isinstance(o, float) and not np.isfinite(o):
                return None
            if isinstance(o, dict):
                return {a: clean(b) for a, b in o.items()}
            if isinstance(o, list):
                return [clean(b) for b in o]
            return o

        print(json.dumps(clean(line), allow_nan=False), flush=True)
        if parity_ok is False:
            print(f"[bench] PARITY FAILURE: max rel err {max_rel_err} vs the CPU oracle exceeds {dtype_tol}", file=sys.stderr)
            rc = 3
    if world > 1:
        t_rc = torch.tensor([rc], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t_rc, op=dist.ReduceOp.MAX)
        rc = int(t_rc.item())
        barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()

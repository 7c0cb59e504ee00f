#!/usr/bin/env python3
"""bench.py — subject-event-steps/s of the HIP prediction path on BASELINE.json's headline workload.

A "step" of this benchmark = one pass of the hot path (every subject x every support point, one
kernel launch) over one resident batch of synthetic input.  Workload at N=1 = C3, the configuration
the north-star target is quoted on: two-compartment IV analytical, 100k subjects x 1000 support
points, 8 events per subject (8e8 subject-event-steps per pass).  For N>1 (weak scaling) every rank
holds its own 100k-subject shard of an N x 100k population and the full theta grid; there is no
data-path collective (the reference's loop nest, likelihood/matrix.rs:79-98, has no exchange step).

Contract: W untimed warm-up passes, then exactly K timed passes bracketed by barrier +
torch.cuda.synchronize() on both sides; MAX over ranks; rank 0 prints ONE JSON line.
Inputs (population, theta) are resident in HBM before the timed region starts.

Extra objects on the line:
  roofline      dominant kernel vs the HBM roofline: achieved = algorithmic bytes per launch
                (8*S*O*P + 8*P*k + 26*S*E) / mean launch duration measured with HIP
                events on the launch stream inside the timed region; peak = 8 TB/s (MI355X HBM3E).
  cpu_baseline  the CPU oracle (a port of the reference algorithm, OpenMP over subjects like the
                reference's rayon loop) on a bounded sample of the same workload, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(S, O, P, k, E, cov_segments=0):
    """each array touched once."""
    return 8 * S * O * P + 8 * P * k + 26 * S * E + 24 * cov_segments


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--subjects", type=int, default=100_000, help="subjects per GPU (C3: 100000)")
    ap.add_argument("--support", type=int, default=1000, help="support points (C3: 1000)")
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c4", "c5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle sample")
    ap.add_argument("--gather", action="store_true", help="N>1: also time the optional RCCL all-gather")
    ap.add_argument("--no-status", action="store_true", help="do not write the per-pair status bytes")
    ap.add_argument("--loglik", action="store_true",
                    help="time the fused log-likelihood entry (pmx_loglik_device) instead of predictions: output S x P")
    ap.add_argument("--spin-up-ms", type=float, default=80.0,
                    help="untimed back-to-back passes before the warm-up, to bring the device clocks up after set-up")
    ap.add_argument("--place-gib", type=float, default=48.0,
                    help="size of the arena searched for the fastest window for the prediction matrix "
                         "(runtime.place_predictions); 0 = plain first allocation")
    ap.add_argument("--alloc-tries", type=int, default=8,
                    help="candidate allocations for the prediction matrix; the fastest is kept (runtime.alloc_predictions; "
                         "1 = take the first)")
    ap.add_argument("--ragged", action="store_true",
                    help="C3 with per-subject jittered sampling times (no shared design, no related step lengths)")
    ap.add_argument("--constant-cov", action="store_true",
                    help="c5: one wt value per subject instead of 2-4 interpolation knots (coefficients kept across PROPs)")
    ap.add_argument("--ld", type=int, default=0, help="leading dimension of the prediction rows (>= support points; 0 = dense)")
    ap.add_argument("--no-class", action="store_true",
                    help="A/B: disable the classed kernel (shared-design propagator reuse); every subject walks the generic kernel")
    args = ap.parse_args()
    if args.no_class:
        os.environ["PMX_DISABLE_CLASSING"] = "1"

    import numpy as np
    import torch
    import torch.distributed as dist

    from pharmsol_amd import runtime, synth
    from pharmsol_amd.distributed import ShardedPopulation, all_gather_predictions

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run "
                  f"--nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    backend = os.environ.get("PMX_BENCH_BACKEND", "nccl")
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

    # ---------------------------------------------------------------- workload (synthetic, seeded)
    S_local, P = args.subjects, args.support
    batch = False
    if args.workload in ("c3", "c2"):
        if args.workload == "c2":
            S_local, P = (10_000 if args.subjects == 100_000 else args.subjects), 1
        model = synth.model_two_cpt_iv()
        theta = synth.theta_c3(P) if args.workload == "c3" else synth.theta_c2()
        flat_global = synth.population_c23(S_local * world, ragged=args.ragged)
        label = f"C{'3' if args.workload == 'c3' else '2'}: two_compartments analytical, {S_local} subjects/GPU x {P} support points, 8 events/subject"
        dtype_tol = 1e-6
    elif args.workload == "c4":
        S_local = 50_000 if args.subjects == 100_000 else args.subjects
        model, flat_global, theta_all = synth.config_c4(S_local * world)
        batch, P = True, 1
        label = f"C4: ode one_cmt_iv RK4 h<=0.02, {S_local} subjects/GPU, irregular schedules, one theta per subject"
        dtype_tol = 1e-4
    else:
        S_local = 200_000 if args.subjects == 100_000 else args.subjects
        P = 512 if args.support == 1000 else args.support
        model = synth.model_three_cpt_abs_wt()
        theta = synth.theta_c5(P)
        flat_global = synth.population_c5(S_local * world, constant_wt=args.constant_cov)
        label = f"C5: three_compartments_with_absorption + {'subject-constant' if args.constant_cov else 'time-varying'} wt covariate, {S_local} subjects/GPU x {P} support points"
        dtype_tol = 1e-6
    sh = ShardedPopulation(flat_global, rank, world)
    flat = sh.local
    if batch:
        s0, s1 = sh.bounds[rank]
        theta = theta_all[s0:s1]
    k = theta.shape[1]

    em = None
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
    placed = "first allocation"
    ld = max(args.ld, P) if (args.ld and not batch and not args.loglik) else None
    if ld is not None:  # rows padded to a leading dimension (ld_pred of pmx_predict_device); same bytes written
        pred = torch.empty((n_obs, ld), dtype=torch.float64, device=dev)[:, :P]
    elif not batch and not args.loglik and args.place_gib > 0:
        # where the matrix lands in HBM changes the write rate of the row-strided stream by up to 25 %: the library maps an
        # arena, times the kernel into every window of it, keeps the best window and returns the rest (set-up, outside the
        # timed region; the buffer is then reused by every pass)
        try:
            pred = runtime.place_predictions(model, pop, d_theta, search_gib=args.place_gib)
            placed = "best window of a %g GiB arena (%.3f ms during the search)" % (args.place_gib, pred._pmx_owner.ms_per_pass)
        except Exception as e:  # no virtual-memory API / not enough memory: best of a few plain allocations
            alloc_log = []
            pred = runtime.alloc_predictions(model, pop, d_theta, tries=args.alloc_tries, log=alloc_log)
            placed = "best of %d candidate allocations (%s)" % (len(alloc_log), type(e).__name__)
    else:
        pred = torch.empty((n_obs,) if batch else ((pop.n_subjects, P) if args.loglik else (n_obs, P)),
                           dtype=torch.float64, device=dev)
    em_c = em.to_c(model) if em is not None else None
    status = None if args.no_status else torch.zeros((pop.n_subjects,) if batch else (pop.n_subjects, P),
                                                     dtype=torch.uint8, device=dev)
    steps_per_pass_local = pop.n_events * (1 if batch else P)

    def one_pass():
        if args.loglik:
            runtime.loglik(model, pop, em_c, d_theta, ll=pred, status=status, want_status=not args.no_status)
        else:
            runtime.predict(model, pop, d_theta, pred=pred, status=status, batch=batch, want_status=not args.no_status)

    # Spin-up (untimed, before the W warm-up passes): after the set-up phase (allocations, frees, host work) the device
    # sits idle for a while and comes back with reduced clocks; the per-dispatch trace shows the pass time falling from
    # 1.36 ms to its steady 0.86-0.91 ms over ~25 ms of back-to-back work (profiles/r01_dispatch_ramp.txt).  Passes
    # are ~1 ms, so a fixed ~80 ms of them is ample and costs nothing measurable.
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spin_up_ms * 1e-3:
        for _ in range(8):
            one_pass()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize()

    # ---------------------------------------------------------------- timed region
    # one event pair around the K passes (torch's current stream == the stream the kernels are enqueued on): a pair per
    # pass would put two marker packets next to every launch, which doubles the pass time of the launch-bound C2 shape
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        one_pass()
    ev1.record()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = [ev0.elapsed_time(ev1) / args.steps]
    kernel_name = runtime.last_kernel_name()

    t_el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    t_steps = torch.tensor([steps_per_pass_local], dtype=torch.int64, device=red_dev)
    t_kms = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_steps, op=dist.ReduceOp.SUM)
        dist.all_reduce(t_kms, op=dist.ReduceOp.MAX)
    elapsed_max = float(t_el.item())
    steps_per_pass = int(t_steps.item())
    kernel_ms_mean = float(t_kms.item())

    # ---------------------------------------------------------------- optional all-gather (outside `value`)
    gather = None
    if world > 1 and args.gather and not batch:
        try:
            torch.cuda.synchronize()
            barrier()
            g0 = time.perf_counter()
            full = all_gather_predictions(pred, sh)
            torch.cuda.synchronize()
            barrier()
            gms = (time.perf_counter() - g0) * 1e3
            gather = {"ms": gms, "GB_per_rank_out": full.numel() * 8 / 1e9, "backend": backend}
            del full
        except Exception as e:  # never let the optional leg kill the measured line
            gather = {"error": repr(e)[:200]}

    # ---------------------------------------------------------------- parity sample + CPU baseline (rank 0)
    cpu_baseline = None
    max_rel_err = None
    if rank == 0:
        import oracle  # test infrastructure: the checker / the timed CPU baseline, never the product path

        cores = oracle.max_threads()
        if args.loglik:
            run = lambda f, th: oracle.loglik(model, f, em, th)
        else:
            run = (lambda f, th: oracle.predict_batch(model, f, th)) if batch else (lambda f, th: oracle.predict(model, f, th))
        per_subject = max(flat.n_events / max(flat.n_subjects, 1) * (1 if batch else P), 1.0)

        def timed(n):
            f = flat.subject_slice(0, n)
            th = theta[:n] if batch else theta
            c0 = time.perf_counter()
            w, _ = run(f, th)
            return w, time.perf_counter() - c0, f.n_events * (1 if batch else P)

        # parity sample: the first subjects of this rank's shard, GPU vs oracle
        n_probe = min(flat.n_subjects, max(64, 4 * cores))
        want, probe_s, probe_steps = timed(n_probe)
        got = pred[: want.shape[0]].cpu().numpy()
        scale = max(float(np.nanmax(np.abs(want))), 1e-300)
        max_rel_err = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-12 * scale)))
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

    if rank == 0:
        E_tot = flat.n_events
        O_tot = n_obs
        # per-launch algorithmic bytes of THIS rank's kernel (S*O = n_obs rows, S*E = n_events)
        b_alg = 8 * O_tot * (1 if batch else P) + 8 * theta.shape[0] * k + 26 * E_tot
        if args.loglik:  # S x P sums out, 24 B of {value, const, weight} per observation in
            b_alg = 8 * pop.n_subjects * P + 8 * theta.shape[0] * k + 26 * E_tot + 24 * O_tot
        achieved = b_alg / (kernel_ms_mean * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp) and args.workload == "c3" and S_local == 100_000 and P == 1000:
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        value = steps_per_pass * args.steps / elapsed_max
        line = {
            "metric": "subject_event_steps_per_sec", "value": value, "unit": "subject-event-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": label, "subjects_per_gpu": S_local, "support_points": P,
                       "steps_per_pass": steps_per_pass, "kernel": kernel_name,
                       "status_bytes_written": not args.no_status,
                       "prediction_buffer": placed, "sharding": f"subjects x{world}, no collective"},
            "max_rel_err_vs_cpu_ref": max_rel_err, "rel_err_tolerance": dtype_tol,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel_name, "kernel_ms": kernel_ms_mean, "algorithmic_bytes": b_alg},
            "cpu_baseline": cpu_baseline,
        }
        if gather is not None:
            line["gather"] = gather
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Randomised combinations of everything the path supports (structure x CL form x pm_ indexing x init x lag x fa x
multi-occasion x shared/ragged designs x covariate-derived parameters x lane mapping x prediction / log-likelihood),
GPU against the CPU oracle.  Each case is seeded; a failure prints its recipe."""
import os

import numpy as np
import pytest

import oracle
from pharmsol_amd import (ODE, Analytical, AssayErrorModel, AssayErrorModels, Data, ErrorPoly, Pow, Ratio, Scaled, Subject,
                          _abi, analytical, bolus, infusion, runtime, synth)
from tests import models

pytestmark = pytest.mark.gpu
# PMX_FUZZ_OFFSET shifts every family's random stream: fresh cases beyond the seeds 0..N-1 of the committed / recorded runs
OFFSET = int(os.environ.get("PMX_FUZZ_OFFSET", "0"))

STRUCTS = {  # name -> (n states, n kernel params, central state, theta builder)
    "one_compartment": (1, 1, 0), "one_compartment_cl": (1, 2, 0),
    "one_compartment_with_absorption": (2, 2, 1), "one_compartment_cl_with_absorption": (2, 3, 1),
    "two_compartments": (2, 3, 0), "two_compartments_cl": (2, 4, 0),
    "two_compartments_with_absorption": (3, 4, 1), "two_compartments_cl_with_absorption": (3, 5, 1),
    "three_compartments": (3, 5, 0), "three_compartments_cl": (3, 6, 0),
    "three_compartments_with_absorption": (4, 6, 1), "three_compartments_cl_with_absorption": (4, 7, 1),
}


def kernel_theta(name, n, rng):
    """Kernel-order parameters with real eigenvalues (micro-constants from the C3/C5 generators, CL forms derived)."""
    t2, t3 = synth.theta_c3(n, synth.SplitMix64(int(rng.integers(1 << 30)))), synth.theta_c5(n, synth.SplitMix64(int(rng.integers(1 << 30))))
    ke, kcp, kpc = t2[:, 0], t2[:, 1], t2[:, 2]
    ka, k10, k12, k13, k21, k31 = (t3[:, i] for i in range(6))
    v = rng.uniform(10, 60, n)
    if name == "one_compartment":
        return np.stack([ke], 1)
    if name == "one_compartment_cl":
        return np.stack([ke * v, v], 1)
    if name == "one_compartment_with_absorption":
        return np.stack([ka, ke], 1)
    if name == "one_compartment_cl_with_absorption":
        return np.stack([ka, ke * v, v], 1)
    if name == "two_compartments":
        return np.stack([ke, kcp, kpc], 1)
    if name == "two_compartments_cl":  # cl, q, vc, vp
        return np.stack([ke * v, kcp * v, v, kcp * v / kpc], 1)
    if name == "two_compartments_with_absorption":  # ke, ka, kcp, kpc
        return np.stack([ke, ka, kcp, kpc], 1)
    if name == "two_compartments_cl_with_absorption":  # ka, cl, q, vc, vp
        return np.stack([ka, ke * v, kcp * v, v, kcp * v / kpc], 1)
    if name == "three_compartments":
        return np.stack([k10, k12, k13, k21, k31], 1)
    if name == "three_compartments_cl":  # cl, q2, q3, vc, v2, v3
        return np.stack([k10 * v, k12 * v, k13 * v, v, k12 * v / k21, k13 * v / k31], 1)
    if name == "three_compartments_with_absorption":
        return np.stack([ka, k10, k12, k13, k21, k31], 1)
    return np.stack([ka, k10 * v, k12 * v, k13 * v, v, k12 * v / k21, k13 * v / k31], 1)


def build_case(seed):
    rng = np.random.default_rng(seed + OFFSET)
    name = list(STRUCTS)[int(rng.integers(0, len(STRUCTS)))]
    ns, nk, central = STRUCTS[name]
    pm = bool(rng.random() < 0.2)
    use_lag = bool(rng.random() < 0.3) and not pm
    use_fa = bool(rng.random() < 0.3)
    use_init = bool(rng.random() < 0.3)
    shared = bool(rng.random() < 0.4)
    multi = bool(rng.random() < 0.4)
    n_sub = int(rng.integers(3, 40))
    n_support = int(rng.choice([1, 3, 9, 40, 64, 70, 72, 300, 304]))
    batch = bool(rng.random() < 0.15)
    # theta layout: kernel params | v | lag | fa | init
    cols = nk
    v_col = cols
    cols += 1
    lag_col = fa_col = init_col = None
    if use_lag:
        lag_col, cols = cols, cols + 1
    if use_fa:
        fa_col, cols = cols, cols + 1
    if use_init:
        init_col, cols = cols, cols + 1
    off = 1 if pm else 0
    m = Analytical.new(("pm_" if pm else "") + name, {0: Ratio(central + off, v_col)}, nparams=cols,
                       init={central + off: init_col} if use_init else None,
                       lag={off: lag_col} if use_lag else None, fa={off: fa_col} if use_fa else None)
    m = m.with_nstates(ns + off).with_ndrugs(1 + off).with_nout(1)
    subs = []
    if shared:
        proto = models.random_subject(rng, multi_occasion=multi)
        for i in range(n_sub):
            b = Subject.builder(f"s{i}")
            for oi, occ in enumerate(proto.occasions):
                if oi:
                    b = b.reset()
                for ev in occ.events:
                    if hasattr(ev, "duration"):
                        b = b.infusion(ev.time, ev.amount * (1 + 0.01 * i), off, ev.duration)
                    elif hasattr(ev, "amount"):
                        b = b.bolus(ev.time, ev.amount * (1 + 0.01 * i), off)
                    else:
                        b = b.missing_observation(ev.time, 0)
            subs.append(b.build())
    else:
        for _ in range(n_sub):
            s = models.random_subject(rng, multi_occasion=multi)
            for occ in s.occasions:
                for ev in occ.events:
                    if hasattr(ev, "input"):
                        ev.input = off
            subs.append(s)
        if rng.random() < 0.3:
            subs.insert(int(rng.integers(0, len(subs))), Subject.builder("empty").build())
    n = len(subs) if batch else n_support
    th = [kernel_theta(name, n, rng), rng.uniform(10, 80, (n, 1))]
    if use_lag:
        th.append(np.round(rng.uniform(-1, 3, (n, 1)) * 2) / 2)  # (negative: the bolus moves earlier, structs.rs:629-634)
    if use_fa:
        th.append(rng.uniform(0.3, 1.0, (n, 1)))
    if use_init:
        th.append(rng.uniform(0, 50, (n, 1)))
    theta = np.concatenate(th, axis=1)
    recipe = dict(seed=seed, structure=name, pm=pm, lag=use_lag, fa=use_fa, init=use_init, shared=shared, multi=multi,
                  subjects=len(subs), support=n_support, batch=batch)
    return m, subs, theta, batch, recipe


def _denormal_cdf_pairs(flat, pred, obs_off):
    """(subject, support point) pairs holding a censored row 36.5-39 sigma on the empty side of the prediction: ln(cdf) of
    a cdf that is a DENORMAL number (erfc(26..27.3) = 1e-300..5e-324) - two or three significant bits in the reference
    (statrs), the oracle (glibc) and the device (ocml) alike, each rounding differently: sums there agree to ~1e-4 only
    (fuzz seeds 822, 3714).  The censored folds inside the normal range are pinned at 1e-9 by tests/test_gpu_likelihood.py."""
    is_obs = flat.ev_kind == _abi.PMX_EV_OBSERVATION
    y = flat.ev_value[is_obs]
    cens = flat.ev_censor[is_obs].astype(np.float64)
    poly = flat.ev_errorpoly[is_obs]
    own = ~np.isnan(poly[:, 0])
    # (the families' error models: output 0 additive, poly (0.05, 0.1), lambda 0.1; output 1 proportional, poly
    # (0.02, 0.15, 0.001), gamma 1.3; an observation's own polynomial replaces the model's)
    out1 = flat.ev_io[is_obs] == 1
    c0 = np.where(own, poly[:, 0], np.where(out1, 0.02, 0.05))
    c1 = np.where(own, poly[:, 1], np.where(out1, 0.15, 0.1))
    c2 = np.where(own, poly[:, 2], np.where(out1, 0.001, 0.0))
    alpha = c0 + c1 * y + c2 * y * y
    sigma = np.where(out1, 1.3 * alpha, np.sqrt(alpha ** 2 + 0.1 ** 2))
    with np.errstate(invalid="ignore"):
        z = (y[:, None] - pred) / sigma[:, None] * cens[:, None]  # BLOQ (+1): far below the prediction = z << 0
        row_bad = (cens[:, None] != 0) & (z < -36.5) & np.isfinite(z)
    out = np.zeros((len(obs_off) - 1, pred.shape[1]), dtype=bool)
    for s_ in range(len(obs_off) - 1):
        out[s_] = row_bad[obs_off[s_]:obs_off[s_ + 1]].any(axis=0)
    return out


@pytest.mark.parametrize("seed", range(int(os.environ.get("PMX_FUZZ_ANALYTICAL", "120"))))  # (more seeds: set the variable)
def test_random_analytical_configuration(seed):
    import torch

    m, subs, theta, batch, recipe = build_case(1000 + seed)
    flat = m.flatten(Data(subs))
    pop = runtime.DevicePopulation(flat, 0)
    # a dirty status buffer: whichever way the launch clears it (memset or the kernels themselves), it must come back exact
    dirty = torch.full((flat.n_subjects,) if batch else (flat.n_subjects, theta.shape[0]), 201, dtype=torch.uint8, device="cuda")
    pred, st = runtime.predict(m, pop, np.ascontiguousarray(theta), batch=batch, status=dirty)
    torch.cuda.synchronize()
    got, st = pred.cpu().numpy(), st.cpu().numpy()
    want, wst = (oracle.predict_batch if batch else oracle.predict)(m, flat, theta)
    assert got.shape == want.shape, recipe
    np.testing.assert_array_equal(st, wst, err_msg=str(recipe))
    ok = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), ok, err_msg=str(recipe))
    if ok.any():
        scale = np.maximum(np.abs(want[ok]), 1e-9 * np.abs(want[ok]).max() + 1e-300)
        err = (np.abs(got[ok] - want[ok]) / scale).max()
        assert err < 1e-6, (err, recipe, runtime.last_kernel_name())
    if not batch and seed % 3 == 0 and flat.n_observations:  # the same case through the fused log-likelihood
        rng = np.random.default_rng(seed)
        vals = np.abs(np.where(np.isfinite(want[:, 0]), want[:, 0], 1.0)) * np.exp(rng.normal(0, 0.2, want.shape[0])) + 0.05
        vals[rng.random(vals.shape) < 0.2] = np.nan
        order_ok = all(list(map(lambda e: e.time, o.events)) == sorted(e.time for e in o.events) for s in subs for o in s.occasions)
        if order_ok:
            flat.ev_value = flat.ev_value.copy()
            flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION] = vals
            if seed % 6 == 0:  # BLOQ / ALOQ rows and per-observation error polynomials through the generic / PAIR / lag folds
                from tests.test_gpu_likelihood import _censor_some

                flat = _censor_some(flat, rng)
            em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
            pop2 = runtime.DevicePopulation(flat, 0)
            ll, lst = runtime.loglik(m, pop2, em, np.ascontiguousarray(theta))
            torch.cuda.synchronize()
            wll, wlst = oracle.loglik(m, flat, em, theta)
            np.testing.assert_array_equal(lst.cpu().numpy(), wlst, err_msg=str(recipe))
            okl = np.isfinite(wll)
            gl = ll.cpu().numpy()
            np.testing.assert_array_equal(np.isfinite(gl), okl, err_msg=str(recipe))
            if seed % 6 == 0:
                okl &= ~_denormal_cdf_pairs(flat, want, pop2.observation_offsets())
            if okl.any():
                assert (np.abs(gl[okl] - wll[okl]) / np.maximum(np.abs(wll[okl]), 1.0)).max() < 1e-6, recipe


ODE_MODELS = {  # name -> (n states, n diffeq params, central)
    "one_cmt_iv": (1, 1, 0), "one_cmt_oral": (2, 2, 1), "two_cmt_iv": (2, 3, 0), "two_cmt_oral": (3, 4, 1),
    "three_cmt_iv": (3, 5, 0), "three_cmt_oral": (4, 6, 1), "one_cmt_mm": (1, 3, 0),
}


@pytest.mark.parametrize("seed", range(int(os.environ.get("PMX_FUZZ_ODE", "40"))))
def test_random_ode_configuration(seed):
    import torch

    rng = np.random.default_rng(5000 + seed + OFFSET)
    name = list(ODE_MODELS)[int(rng.integers(0, len(ODE_MODELS)))]
    ns, nk, central = ODE_MODELS[name]
    use_lag, use_fa = bool(rng.random() < 0.4), bool(rng.random() < 0.3)
    adaptive = bool(rng.random() < 0.4)
    batch = bool(rng.random() < 0.2)
    n_support = int(rng.choice([2, 20, 40, 90]))
    mm = name == "one_cmt_mm"
    cols = nk + (0 if mm else 1)  # the MM body's third parameter is its volume
    v_col = 2 if mm else nk
    lag_col = fa_col = None
    if use_lag:
        lag_col, cols = cols, cols + 1
    if use_fa:
        fa_col, cols = cols, cols + 1
    m = ODE.new(name, {0: Ratio(central, v_col)}, nparams=cols, lag={0: lag_col} if use_lag else None,
                fa={0: fa_col} if use_fa else None, h_max=0.05).with_nstates(ns).with_ndrugs(1).with_nout(1)
    stiff = adaptive and seed % 3 == 2  # (every third adaptive case: the L-stable stepper behind the same step control)
    if adaptive:
        m = m.with_step(4.0).with_solver("ros2" if stiff else "dopri5").with_tolerances(*((1e-6, 1e-7) if stiff else (1e-8, 1e-8)))
    subs = [models.random_subject(rng, multi_occasion=bool(rng.random() < 0.3)) for _ in range(int(rng.integers(3, 25)))]
    n = len(subs) if batch else n_support
    if mm:
        th = [np.stack([rng.uniform(5, 30, n), rng.uniform(1, 10, n), rng.uniform(10, 40, n)], 1)]
    else:
        src = {1: synth.theta_c3(n)[:, :1], 2: np.stack([synth.theta_c5(n)[:, 0], synth.theta_c3(n)[:, 0]], 1),
               3: synth.theta_c3(n)[:, :3],
               4: np.concatenate([synth.theta_c3(n)[:, :1], synth.theta_c5(n)[:, :1], synth.theta_c3(n)[:, 1:3]], 1),
               5: synth.theta_c5(n)[:, 1:6], 6: synth.theta_c5(n)[:, :6]}[nk]
        th = [src, rng.uniform(10, 80, (n, 1))]
    if use_lag:
        th.append(np.round(rng.uniform(-1, 3, (n, 1)) * 2) / 2)  # (negative: the bolus moves earlier, structs.rs:629-634)
    if use_fa:
        th.append(rng.uniform(0.3, 1.0, (n, 1)))
    theta = np.concatenate(th, axis=1)
    recipe = dict(seed=seed, model=name, lag=use_lag, fa=use_fa, adaptive=adaptive, stiff=stiff, batch=batch, support=n_support)
    flat = m.flatten(Data(subs))
    pop = runtime.DevicePopulation(flat, 0)
    pred, st = runtime.predict(m, pop, np.ascontiguousarray(theta), batch=batch)
    torch.cuda.synchronize()
    got, st = pred.cpu().numpy(), st.cpu().numpy()
    want, wst = (oracle.predict_batch if batch else oracle.predict)(m, flat, theta)
    np.testing.assert_array_equal(st, wst, err_msg=str(recipe))
    ok = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), ok, err_msg=str(recipe))
    if ok.any():
        scale = np.maximum(np.abs(want[ok]), 1e-3 * np.abs(want[ok]).max() + 1e-300)
        err = (np.abs(got[ok] - want[ok]) / scale).max()
        # (adaptive: same method and tolerances on both sides; FMA contraction may move a step boundary, so agreement is at
        # the solver's tolerance - ROS2's steps, sized by a first-order estimate, are many and each is accepted at <= rtol)
        assert err < ((5e-5 if stiff else 2e-6) if adaptive else 1e-9), (err, recipe, runtime.last_kernel_name())


STATE_NAMES = {1: ["central"], 2: ["central", "periph"], 3: ["central", "periph1", "periph2"]}


@pytest.mark.parametrize("seed", range(int(os.environ.get("PMX_FUZZ_COVARIATE", "60"))))
def test_random_covariate_model(seed):
    """Covariate-derived rate constants (and, half the time, a covariate-derived volume) on every structure, CL forms
    included: the per-segment coefficient rebuild (Newton reciprocals, single-precision-seeded cube root) against the
    oracle's IEEE arithmetic, GRID and PAIR lane mappings, one or two occasions, constant and interpolated covariates."""
    import torch
    from pharmsol_amd import Lin

    rng = np.random.default_rng(9000 + seed + OFFSET)
    name = list(STRUCTS)[int(rng.integers(0, len(STRUCTS)))]
    ns, nk, central = STRUCTS[name]
    knames = _abi.KERNEL_PARAMETER_NAMES[name]
    absorb = "absorption" in name
    states = (["gut"] if absorb else []) + STATE_NAMES[ns - (1 if absorb else 0)]
    # the elimination parameter (ke / k10 / cl) scales with weight; the volume behind the output may too
    elim = next(k for k in knames if k in ("ke", "k10", "cl"))
    params = [k + "0" if k == elim else k for k in knames]
    derived = {elim: Scaled(elim + "0", (Pow("wt", 70.0, 0.75),) if rng.random() < 0.6 else (Lin("wt", 70.0, 0.004),))}
    vol_in_kernel = next((k for k in knames if k in ("v", "vc")), None)
    if vol_in_kernel is None:
        params.append("v")
        out_vol = "v"
        if rng.random() < 0.5:
            params[-1] = "v0"
            derived["v"] = Scaled("v0", (Pow("wt", 70.0, 1.0),))
    else:
        out_vol = vol_in_kernel
    # (every fourth case: covariates bound at the absolute segment end, the DSL run-time's rule, src/dsl/native.rs:1907-1916)
    m = analytical(name=f"fz{seed}", params=params, derived=derived, covariates=["wt"], structure=name, states=states,
                   outputs=["cp"], routes=[bolus("dose", states[0]), infusion("iv", "central")], out={"cp": Ratio("central", out_vol)},
                   cov_time="segment_end_abs" if seed % 4 == 3 else "segment_dt")
    subs = []
    n_sub = int(rng.integers(3, 30))
    for i in range(n_sub):
        b = Subject.builder(f"s{i}").covariate("wt", 0.0, float(rng.uniform(45, 110)))
        if rng.random() < 0.5:
            b = b.covariate("wt", float(rng.uniform(5, 30)), float(rng.uniform(45, 110)))
        b = b.bolus(0.0, float(rng.uniform(50, 300)), "dose")
        # (every third case: nobody is infused - the shape that sends the three-compartment structures to the matrix-free
        # walker pmx_analytical_dyn3; a repeated segment length now and then exercises its kept segments)
        if seed % 3 != 1 and rng.random() < 0.5:
            b = b.infusion(float(rng.uniform(0.5, 6)), float(rng.uniform(50, 200)), "iv", float(rng.uniform(0.5, 3)))
        if seed % 3 == 1 and i % 2 == 0:
            for t in (6.0, 12.0, 18.0, 24.0, 36.0):
                b = b.missing_observation(t, "cp")
        for t in np.sort(rng.uniform(0.2, 40.0, int(rng.integers(2, 9)))):
            b = b.missing_observation(float(t), "cp")
        if rng.random() < 0.3:
            b = b.reset().covariate("wt", 0.0, float(rng.uniform(45, 110))).bolus(0.0, 80.0, "dose")
            for t in np.sort(rng.uniform(0.5, 20.0, 3)):
                b = b.missing_observation(float(t), "cp")
        subs.append(b.build())
    n = int(rng.choice([2, 9, 40, 70, 130]))
    kt = kernel_theta(name, n, rng)
    theta = kt if vol_in_kernel is not None else np.concatenate([kt, rng.uniform(10, 80, (n, 1))], axis=1)
    recipe = dict(seed=seed, structure=name, derived=list(derived), subjects=n_sub, support=n)
    flat = m.flatten(Data(subs))
    pop = runtime.DevicePopulation(flat, 0)
    pred, st = runtime.predict(m, pop, np.ascontiguousarray(theta))
    torch.cuda.synchronize()
    got, st = pred.cpu().numpy(), st.cpu().numpy()
    want, wst = oracle.predict(m, flat, theta)
    np.testing.assert_array_equal(st, wst, err_msg=str(recipe))
    ok = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), ok, err_msg=str(recipe))
    if ok.any():
        scale = np.maximum(np.abs(want[ok]), 1e-9 * np.abs(want[ok]).max() + 1e-300)
        err = (np.abs(got[ok] - want[ok]) / scale).max()
        assert err < 1e-6, (err, recipe, runtime.last_kernel_name())


@pytest.mark.parametrize("seed", range(int(os.environ.get("PMX_FUZZ_CLASSED", "60"))))
def test_random_classed_design_with_outputs_and_likelihoods(seed):
    """Populations that share a program SHAPE: one prototype schedule, every subject's times stretched by its own factor
    (loose classes: step lengths per member) or not at all (exact classes), one or two outputs (a second state / another
    volume), predictions and the fused log-likelihood with missing, censored and own-polynomial rows in any mix -
    the classed kernel's per-step fold with and without tests (pmx_ll_prepare_chunks' flag word), its partial last chunk,
    the secondary-output volume path."""
    import torch

    from tests.test_gpu_likelihood import _censor_some

    rng = np.random.default_rng(31000 + seed + OFFSET)
    name = list(STRUCTS)[int(rng.integers(0, len(STRUCTS)))]
    ns, nk, central = STRUCTS[name]
    nout = int(rng.integers(1, 3))
    stretch = bool(rng.random() < 0.6)
    multi = bool(rng.random() < 0.3)
    n_sub = int(rng.integers(9, 60))
    n = int(rng.choice([8, 33, 40, 64, 70, 71, 129, 130]))
    outs = {0: Ratio(central, nk)}
    if nout == 2:
        outs[1] = Ratio(int(rng.integers(0, ns)), nk + 1)
    m = Analytical.new(name, outs, nparams=nk + 2).with_nstates(ns).with_ndrugs(1).with_nout(nout)
    proto = models.random_subject(rng, multi_occasion=multi, ties=bool(rng.random() < 0.5))
    subs = []
    for i in range(n_sub):
        f = float(rng.uniform(0.85, 1.2)) if stretch else 1.0
        b = Subject.builder(f"s{i}")
        for oi, occ in enumerate(proto.occasions):
            if oi:
                b = b.reset()
            for ev in occ.events:
                if hasattr(ev, "duration"):
                    b = b.infusion(ev.time * f, ev.amount * (1 + 0.01 * i), 0, ev.duration * f)
                elif hasattr(ev, "amount"):
                    b = b.bolus(ev.time * f, ev.amount * (1 + 0.01 * i), 0)
                else:
                    b = b.missing_observation(ev.time * f, 0)
        subs.append(b.build())
    # (the output of each observation is part of the shape: drawn once per prototype position)
    flat = m.flatten(Data(subs))
    if nout == 2:
        is_obs = flat.ev_kind == _abi.PMX_EV_OBSERVATION
        per_subject = int(is_obs.sum()) // n_sub
        pattern = rng.integers(0, 2, per_subject).astype(flat.ev_io.dtype)
        io = flat.ev_io.copy()
        io[is_obs] = np.tile(pattern, n_sub)
        flat.ev_io = io
    theta = np.concatenate([kernel_theta(name, n, rng), rng.uniform(10, 80, (n, 2))], axis=1)
    recipe = dict(seed=seed, structure=name, nout=nout, stretch=stretch, multi=multi, subjects=n_sub, support=n)
    pop = runtime.DevicePopulation(flat, 0)
    pred, st = runtime.predict(m, pop, np.ascontiguousarray(theta))
    torch.cuda.synchronize()
    if n >= 33 and flat.n_events > 0:  # every subject shares the shape: one class, never the generic walker
        assert runtime.last_kernel_name().startswith("pmx_analytical_classed"), (recipe, runtime.last_kernel_name())
    got, st = pred.cpu().numpy(), st.cpu().numpy()
    want, wst = oracle.predict(m, flat, theta)
    np.testing.assert_array_equal(st, wst, err_msg=str(recipe))
    ok = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), ok, err_msg=str(recipe))
    if ok.any():
        scale = np.maximum(np.abs(want[ok]), 1e-9 * np.abs(want[ok]).max() + 1e-300)
        assert (np.abs(got[ok] - want[ok]) / scale).max() < 1e-6, (recipe, runtime.last_kernel_name())
    if not flat.n_observations:
        return
    vals = np.abs(np.where(np.isfinite(want[:, 0]), want[:, 0], 1.0)) * np.exp(rng.normal(0, 0.2, want.shape[0])) + 0.05
    vals[rng.random(vals.shape) < float(rng.choice([0.0, 0.0, 0.05, 0.3]))] = np.nan
    flat.ev_value = flat.ev_value.copy()
    flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION] = vals
    if rng.random() < 0.5:
        flat = _censor_some(flat, rng, frac_bloq=float(rng.choice([0.0, 0.1])), frac_aloq=float(rng.choice([0.0, 0.1])),
                            frac_poly=float(rng.choice([0.0, 0.2])))
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
    if nout == 2:
        em = em.add(1, AssayErrorModel.proportional(ErrorPoly(0.02, 0.15, 0.001, 0.0), 1.3))
    pop2 = runtime.DevicePopulation(flat, 0)
    ll, lst = runtime.loglik(m, pop2, em, np.ascontiguousarray(theta))
    torch.cuda.synchronize()
    wll, wlst = oracle.loglik(m, flat, em, theta)
    np.testing.assert_array_equal(lst.cpu().numpy(), wlst, err_msg=str(recipe))
    okl = np.isfinite(wll)
    gl = ll.cpu().numpy()
    np.testing.assert_array_equal(np.isfinite(gl), okl, err_msg=str(recipe))
    if flat.ev_censor is not None:
        okl &= ~_denormal_cdf_pairs(flat, want, pop2.observation_offsets())
    if okl.any():
        assert (np.abs(gl[okl] - wll[okl]) / np.maximum(np.abs(wll[okl]), 1.0)).max() < 1e-6, (recipe, runtime.last_kernel_name())


ODE_TWIN = {  # analytical structure -> (built-in diffeq, its states)
    "one_compartment": ("one_cmt_iv", ["central"]), "one_compartment_with_absorption": ("one_cmt_oral", ["gut", "central"]),
    "two_compartments": ("two_cmt_iv", ["central", "periph"]),
    "two_compartments_with_absorption": ("two_cmt_oral", ["gut", "central", "periph"]),
    "three_compartments": ("three_cmt_iv", ["central", "periph1", "periph2"]),
}


@pytest.mark.parametrize("seed", range(int(os.environ.get("PMX_FUZZ_COVARIATE_CLASSED", "12"))))
def test_random_covariate_model_on_a_shared_shape_and_its_ode_twin(seed):
    """Covariate-derived rate constants / volumes on populations that share a program shape (classed<dyn> for the one- and
    two-state structures, the kept-propagator walker above), predictions and log-likelihoods with missing rows; and the
    same model as a built-in ODE body with derived-parameter descriptors (expand/ode.rs:126-185: run-time-compiled) against
    the RK4 oracle."""
    import torch
    from pharmsol_amd import Lin, ode

    rng = np.random.default_rng(47000 + seed + OFFSET)
    name = list(ODE_TWIN)[int(rng.integers(0, len(ODE_TWIN)))]
    ns, nk, central = STRUCTS[name]
    diffeq, states = ODE_TWIN[name]
    knames = _abi.KERNEL_PARAMETER_NAMES[name]
    elim = next(k for k in knames if k in ("ke", "k10"))
    params = [k + "0" if k == elim else k for k in knames] + ["v0"]
    fac = (Pow("wt", 70.0, 0.75),) if rng.random() < 0.6 else (Lin("wt", 70.0, 0.004),)
    derived = {elim: Scaled(elim + "0", fac)}
    if rng.random() < 0.5:
        derived["v"] = Scaled("v0", (Pow("wt", 70.0, 1.0),))
        out_vol = "v"
    else:
        out_vol = "v0"
    routes = [bolus("dose", states[0]), infusion("iv", "central")]
    m = analytical(name=f"cz{seed}", params=params, derived=derived, covariates=["wt"], structure=name, states=states,
                   outputs=["cp"], routes=routes, out={"cp": Ratio("central", out_vol)},
                   cov_time="segment_end_abs" if seed % 4 == 3 else "segment_dt")
    stretch = bool(rng.random() < 0.5)
    times = np.sort(rng.uniform(0.3, 30.0, int(rng.integers(3, 9))))
    t_inf, d_inf = float(rng.uniform(0.5, 6)), float(rng.uniform(0.5, 3))
    with_inf = bool(rng.random() < 0.5)
    two_occ = bool(rng.random() < 0.3)
    subs = []
    n_sub = int(rng.integers(9, 40))
    for i in range(n_sub):
        f = float(rng.uniform(0.9, 1.15)) if stretch else 1.0
        b = Subject.builder(f"s{i}").covariate("wt", 0.0, float(rng.uniform(45, 110)))
        if rng.random() < 0.5:
            b = b.covariate("wt", float(rng.uniform(5, 30)), float(rng.uniform(45, 110)))
        b = b.bolus(0.0, float(rng.uniform(50, 300)), "dose")
        if with_inf:
            b = b.infusion(t_inf * f, float(rng.uniform(50, 200)), "iv", d_inf * f)
        for t in times:
            b = b.missing_observation(float(t) * f, "cp")
        if two_occ:
            b = b.reset().covariate("wt", 0.0, float(rng.uniform(45, 110))).bolus(0.0, 80.0, "dose")
            for t in (1.0, 4.0, 9.5):
                b = b.missing_observation(t * f, "cp")
        subs.append(b.build())
    n = int(rng.choice([3, 33, 40, 70, 71, 130]))
    theta = np.concatenate([kernel_theta(name, n, rng), rng.uniform(10, 80, (n, 1))], axis=1)
    recipe = dict(seed=seed, structure=name, derived=list(derived), stretch=stretch, subjects=n_sub, support=n)
    flat = m.flatten(Data(subs))
    pop = runtime.DevicePopulation(flat, 0)
    pred, st = runtime.predict(m, pop, np.ascontiguousarray(theta))
    torch.cuda.synchronize()
    kname = runtime.last_kernel_name()
    if n >= 33 and ns <= 2:
        assert kname.startswith("pmx_analytical_classed<dyn>"), (recipe, kname)
    got, st = pred.cpu().numpy(), st.cpu().numpy()
    want, wst = oracle.predict(m, flat, theta)
    np.testing.assert_array_equal(st, wst, err_msg=str(recipe))
    ok = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), ok, err_msg=str(recipe))
    scale = np.maximum(np.abs(want[ok]), 1e-9 * np.abs(want[ok]).max() + 1e-300)
    assert (np.abs(got[ok] - want[ok]) / scale).max() < 1e-6, (recipe, kname)
    # the fused log-likelihood of the same case
    vals = np.abs(np.where(np.isfinite(want[:, 0]), want[:, 0], 1.0)) * np.exp(rng.normal(0, 0.2, want.shape[0])) + 0.05
    vals[rng.random(vals.shape) < float(rng.choice([0.0, 0.2]))] = np.nan
    flat_l = m.flatten(Data(subs))
    flat_l.ev_value = flat_l.ev_value.copy()
    flat_l.ev_value[flat_l.ev_kind == _abi.PMX_EV_OBSERVATION] = vals
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
    pop_l = runtime.DevicePopulation(flat_l, 0)
    ll, lst = runtime.loglik(m, pop_l, em, np.ascontiguousarray(theta))
    torch.cuda.synchronize()
    wll, wlst = oracle.loglik(m, flat_l, em, theta)
    np.testing.assert_array_equal(lst.cpu().numpy(), wlst, err_msg=str(recipe))
    okl = np.isfinite(wll)
    gl = ll.cpu().numpy()
    np.testing.assert_array_equal(np.isfinite(gl), okl, err_msg=str(recipe))
    assert (np.abs(gl[okl] - wll[okl]) / np.maximum(np.abs(wll[okl]), 1.0)).max() < 1e-6, (recipe, runtime.last_kernel_name())
    # the ODE twin: covariates bound at the stage times, so it is its own model - checked against the RK4 oracle
    mo = ode(name=f"oz{seed}", params=params, derived=derived, covariates=["wt"], diffeq=diffeq, states=states, outputs=["cp"],
             routes=routes, out={"cp": Ratio("central", out_vol)}, h_max=0.05)
    flat_o = mo.flatten(Data(subs[:8]))
    th_o = np.ascontiguousarray(theta[: min(n, 40)])
    pop_o = runtime.DevicePopulation(flat_o, 0)
    po, so = runtime.predict(mo, pop_o, th_o)
    torch.cuda.synchronize()
    wo, wso = oracle.predict(mo, flat_o, th_o)
    np.testing.assert_array_equal(so.cpu().numpy(), wso, err_msg=str(recipe))
    go = po.cpu().numpy()
    oko = np.isfinite(wo)
    np.testing.assert_array_equal(np.isfinite(go), oko, err_msg=str(recipe))
    sc = np.maximum(np.abs(wo[oko]), 1e-9 * np.abs(wo[oko]).max() + 1e-300)
    assert (np.abs(go[oko] - wo[oko]) / sc).max() < 1e-8, (recipe, runtime.last_kernel_name())

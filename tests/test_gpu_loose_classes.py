"""Loose classes (pmx_compile.hpp ClassPlan): subjects that share a program SHAPE (op kinds, inputs, outputs in the
same order) but not its step lengths - recorded sampling times instead of protocol times - are batched G at a time
like the members of an exact class, each with its own propagator per step.  GPU against the CPU oracle, predictions
and fused log-likelihood, alone and mixed with exact classes and with subjects no class takes."""
import numpy as np
import pytest

import oracle
from pharmsol_amd import Analytical, AssayErrorModel, AssayErrorModels, Data, ErrorPoly, Ratio, Subject, _abi, runtime, synth
from pharmsol_amd import bolus as bolus_route
from tests import models
from tests.test_gpu_fuzz import STRUCTS, kernel_theta

pytestmark = pytest.mark.gpu


def check(model, flat, theta, expect_kernel, loglik=False, tol=1e-6, em=None):
    import torch

    pop = runtime.DevicePopulation(flat, 0)
    dirty = torch.full((flat.n_subjects, theta.shape[0]), 177, dtype=torch.uint8, device="cuda")
    if loglik:
        if em is None:
            em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
        got, st = runtime.loglik(model, pop, em, np.ascontiguousarray(theta), status=dirty)
        want, wst = oracle.loglik(model, flat, em, theta)
    else:
        got, st = runtime.predict(model, pop, np.ascontiguousarray(theta), status=dirty)
        want, wst = oracle.predict(model, flat, theta)
    torch.cuda.synchronize()
    assert runtime.last_kernel_name().startswith(expect_kernel), runtime.last_kernel_name()
    got, st = got.cpu().numpy(), st.cpu().numpy()
    np.testing.assert_array_equal(st, wst)
    ok = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), ok)
    scale = np.maximum(np.abs(want[ok]), (1.0 if loglik else 1e-9 * np.abs(want[ok]).max()) + 1e-300)
    assert (np.abs(got[ok] - want[ok]) / scale).max() < tol


def with_observed_values(flat, model, theta, seed):
    """Measured values for the log-likelihood: the oracle's predictions at the first support point, with noise; a fifth missing."""
    want, _ = oracle.predict(model, flat, theta[:1])
    rng = np.random.default_rng(seed)
    vals = np.abs(np.where(np.isfinite(want[:, 0]), want[:, 0], 1.0)) * np.exp(rng.normal(0, 0.2, want.shape[0])) + 0.05
    vals[rng.random(vals.shape) < 0.2] = np.nan
    flat.ev_value = flat.ev_value.copy()
    flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION] = vals
    return flat


@pytest.mark.parametrize("n_support", [64, 96, 301])
@pytest.mark.parametrize("loglik", [False, True])
def test_jittered_c3_population_runs_as_loose_classes(n_support, loglik):
    model = synth.model_two_cpt_iv()
    flat = synth.population_c23(1003, ragged=True)  # 1003: the last chunk is partial
    theta = synth.theta_c3(n_support)
    if loglik:
        flat = with_observed_values(flat, model, theta, 3)
    check(model, flat, theta, "pmx_analytical_classed<ll,loose>" if loglik else "pmx_analytical_classed<loose>", loglik)


def shaped_subject(rng, shape, i):
    """One of three program shapes with this subject's own times: 0 = infusion + 6 samples, 1 = bolus + infusion + 4
    samples over two occasions, 2 = two boluses with a sample at the second dose time."""
    b = Subject.builder(f"s{i}")
    if shape == 0:
        b = b.infusion(0.0, 300.0 + i, 0, 0.5 + 0.5 * rng.random())
        for t in np.sort(rng.uniform(0.6, 30.0, 6)):
            b = b.missing_observation(float(t), 0)
    elif shape == 1:
        b = b.bolus(0.0, 100.0 + i, 0).infusion(float(rng.uniform(1.0, 2.0)), 200.0, 0, 1.0)
        for t in np.sort(rng.uniform(3.5, 20.0, 2)):
            b = b.missing_observation(float(t), 0)
        b = b.reset().bolus(0.0, 50.0, 0)
        for t in np.sort(rng.uniform(0.5, 12.0, 2)):
            b = b.missing_observation(float(t), 0)
    else:
        t2 = float(rng.uniform(6.0, 12.0))
        b = b.bolus(0.0, 80.0, 0).missing_observation(float(rng.uniform(0.5, 5.0)), 0).bolus(t2, 80.0, 0)
        b = b.missing_observation(t2, 0).missing_observation(t2 + float(rng.uniform(1.0, 9.0)), 0)
    return b.build()


@pytest.mark.parametrize("structure", ["one_compartment", "one_compartment_with_absorption", "two_compartments_cl",
                                       "two_compartments_with_absorption", "three_compartments",
                                       "three_compartments_cl_with_absorption"])
@pytest.mark.parametrize("loglik", [False, True])
def test_exact_loose_and_unclassed_subjects_in_one_population(structure, loglik):
    ns, nk, central = STRUCTS[structure]
    rng = np.random.default_rng(21)
    model = Analytical.new(structure, {0: Ratio(central, nk)}, nparams=nk + 1).with_nstates(ns).with_ndrugs(1).with_nout(1)
    subs = []
    proto = shaped_subject(rng, 0, 0)
    for i in range(37):  # a shared design: exact classes
        b = Subject.builder(f"e{i}")
        for ev in proto.occasions[0].events:
            b = b.infusion(ev.time, ev.amount + i, 0, ev.duration) if hasattr(ev, "duration") else b.missing_observation(ev.time, 0)
        subs.append(b.build())
    for i in range(61):  # three shapes with individual times: loose classes
        subs.append(shaped_subject(rng, i % 3, i))
    for i in range(5):  # shapes nobody shares, and an empty subject: the generic walker
        subs.append(models.random_subject(rng, multi_occasion=bool(i % 2)))
    subs.insert(40, Subject.builder("empty").build())
    order = rng.permutation(len(subs))
    flat = model.flatten(Data([subs[k] for k in order]))
    n = 72
    theta = np.concatenate([kernel_theta(structure, n, rng), rng.uniform(10, 80, (n, 1))], axis=1)
    if loglik:
        flat = with_observed_values(flat, model, theta, 5)
    check(model, flat, theta, "pmx_analytical_classed", loglik)


def test_loose_classes_can_be_switched_off(monkeypatch):
    from pharmsol_amd import _ffi

    monkeypatch.setenv("PMX_TUNE_LOOSE", "0")
    _ffi.lib().pmx_debug_reload_env()  # the switches are read once per process; this re-reads them
    try:
        model = synth.model_two_cpt_iv()
        check(model, synth.population_c23(200, ragged=True), synth.theta_c3(64), "pmx_analytical_steps")
    finally:
        monkeypatch.delenv("PMX_TUNE_LOOSE")
        _ffi.lib().pmx_debug_reload_env()


@pytest.mark.parametrize("ragged", [False, True])
def test_bioavailability_models_are_classed(ragged):
    # fa scales the recorded bolus amount per lane (structs.rs:645-666): the plan keeps the recorded amounts
    model = Analytical.new("one_compartment_with_absorption", {0: Ratio(1, 2)}, nparams=4, fa={0: 3})
    model = model.with_nstates(2).with_ndrugs(1).with_nout(1)
    rng = np.random.default_rng(8)
    subs = []
    for i in range(90):
        b = Subject.builder(f"s{i}").bolus(0.0, 100.0 + i, 0).bolus(12.0, 50.0, 0)
        times = [1.0, 2.0, 6.0, 12.0, 13.0, 20.0]
        if ragged:
            times = [t * (1.0 + 0.1 * (rng.random() - 0.5)) if t != 12.0 else t for t in times]
        for t in times:
            b = b.missing_observation(float(t), 0)
        subs.append(b.build())
    n = 80
    theta = np.stack([rng.uniform(1.0, 3.0, n), rng.uniform(0.05, 0.4, n), rng.uniform(10, 60, n), rng.uniform(0.3, 1.0, n)], 1)
    check(model, model.flatten(Data(subs)), theta, "pmx_analytical_classed<loose>" if ragged else "pmx_analytical_classed")


@pytest.mark.parametrize("ragged", [False, True])
def test_pmetrics_indexed_models_are_classed(ragged):
    # pm_ wrappers (analytical/mod.rs:62-90): model state / input 1 is the kernel's 0; subjects that dose the pad slot
    # (input 0) stay with the generic walker, the others are classed
    model = Analytical.new("pm_one_compartment_with_absorption", {0: Ratio(2, 2)}, nparams=3)
    model = model.with_nstates(3).with_ndrugs(2).with_nout(1)
    rng = np.random.default_rng(9)
    subs = []
    for i in range(70):
        b = Subject.builder(f"s{i}").bolus(0.0, 100.0 + i, 1).infusion(6.0, 60.0, 1, 2.0)
        if i % 10 == 0:
            b = b.bolus(3.0, 25.0, 0)  # into the pad slot: dropped by the wrapper, not classed here
        times = [1.0, 2.0, 6.0, 9.0, 13.0, 20.0]
        if ragged:
            times = [t * (1.0 + 0.1 * (rng.random() - 0.5)) for t in times]
        for t in times:
            b = b.missing_observation(float(t), 0)
        subs.append(b.build())
    flat = model.flatten(Data(subs))
    plan = runtime.class_plan(model, flat)
    assert plan["generic_subjects"] == 7 and plan["classed_subjects"] == 63
    assert (plan["chunks_loose"] > 0) == ragged
    n = 80
    theta = np.stack([rng.uniform(1.0, 3.0, n), rng.uniform(0.05, 0.4, n), rng.uniform(10, 60, n)], 1)
    check(model, flat, theta, "pmx_analytical_classed<loose>" if ragged else "pmx_analytical_classed")


@pytest.mark.parametrize("structure", ["one_compartment_with_absorption", "two_compartments_with_absorption",
                                       "three_compartments_cl_with_absorption"])
@pytest.mark.parametrize("loglik", [False, True])
def test_lag_time_models_on_a_shared_design_are_classed(structure, loglik):
    # one lagged input: the bolus times are the class's, so every lane's landing times t + lag(theta) split the PROP
    # steps of all G members alike (lag_prop / lag_open_occasion of the generic walker over the whole batch).  Lags
    # of 0, inside a step, across several observations, across a later dose and beyond the last event; two occasions,
    # the second dosed before its first sample.
    ns, nk, central = STRUCTS[structure]
    rng = np.random.default_rng(31)
    model = Analytical.new(structure, {0: Ratio(central, nk)}, nparams=nk + 3, lag={0: nk + 1}, fa={0: nk + 2})
    model = model.with_nstates(ns).with_ndrugs(1).with_nout(1)
    subs = []
    for i in range(45):
        b = Subject.builder(f"s{i}").bolus(0.0, 100.0 + i, 0).bolus(12.0, 40.0 + i, 0).infusion(2.0, 30.0, 0, 1.5)
        for t in (0.5, 1.0, 2.0, 4.0, 12.0, 12.5, 20.0):
            b = b.missing_observation(t, 0)
        b = b.reset().bolus(0.0, 75.0, 0).bolus(0.0, 5.0 + i, 0)
        for t in (3.0, 6.0):
            b = b.missing_observation(t, 0)
        subs.append(b.build())
    subs.append(models.random_subject(rng))  # a design of its own: generic walker
    flat = model.flatten(Data(subs))
    n = 64
    lag = rng.choice([0.0, 0.25, 0.75, 1.5, 3.0, 13.0, 40.0], size=n)
    theta = np.concatenate([kernel_theta(structure, n, rng), rng.uniform(10, 80, (n, 1)), lag[:, None],
                            rng.uniform(0.4, 1.0, (n, 1))], axis=1)
    if loglik:
        flat = with_observed_values(flat, model, theta, 7)
    check(model, flat, theta, "pmx_analytical_classed<ll,lag>" if loglik else "pmx_analytical_classed<lag>", loglik)
    plan = runtime.class_plan(model, flat)
    assert plan["classed_subjects"] == 45 and plan["chunks_loose"] == 0 and plan["generic_subjects"] == 1
    if not loglik:
        import torch

        # a negative lag is a shift to an earlier time, like the reference (structs.rs:629-634); only NaN is flagged
        neg = theta.copy()
        neg[5, nk + 1], neg[9, nk + 1] = -1.0, -30.0
        check(model, flat, neg, "pmx_analytical_classed<lag>")
        bad = theta.copy()
        bad[6, nk + 1] = np.nan
        pred, st = runtime.predict(model, runtime.DevicePopulation(flat, 0), np.ascontiguousarray(bad))
        torch.cuda.synchronize()
        pred, st = pred.cpu().numpy(), st.cpu().numpy()
        assert (st[:, 6] == _abi.PMX_PAIR_BAD_LAG).all() and np.isnan(pred[:, 6]).all()
        assert (np.delete(st, 6, axis=1) == 0).all() and np.isfinite(np.delete(pred, 6, axis=1)).all()


def test_censored_observations_in_loose_classes():
    # BLOQ / ALOQ rows and per-observation error polynomials on a population with individual sampling times: the
    # loose chunks mark the censored rows and fold them from their full records
    from tests.test_gpu_likelihood import EM_ADD, _censor_some, assert_ll_parity
    from tests.test_gpu_likelihood import with_observed_values as observed

    rng = np.random.default_rng(12)
    model = synth.model_two_cpt_iv()
    flat = synth.population_c23(203, ragged=True)
    theta = synth.theta_c3(72)
    flat = _censor_some(observed(model, flat, theta[:1], rng), rng)
    assert_ll_parity(model, flat, EM_ADD, theta, expect_kernel="pmx_analytical_classed<ll,loose>")


# --------------------------------------------------------------------------- more than one output equation
def _two_output_subjects(rng, kind, n):
    """Observations alternate between output 0 (central / v) and output 1 (peripheral / v2)."""
    subs = []
    if kind == "exact":
        for i in range(n):
            b = Subject.builder(f"x{i}").infusion(0.0, 300.0 + i, 0, 0.5).bolus(6.0, 40.0 + i, 0)
            for k, t in enumerate((0.5, 1.0, 2.0, 6.0, 6.0, 8.0, 12.0)):
                b = b.missing_observation(t, k % 2)
            subs.append(b.build())
    else:
        for i in range(n):
            b = Subject.builder(f"l{i}").infusion(0.0, 300.0 + i, 0, 0.5 + 0.5 * rng.random())
            for k, t in enumerate(np.sort(rng.uniform(0.6, 30.0, 6))):
                b = b.missing_observation(float(t), k % 2)
            subs.append(b.build())
    return subs


@pytest.mark.parametrize("kind", ["exact", "loose"])
@pytest.mark.parametrize("loglik", [False, True])
def test_two_output_equations_in_classed_kernels(kind, loglik):
    """nout = 2: the classed kernels re-derive the volume of outputs beyond the first (the fused 2-bit outeq of a
    program step); volumes are a primary parameter and a derived value WITHOUT covariate factors (still classed:
    its base parameter is the volume, like lane_setup / the oracle's model_out)."""
    from pharmsol_amd import Scaled, analytical, infusion as inf_route

    rng = np.random.default_rng(31)
    m = analytical(name="two_out", params=["ke", "kcp", "kpc", "v", "v2raw"], derived={"v2": Scaled("v2raw", ())},
                   structure="two_compartments", states=["central", "peripheral"], outputs=["0", "1"],
                   routes=[inf_route("0", "central"), bolus_route("0", "central")],
                   out={"0": Ratio("central", "v"), "1": Ratio("peripheral", "v2")})
    subs = _two_output_subjects(rng, kind, 43)
    flat = m.flatten(Data(subs))
    theta = np.concatenate([synth.theta_c3(72), rng.uniform(5, 40, (72, 1))], axis=1)
    if loglik:
        flat = with_observed_values(flat, m, theta, 9)
        em = (AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
              .add(1, AssayErrorModel.proportional(ErrorPoly(0.1, 0.05, 0.0, 0.0), 0.5)))
        check(m, flat, theta, "pmx_analytical_classed", loglik, em=em)
    else:
        check(m, flat, theta, "pmx_analytical_classed", loglik)


def test_two_output_equations_in_a_lag_class_with_a_bad_lane():
    """LAGC variant + nout = 2: a NaN-lag lane must come back NaN on BOTH outputs (the secondary volume used to be
    recomputed without the lane's poison)."""
    import torch

    m = Analytical.new("one_compartment_with_absorption", {0: Ratio(1, 2), 1: Ratio(0, 3)}, nparams=5, lag={0: 4})
    m = m.with_nstates(2).with_ndrugs(1).with_nout(2)
    subs = []
    for i in range(40):
        b = Subject.builder(f"g{i}").bolus(0.0, 100.0 + i, 0).bolus(12.0, 50.0, 0)
        for k, t in enumerate((0.5, 1.0, 2.0, 4.0, 12.0, 13.0, 16.0)):
            b = b.missing_observation(t, k % 2)
        subs.append(b.build())
    flat = m.flatten(Data(subs))
    rng = np.random.default_rng(33)
    n = 64
    th = np.stack([rng.uniform(1.0, 2.0, n), rng.uniform(0.05, 0.3, n), rng.uniform(10, 50, n), rng.uniform(1, 3, n),
                   np.round(rng.uniform(-1, 3, n) * 2) / 2], axis=1)
    check(m, flat, th, "pmx_analytical_classed<lag>")
    th[11, 4] = np.nan
    pop = runtime.DevicePopulation(flat, 0)
    got, st = runtime.predict(m, pop, th)
    torch.cuda.synchronize()
    got, st = got.cpu().numpy(), st.cpu().numpy()
    assert np.isnan(got[:, 11]).all() and (st[:, 11] == _abi.PMX_PAIR_BAD_LAG).all()
    assert np.isfinite(np.delete(got, 11, axis=1)).all() and (np.delete(st, 11, axis=1) == 0).all()

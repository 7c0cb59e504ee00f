"""GPU parity tests (run with `-m gpu` on the MI355X box): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs, against the committed golden fixtures, and — at
BASELINE.json's full sizes — through size-independent properties.

Tolerances (BASELINE.json north_star): 1e-6 relative for the analytical back-end, 1e-4 for the ODE
back-end.  Relative error is |gpu - cpu| / max(|cpu|, floor) with floor = 1e-12 x the column scale,
so pre-dose zeros compare absolutely (SURVEY.md §8d "Error metric").
"""
import json
import math
import os

import numpy as np
import pytest

import oracle
from pharmsol_amd import (Analytical, Data, Parameters, Pow, Ratio, Scaled, Subject, _abi, analytical, bolus,
                          infusion, runtime, synth)
from tests import models
from tests.test_oracle_independent_math import CENTRAL, GOLD, build_subject

pytestmark = pytest.mark.gpu

TOL_ANALYTICAL = 1e-6
TOL_ODE = 1e-4


def rel_err(got, want):
    scale = max(float(np.nanmax(np.abs(want))) if want.size else 0.0, 1e-300)
    return np.abs(got - want) / np.maximum(np.abs(want), 1e-12 * scale)


def gpu_predict(model, flat, theta, batch=False):
    """Through the device-pointer C ABI (pmx_predict_device / pmx_predict_batch_device)."""
    import torch

    pop = runtime.DevicePopulation(flat, 0)
    pred, status = runtime.predict(model, pop, np.ascontiguousarray(theta, dtype=np.float64), batch=batch)
    torch.cuda.synchronize()
    return pred.cpu().numpy(), status.cpu().numpy()


def assert_parity(model, flat, theta, tol, batch=False, expect_kernel=None):
    got, st = gpu_predict(model, flat, theta, batch=batch)
    if expect_kernel:
        assert runtime.last_kernel_name().startswith(expect_kernel), runtime.last_kernel_name()
    want, wst = (oracle.predict_batch if batch else oracle.predict)(model, flat, theta)
    assert got.shape == want.shape
    np.testing.assert_array_equal(st, wst)
    ok = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), ok)
    if ok.any():
        err = rel_err(got[ok], want[ok]).max()
        assert err <= tol, f"max rel err {err:.3e} > {tol:g}"
    return got, want


# --------------------------------------------------------------------------- C1..C5 (reduced sizes vs oracle)
def test_c1_analytical_readme_through_the_equation_api():
    # examples/analytical_readme.rs, read like the reference example
    m = models.readme_analytical()
    params = Parameters.with_model(m, [("ka", 1.2), ("ke0", 0.08), ("v", 194.0)])
    predictions = m.estimate_predictions(models.readme_subject(), params)
    assert predictions.flat_times() == [0.5, 1.0, 2.0, 4.0]
    want = [1.1363216631314599, 1.7130756583758835, 2.0906323551896495, 1.956103669112038]
    np.testing.assert_allclose(predictions.flat_predictions(), want, rtol=TOL_ANALYTICAL)


def test_c2_two_compartment_10k_subjects_one_support_point():
    m, flat, theta = synth.config_c2(10_000)
    assert_parity(m, flat, theta, TOL_ANALYTICAL, expect_kernel="pmx_analytical_pair")


def test_c3_grid_slice():
    m, flat, theta = synth.config_c3(500, 1000)
    assert_parity(m, flat, theta, TOL_ANALYTICAL, expect_kernel="pmx_analytical_classed")


def test_c4_ode_rk4_divergent_schedules_batch():
    m, flat, theta = synth.config_c4(5_000)
    got, want = assert_parity(m, flat, theta, TOL_ODE, batch=True, expect_kernel="pmx_ode_rk4_pair")
    # and against the EXACT solution (closed form one_compartment) at the ODE tolerance
    ma = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=2).with_nstates(1).with_ndrugs(1).with_nout(1)
    exact, _ = oracle.predict_batch(ma, flat, theta)
    assert rel_err(got, exact).max() <= TOL_ODE


def test_c4_at_its_full_size_every_prediction():
    # BASELINE configs[3] whole: 50k irregular subjects, one lane each (782 waves: the latency-bound shape the lane
    # state machine, its LDS op ring and the steps-per-trip bound were tuned on) against the RK4 oracle
    m, flat, theta = synth.config_c4(50_000)
    assert_parity(m, flat, theta, 1e-9, batch=True, expect_kernel="pmx_ode_rk4_pair")


@pytest.mark.parametrize("cov_time", ["segment_dt", "segment_end_abs"])
def test_c5_three_compartment_absorption_time_varying_wt(cov_time):
    m, flat, theta = synth.config_c5(300, 512, cov_time)
    assert_parity(m, flat, theta, TOL_ANALYTICAL, expect_kernel="pmx_analytical_dyn3")


def test_c5_subject_constant_covariate():
    # one wt value per subject (the usual allometric-scaling case) and a mix of constant and interpolated subjects
    m = synth.model_three_cpt_abs_wt()
    flat = synth.population_c5(300, constant_wt=True)
    assert_parity(m, flat, synth.theta_c5(512), TOL_ANALYTICAL, expect_kernel="pmx_analytical_dyn3")


@pytest.mark.parametrize("structure,params,states,central", [
    ("one_compartment", ["ke0", "v0"], ["central"], "central"),
    ("one_compartment_with_absorption", ["ka", "ke0", "v0"], ["gut", "central"], "central"),
    ("two_compartments", ["ke0", "kcp", "kpc", "v0"], ["central", "periph"], "central"),
    ("two_compartments_with_absorption", ["ke0", "ka", "kcp", "kpc", "v0"], ["gut", "central", "periph"], "central"),
])
@pytest.mark.parametrize("n_support", [3, 96])
def test_covariate_derived_constants_of_the_smaller_structures(structure, params, states, central, n_support):
    # ke = ke0 (wt/70)^0.75 and v = v0 (wt/70), on subjects with a constant wt, with knots, and with two occasions;
    # GRID and PAIR lane mappings against the oracle
    m = analytical(name="wt_" + structure, params=params,
                   derived={"ke": Scaled("ke0", (Pow("wt", 70.0, 0.75),)), "v": Scaled("v0", (Pow("wt", 70.0, 1.0),))},
                   covariates=["wt"], structure=structure, states=states, outputs=["cp"],
                   routes=[bolus("dose", states[0])], out={"cp": Ratio(central, "v")})
    rng = np.random.default_rng(5)
    subs = []
    for i in range(40):
        b = Subject.builder(f"s{i}").bolus(0.0, 100.0 + i, "dose")
        kind = i % 3
        b = b.covariate("wt", 0.0, 50.0 + i)
        if kind == 1:
            b = b.covariate("wt", 10.0, 90.0 - i)
        for t in np.sort(rng.uniform(0.2, 30.0, 6)):
            b = b.missing_observation(float(t), "cp")
        if kind == 2:
            b = b.reset().covariate("wt", 0.0, 75.0).bolus(0.0, 50.0, "dose")
            for t in (1.0, 3.0, 9.0):
                b = b.missing_observation(t, "cp")
        subs.append(b.build())
    flat = m.flatten(Data(subs))
    n = n_support
    th = {"ke0": rng.uniform(0.05, 0.4, n), "v0": rng.uniform(10, 60, n), "ka": rng.uniform(1.5, 3.0, n),
          "kcp": rng.uniform(0.1, 0.6, n), "kpc": rng.uniform(0.05, 0.3, n)}
    theta = np.stack([th[p] for p in params], axis=1)
    assert_parity(m, flat, theta, TOL_ANALYTICAL,
                  expect_kernel=("pmx_analytical_classed<dyn>" if structure != "two_compartments_with_absorption" else "pmx_analytical_grid<dyn>")
                  if n_support >= 48 else "pmx_analytical_pair<dyn>")


# --------------------------------------------------------------------------- every kernel, both lane mappings
@pytest.mark.parametrize("structure,central,theta,subject_fn,diffeq", models.KERNEL_CASES)
@pytest.mark.parametrize("n_support", [1, 70])
def test_every_analytical_kernel_on_reference_fixtures(structure, central, theta, subject_fn, diffeq, n_support):
    m = models.handwritten_analytical(structure, central, len(theta))
    rng = np.random.default_rng(11)
    th = np.array(theta)[None, :] * np.exp(rng.uniform(-0.3, 0.3, size=(n_support, len(theta))))
    if "absorption" in structure:  # keep ka away from the elimination eigenvalues (singular closed form)
        th[:, 1 if structure.startswith("two_compartments_with") else 0] = rng.uniform(9.0, 15.0, size=n_support)
    th[0] = theta
    assert_parity(m, m.flatten(subject_fn()), th, TOL_ANALYTICAL)


@pytest.mark.parametrize("structure,central,theta,subject_fn,diffeq", [c for c in models.KERNEL_CASES if c[4]])
@pytest.mark.parametrize("n_support", [1, 70])
def test_every_ode_model_matches_rk4_oracle_and_closed_form(structure, central, theta, subject_fn, diffeq, n_support):
    mo = models.handwritten_ode(diffeq, central, len(theta), h_max=0.01)
    ma = models.handwritten_analytical(structure, central, len(theta))
    rng = np.random.default_rng(12)
    th = np.array(theta)[None, :] * np.exp(rng.uniform(-0.2, 0.2, size=(n_support, len(theta))))
    if "absorption" in structure:
        th[:, 1 if structure.startswith("two_compartments_with") else 0] *= 2.5
    subj = subject_fn()
    got, _ = assert_parity(mo, mo.flatten(subj), th, TOL_ODE)
    exact, _ = oracle.predict(ma, ma.flatten(subj), th)
    scale = np.abs(exact).max()
    assert np.abs(got - exact).max() / scale <= TOL_ODE


def test_nonlinear_ode_michaelis_menten():
    from pharmsol_amd import ODE

    m = ODE.new("one_cmt_mm", {0: Ratio(0, 2)}, nparams=3, h_max=0.01).with_nstates(1).with_ndrugs(1).with_nout(1)
    rng = np.random.default_rng(5)
    subjects = [models.random_subject(rng) for _ in range(40)]
    flat = m.flatten(Data(subjects))
    th = np.stack([rng.uniform(5, 50, 33), rng.uniform(0.5, 5, 33), rng.uniform(5, 50, 33)], axis=1)
    assert_parity(m, flat, th, TOL_ODE, expect_kernel="pmx_ode_rk4_grid")


# --------------------------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("n_support", [1, 64])
def test_golden_independent_math_fixtures(n_support):
    """tests/golden/independent_math.json (mpmath, 40 digits) straight against the GPU."""
    worst = 0.0
    for case in GOLD:
        st = case["structure"]
        m = models.handwritten_analytical(st, CENTRAL[st], len(case["theta"]))
        flat = m.flatten(build_subject(case["events"]))
        th = np.tile(np.array(case["theta"]), (n_support, 1))
        got, status = gpu_predict(m, flat, th)
        want = np.array(case["expected"])
        assert (status == 0).all()
        scale = max(np.abs(want).max(), 1e-300)
        worst = max(worst, np.abs(got - want[:, None]).max() / scale)
    assert worst <= TOL_ANALYTICAL, worst


# --------------------------------------------------------------------------- edge cases
def test_ragged_population_with_empty_and_doseless_subjects():
    rng = np.random.default_rng(3)
    subjects = [models.random_subject(rng, n_bolus_inputs=2, multi_occasion=True) for _ in range(300)]
    subjects.insert(17, Subject.builder("empty").build())
    subjects.insert(40, Subject.builder("only_dose").bolus(1.0, 5.0, 0).build())
    subjects.append(Subject.builder("only_obs").missing_observation(0.0, 0).missing_observation(3.0, 0).build())
    subjects.append(Subject.builder("empty_last").build())
    m = models.handwritten_analytical("two_compartments", 0, 4)
    flat = m.flatten(Data(subjects))
    th = synth.theta_c3(97)
    assert_parity(m, flat, th, TOL_ANALYTICAL, expect_kernel="pmx_analytical_steps")
    assert_parity(m, flat, th[:3], TOL_ANALYTICAL, expect_kernel="pmx_analytical_pair")
    th_b = synth.theta_c3(len(subjects))
    assert_parity(m, flat, th_b, TOL_ANALYTICAL, batch=True, expect_kernel="pmx_analytical_pair")


def test_matrix_free_covariate_walker_with_empty_subjects_and_occasions():
    """pmx_analytical_dyn3 requests every op and every subject header one ahead, across subjects: subjects without events
    in the middle and at the very end of the stream, several occasions, a subject-constant covariate beside interpolated ones
    (the EIGR instantiation: kept eigenvalues), equal segment lengths (kept segments)."""
    from pharmsol_amd import Pow, Scaled, analytical, bolus

    rng = np.random.default_rng(808)
    m = analytical(name="three_cmt_oral_wt", params=["ka", "k10_0", "k12", "k13", "k21", "k31", "v"],
                   derived={"k10": Scaled("k10_0", (Pow("wt", 70.0, 0.75),))}, covariates=["wt"],
                   structure="three_compartments_with_absorption", states=["gut", "central", "periph1", "periph2"],
                   outputs=["cp"], routes=[bolus("oral", "gut")], out={"cp": Ratio("central", "v")})
    subs = []
    for i in range(60):
        b = Subject.builder(f"s{i}").covariate("wt", 0.0, float(rng.uniform(50, 110)))
        if i % 3:
            b = b.covariate("wt", float(rng.uniform(6, 40)), float(rng.uniform(50, 110)))
        b = b.bolus(0.0, float(rng.uniform(100, 500)), "oral")
        for t in (1.0, 2.0, 4.0, 6.0, 8.0, 12.0, 24.0):
            b = b.missing_observation(t, "cp")
        if i % 4 == 0:
            b = b.reset().covariate("wt", 0.0, float(rng.uniform(50, 110))).bolus(0.0, 100.0, "oral")
            for t in (2.0, 4.0, 6.0, 30.0):
                b = b.missing_observation(t, "cp")
        subs.append(b.build())
    # (a covariate model wants its covariate in every occasion, also in one without events: src/lib.rs:433-443)
    subs.insert(5, Subject.builder("empty_mid").covariate("wt", 0.0, 70.0).build())
    subs.append(Subject.builder("empty_last_but_one").covariate("wt", 0.0, 70.0).build())
    subs.append(Subject.builder("empty_last").covariate("wt", 0.0, 70.0).build())
    flat = m.flatten(Data(subs))
    th = synth.theta_c5(70)
    assert_parity(m, flat, th, TOL_ANALYTICAL, expect_kernel="pmx_analytical_dyn3")
    assert_parity(m, flat, th[:5], TOL_ANALYTICAL, expect_kernel="pmx_analytical_pair")


def test_support_point_counts_around_the_tile_edges():
    m, flat, _ = synth.config_c3(37, 8)
    for P in (31, 32, 33, 255, 256, 257, 1000):
        assert_parity(m, flat, synth.theta_c3(P), TOL_ANALYTICAL)


def test_padded_leading_dimension_is_respected():
    import torch

    m, flat, theta = synth.config_c3(50, 100)
    pop = runtime.DevicePopulation(flat, 0)
    buf = torch.full((pop.n_observations, 128), -7.0, dtype=torch.float64, device="cuda")
    pred, _ = runtime.predict(m, pop, theta, pred=buf[:, :100])
    torch.cuda.synchronize()
    want, _ = oracle.predict(m, flat, theta)
    host = buf.cpu().numpy()
    assert rel_err(host[:, :100], want).max() <= TOL_ANALYTICAL
    assert (host[:, 100:] == -7.0).all()  # padding untouched


def test_complex_roots_are_flagged_per_pair():
    m = models.handwritten_analytical("two_compartments", 0, 4).with_ndrugs(1)
    s = Subject.builder("cx").bolus(0.0, 10.0, 0).missing_observation(0.0, 0).missing_observation(1.0, 0).build()
    th = np.tile(np.array([[0.1, 0.3, 0.2, 50.0]]), (64, 1))
    th[5] = [1.0, -1.9, 1.0, 1.0]
    got, st = gpu_predict(m, m.flatten(Data([s, s])), th)
    want, wst = oracle.predict(m, m.flatten(Data([s, s])), th)
    np.testing.assert_array_equal(st, wst)
    assert st[0, 5] == _abi.PMX_PAIR_COMPLEX_ROOTS and st[:, :5].sum() == 0
    assert np.isnan(got[1, 5]) and np.isfinite(np.delete(got, 5, axis=1)).all()
    # host-pointer form reports the failed pair like log_likelihood_matrix aborting (matrix.rs:83,104)
    with pytest.raises(_abi.PmxError) as e:
        runtime.predict_host(m, m.flatten(s), th, raise_on_pair_failure=True)
    assert e.value.status == _abi.PMX_ERR_PAIR_FAILED


def test_input_and_outeq_out_of_range_errors():
    m = models.handwritten_analytical("one_compartment", 0, 2).with_ndrugs(1)
    s = Subject.builder("oor").bolus(0.0, 1.0, 1).missing_observation(1.0, 0).build()
    with pytest.raises(_abi.PmxError) as e:
        runtime.predict_host(m, m.flatten(s), np.array([[0.1, 1.0]]))
    assert e.value.status == _abi.PMX_ERR_INPUT_OUT_OF_RANGE
    s = Subject.builder("oor2").bolus(0.0, 1.0, 0).missing_observation(1.0, 3).build()
    with pytest.raises(_abi.PmxError) as e:
        runtime.predict_host(m, m.flatten(s), np.array([[0.1, 1.0]]))
    assert e.value.status == _abi.PMX_ERR_OUTEQ_OUT_OF_RANGE


def test_init_and_multiple_occasions():
    m = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=3, init={0: 2}).with_nstates(1).with_ndrugs(
        1).with_nout(1)
    s = (Subject.builder("occ").missing_observation(0.0, 0).missing_observation(1.0, 0).reset()
         .missing_observation(0.0, 0).bolus(0.0, 10.0, 0).missing_observation(1.0, 0).build())
    th = np.array([[0.5, 2.0, 40.0]] * 40)
    got, want = assert_parity(m, m.flatten(s), th, TOL_ANALYTICAL)
    np.testing.assert_allclose(got[:, 0], [20.0, 40.0 * math.exp(-0.5) / 2.0, 0.0, 10.0 * math.exp(-0.5) / 2.0],
                               rtol=1e-12)


def _lag_subjects(rng, n, two_inputs=False):
    subs = []
    for i in range(n):
        b = Subject.builder(f"lag{i}")
        n_occ = 1 + int(rng.integers(0, 2))
        for occ in range(n_occ):
            if occ:
                b = b.reset()
            for _ in range(int(rng.integers(1, 4))):
                b = b.bolus(float(np.round(rng.uniform(0, 24), 1)), float(rng.uniform(50, 300)),
                            int(rng.integers(0, 2)) if two_inputs else 0)
            if rng.random() < 0.5:
                b = b.infusion(float(np.round(rng.uniform(0, 12), 1)), float(rng.uniform(50, 300)), 1 if two_inputs else 0,
                               float(np.round(rng.uniform(0.5, 4), 1)))
            for _ in range(int(rng.integers(2, 9))):
                # times on a 0.5 grid: a lagged bolus regularly lands EXACTLY on an observation / dose time
                b = b.missing_observation(float(np.round(rng.uniform(0, 36) * 2) / 2), 0)
        subs.append(b.build())
    subs.append(Subject.builder("early").bolus(0.0, 100.0, 0).missing_observation(2.0, 0).missing_observation(4.0, 0).build())
    subs.append(Subject.builder("late").missing_observation(1.0, 0).bolus(3.0, 100.0, 0).missing_observation(3.2, 0).build())
    return subs


@pytest.mark.parametrize("n_support,batch", [(70, False), (3, False), (0, True)])
def test_lag_time_and_bioavailability(n_support, batch):
    """structs.rs:611-666 on the device: the bolus is re-timed per support point (lag), then scaled (fa).  Lags on a
    0.5 grid make boluses land exactly on observation times (observation first, event.rs:292-304), cross other
    doses and infusion boundaries, fall before the first event of an occasion and after its last one."""
    rng = np.random.default_rng(77)
    m = Analytical.new("one_compartment_with_absorption", {0: Ratio(1, 2)}, nparams=5, lag={0: 3}, fa={0: 4}).with_nstates(
        2).with_ndrugs(1).with_nout(1)
    subs = _lag_subjects(rng, 60)
    flat = m.flatten(Data(subs))
    n = len(subs) if batch else n_support
    th = np.stack([rng.uniform(1.0, 2.0, n), rng.uniform(0.05, 0.3, n), rng.uniform(10, 50, n),
                   np.round(rng.uniform(0, 3, n) * 2) / 2, rng.uniform(0.3, 1.0, n)], axis=1)
    th[0, 3] = 0.0  # zero lag: the bolus keeps its place (structs.rs:631)
    kernel = "pmx_analytical_pair<lag>" if (batch or n_support < 32) else "pmx_analytical_grid<lag>"
    assert_parity(m, flat, th, TOL_ANALYTICAL, batch=batch, expect_kernel=kernel)


def test_two_lagged_inputs_with_different_lags():
    rng = np.random.default_rng(78)
    m = Analytical.new("two_compartments", {0: Ratio(0, 3)}, nparams=6, lag={0: 4, 1: 5}).with_nstates(2).with_ndrugs(
        2).with_nout(1)
    flat = m.flatten(Data(_lag_subjects(rng, 40, two_inputs=True)))
    th = np.concatenate([synth.theta_c3(64), rng.uniform(0, 2.5, (64, 2))], axis=1)
    assert_parity(m, flat, th, TOL_ANALYTICAL, expect_kernel="pmx_analytical_grid<lag>")


def test_bioavailability_alone_and_on_ode():
    from pharmsol_amd import ODE

    rng = np.random.default_rng(79)
    subs = _lag_subjects(rng, 30)
    m = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=3, fa={0: 2}).with_nstates(1).with_ndrugs(1).with_nout(1)
    th = np.stack([rng.uniform(0.05, 0.5, 64), rng.uniform(5, 50, 64), rng.uniform(0.2, 1.0, 64)], axis=1)
    assert_parity(m, m.flatten(Data(subs)), th, TOL_ANALYTICAL, expect_kernel="pmx_analytical_steps")
    # below the generic walker's GRID/PAIR crossover (48 support points) the same model takes the PAIR kernel
    assert_parity(m, m.flatten(Data(subs)), th[:40], TOL_ANALYTICAL, expect_kernel="pmx_analytical_pair")
    mo = ODE.new("one_cmt_iv", {0: Ratio(0, 1)}, nparams=3, h_max=0.01).with_nstates(1).with_ndrugs(1).with_nout(1)
    mo.fa = {"0": 2}
    assert_parity(mo, mo.flatten(Data(subs)), th, TOL_ODE, expect_kernel="pmx_ode_rk4_grid")


@pytest.mark.parametrize("n_support,batch", [(70, False), (3, False), (0, True)])
def test_ode_lag_time_and_bioavailability(n_support, batch):
    """The ODE back-end's lagged boluses (ode/mod.rs:609-823 over the re-sorted list, structs.rs:611-666): a landing
    bolus splits the RK4 piece it falls in, and both halves re-derive their step count."""
    from pharmsol_amd import ODE

    rng = np.random.default_rng(91)
    m = ODE.new("one_cmt_oral", {0: Ratio(1, 2)}, nparams=5, lag={0: 3}, fa={0: 4}, h_max=0.02).with_nstates(2).with_ndrugs(
        1).with_nout(1)
    # one_cmt_oral: infusions go to the central state, boluses to the depot (index = input)
    subs = _lag_subjects(rng, 50)
    flat = m.flatten(Data(subs))
    n = len(subs) if batch else n_support
    th = np.stack([rng.uniform(1.0, 2.0, n), rng.uniform(0.05, 0.3, n), rng.uniform(10, 50, n),
                   np.round(rng.uniform(0, 3, n) * 2) / 2, rng.uniform(0.3, 1.0, n)], axis=1)
    th[0, 3] = 0.0
    kernel = "pmx_ode_rk4_pair<lag>" if (batch or n_support < 32) else "pmx_ode_rk4_grid<lag>"
    assert_parity(m, flat, th, TOL_ODE, batch=batch, expect_kernel=kernel)


def test_ode_two_lagged_inputs_and_negative_lag():
    from pharmsol_amd import ODE

    rng = np.random.default_rng(92)
    m = ODE.new("two_cmt_iv", {0: Ratio(0, 3)}, nparams=6, lag={0: 4, 1: 5}, h_max=0.02).with_nstates(2).with_ndrugs(
        2).with_nout(1)
    flat = m.flatten(Data(_lag_subjects(rng, 30, two_inputs=True)))
    th = np.concatenate([synth.theta_c3(64), rng.uniform(0, 2.5, (64, 2))], axis=1)
    assert_parity(m, flat, th, TOL_ODE, expect_kernel="pmx_ode_rk4_grid<lag>")
    # a negative lag moves the bolus earlier (`if l != 0.0 { time += l }`, structs.rs:629-634): device == oracle
    th[5, 4] = -1.0
    th[9, 5] = -2.5
    assert_parity(m, flat, th, TOL_ODE, expect_kernel="pmx_ode_rk4_grid<lag>")
    assert_parity(m, flat, th[:12], TOL_ODE, expect_kernel="pmx_ode_rk4_pair<lag>")
    th[5, 4] = np.nan  # the reference panics in its sort; the device flags the pair
    got, st = gpu_predict(m, flat, th)
    assert (st[:, 5] == _abi.PMX_PAIR_BAD_LAG).all() and np.isnan(got[:, 5]).all()
    assert (np.delete(st, 5, axis=1) == 0).all()


@pytest.mark.parametrize("n_support", [40, 8])
def test_negative_lag_shifts_the_bolus_earlier_like_the_reference(n_support):
    """structs.rs:629-634: `if l != 0.0 { *bolus.mut_time() += l }` has no sign check - a negative lag is a shift to an
    earlier time, possibly before the occasion's first event or before another dose.  Device == oracle on the generic
    walker, on an exact class (shared design: LAGC kernel) and on the PAIR kernel; only NaN is flagged."""
    m = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=3, lag={0: 2}).with_nstates(1).with_ndrugs(1).with_nout(1)
    s = Subject.builder("neg").bolus(1.0, 10.0, 0).missing_observation(0.9, 0).missing_observation(2.0, 0).build()
    th = np.array([[0.1, 5.0, 0.5]] * n_support)
    th[7, 2] = -0.25   # lands at 0.75: before the first observation of the list
    th[3, 2] = -4.0    # lands at -3: before time zero
    got, want = assert_parity(m, m.flatten(s), th, TOL_ANALYTICAL)
    assert got[0, 7] > 0 and got[0, 0] == 0.0  # the observation at 0.9 sees the early bolus, and only that lane
    rng = np.random.default_rng(93)
    subs = _lag_subjects(rng, 40)
    th2 = np.stack([rng.uniform(0.05, 0.3, n_support), rng.uniform(10, 50, n_support),
                    np.round(rng.uniform(-3, 3, n_support) * 2) / 2], axis=1)
    assert_parity(m, m.flatten(Data(subs)), th2, TOL_ANALYTICAL)
    shared = [Subject.builder(f"c{i}").bolus(2.0, 10.0 + i, 0).missing_observation(1.0, 0).bolus(6.0, 5.0, 0)
              .missing_observation(4.0, 0).missing_observation(8.0, 0).build() for i in range(33)]
    assert_parity(m, m.flatten(Data(shared)), th2, TOL_ANALYTICAL, expect_kernel="pmx_analytical_classed<lag>")
    assert_parity(m, m.flatten(Data(shared)), th2[:5], TOL_ANALYTICAL, expect_kernel="pmx_analytical_pair<lag>")
    th[7, 2] = np.nan
    got, st = gpu_predict(m, m.flatten(s), th)
    assert st[0, 7] == _abi.PMX_PAIR_BAD_LAG and np.isnan(got[:, 7]).all()
    assert (np.delete(st, 7, axis=1) == 0).all() and np.isfinite(np.delete(got, 7, axis=1)).all()


def test_lagged_bolus_landing_between_observations_an_ulp_apart():
    """Analytical::solve drops sub-segments shorter than 1e-12 (analytical/mod.rs:327): between two observations one ulp
    apart there is no propagation step at all.  A bolus recorded between them stays between them when its lag is zero
    (structs.rs:629-634: no shift, no re-sort), and a lag can put one there from anywhere.  The device merges lagged
    boluses into PROP steps; where none exists it has to take them at the observation (OBS op bit 31, pmx_compile.cpp).
    Found by the fuzz suite (seed 2235).  GRID, PAIR, the lag classes and the user-closure walker."""
    m = Analytical.new("two_compartments", {0: Ratio(0, 3)}, nparams=6, lag={0: 4}, fa={0: 5}).with_nstates(2).with_ndrugs(
        1).with_nout(1)
    t = 10.6
    t_next = float(np.nextafter(t, 20.0))

    def subject(i, amt):
        return (Subject.builder(f"u{i}").missing_observation(10.3, 0).infusion(10.3, 286.0 + i, 0, 0.3).missing_observation(t, 0)
                .bolus(t, amt, 0).missing_observation(t_next, 0).missing_observation(t_next, 0).missing_observation(14.3, 0)
                .bolus(13.3, 50.0, 0).missing_observation(13.8, 0).missing_observation(float(np.nextafter(13.8, 20.0)), 0).build())

    rng = np.random.default_rng(2235)
    n = 48
    th = np.concatenate([synth.theta_c3(n), np.zeros((n, 1)), rng.uniform(0.4, 1.0, (n, 1))], axis=1)
    th[1::4, 4] = -0.0
    th[2::4, 4] = 0.5    # the second bolus (13.3) lands exactly on the observation at 13.8: behind it, in front of 13.8 + ulp
    th[3::4, 4] = 1.25
    one = m.flatten(subject(0, 309.0))
    got, want = assert_parity(m, one, th, TOL_ANALYTICAL, expect_kernel="pmx_analytical_grid<lag>")
    zero = th[:, 4] == 0.0
    assert (want[2, zero] > want[1, zero] + 1.0).all()   # zero lag: the observation an ulp later sees the bolus
    half = th[:, 4] == 0.5
    assert (want[5, half] > want[4, half] * 1.01).all()  # lag 0.5: 13.3 -> 13.8 exactly, seen an ulp later only
    assert_parity(m, one, th[:6], TOL_ANALYTICAL, expect_kernel="pmx_analytical_pair<lag>")
    shared = m.flatten(Data([subject(i, 300.0 + i) for i in range(19)]))
    assert_parity(m, shared, th, TOL_ANALYTICAL, expect_kernel="pmx_analytical_classed<lag>")


@pytest.mark.parametrize("structure,nparams,central", [("one_compartment", 2, 0), ("two_compartments", 4, 0),
                                                       ("two_compartments_with_absorption", 5, 1),
                                                       ("three_compartments", 6, 0)])
def test_exponential_ladder_designs(structure, nparams, central):
    """Steps whose length is 1x/2x/3x/4x the previous step's reuse its exponentials (exp(-l n dt) = exp(-l dt)^n,
    pmx_structures.hpp ladder_pow).  Long doubling chains, equal steps, a 3x and a 4x rung, a chain long enough to hit
    the restart cap: still ~1e-12 from the oracle, which calls exp() on every segment like the reference."""
    rng = np.random.default_rng(5)
    m = models.handwritten_analytical(structure, central, nparams)
    times = [0.03125 * 2 ** k for k in range(14)]          # dt doubles 12 times: the cap (x1024) restarts the ladder
    times += [times[-1] + 6.0 * (k + 1) for k in range(5)]  # equal steps
    times += [times[-1] + 18.0, times[-1] + 18.0 + 24.0]    # x3, then 24 = 4/3 x 18: unrelated -> fresh exp
    times += [times[-1] + 96.0]                             # x4
    subs = []
    for i in range(24):  # shared design -> classed kernel
        b = Subject.builder(f"d{i}").bolus(0.0, float(rng.uniform(50, 500)), 0).infusion(0.0, float(rng.uniform(50, 500)), 0, 0.03125)
        for t in times:
            b = b.missing_observation(t, 0)
        subs.append(b.build())
    for i in range(5):  # ragged -> generic GRID kernel, same ladder along each subject's own PROP ops
        b = Subject.builder(f"r{i}").bolus(0.0, 100.0, 0)
        for t in times[: 9 + 3 * i]:
            b = b.missing_observation(t, 0)
        subs.append(b.build())
    flat = m.flatten(Data(subs))
    lo = np.array([0.02, 0.01, 0.01, 10, 1, 1])[:nparams]
    hi = np.array([0.5, 0.5, 0.5, 100, 2, 2])[:nparams]
    th = np.exp(rng.uniform(np.log(lo), np.log(hi), (64, nparams)))
    if structure == "three_compartments":
        th = synth.theta_c5(64)[:, 1:7]  # k10,k12,k13,k21,k31,v: draws with real eigenvalues
    if structure == "two_compartments_with_absorption":
        th = np.concatenate([synth.theta_c3(64)[:, :1], rng.uniform(0.8, 3.0, (64, 1)), synth.theta_c3(64)[:, 1:]], axis=1)
    assert_parity(m, flat, th, 2e-11, expect_kernel="pmx_analytical_classed")  # (+ the generic kernel on the ragged five)
    assert_parity(m, m.flatten(Data(subs[24:])), th, 2e-11, expect_kernel="pmx_analytical_steps")


def test_pmetrics_one_based_wrappers():
    # pm_* kernels: state/rateiv slot 0 is a dead pad (analytical/mod.rs:62-90)
    m = Analytical.new("pm_two_compartments", {0: Ratio(1, 3)}, nparams=4).with_nstates(3).with_ndrugs(2).with_nout(1)
    s = (Subject.builder("pm").bolus(0.0, 100.0, 1).infusion(2.0, 50.0, 1, 1.0).missing_observation(1.0, 0)
         .missing_observation(2.5, 0).missing_observation(6.0, 0).build())
    th = synth.theta_c3(40)
    got, _ = assert_parity(m, m.flatten(s), th, TOL_ANALYTICAL)
    m0 = Analytical.new("two_compartments", {0: Ratio(0, 3)}, nparams=4).with_nstates(2).with_ndrugs(1).with_nout(1)
    s0 = (Subject.builder("native").bolus(0.0, 100.0, 0).infusion(2.0, 50.0, 0, 1.0).missing_observation(1.0, 0)
          .missing_observation(2.5, 0).missing_observation(6.0, 0).build())
    want, _ = oracle.predict(m0, m0.flatten(s0), th)
    assert rel_err(got, want).max() <= TOL_ANALYTICAL


def test_macro_projection_with_reordered_params():
    # params declared in a different order than the structure needs -> projection (expand/analytical.rs:214-292)
    m = analytical(name="reordered", params=["v", "kpc", "ke", "kcp"], structure="two_compartments",
                   states=["central", "peripheral"], outputs=["cp"], routes=[infusion("iv", "central")],
                   out={"cp": Ratio("central", "v")})
    assert m.desc().n_bind == 3
    _, flat, th = synth.config_c3(20, 50)
    th2 = th[:, [3, 2, 0, 1]]
    got, _ = assert_parity(m, flat, th2, TOL_ANALYTICAL)
    want, _ = oracle.predict(synth.model_two_cpt_iv(), flat, th)
    assert rel_err(got, want).max() <= TOL_ANALYTICAL


def test_host_pointer_form_matches_device_pointer_form():
    m, flat, theta = synth.config_c3(64, 40)
    a, sa = runtime.predict_host(m, flat, theta)
    b, sb = gpu_predict(m, flat, theta)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(sa, sb)


def test_host_pointer_form_large_pitched_and_page_locked_outputs():
    """pmx_predict / pmx_loglik with outputs larger than the 32 MB pinned bounce buffers (several pieces, the DMA of one
    overlapping the host copy of the previous), rows padded to a leading dimension (pieces split at row ends), pageable
    and page-locked (pmx_host_alloc: one DMA) destinations: all bit-identical to the device-pointer form."""
    import ctypes as C

    from pharmsol_amd import _ffi

    L = _ffi.lib()
    m, flat, theta = synth.config_c3(3000, 504)  # 21 000 rows x 504 x 8 B = 85 MB: three pieces
    want, wst = gpu_predict(m, flat, theta)
    pop = runtime.DevicePopulation(flat, 0)
    dm = runtime._as_model(m)
    n_obs, P = flat.n_observations, theta.shape[0]
    th = np.ascontiguousarray(theta)
    for ld in (P, P + 3):
        out = np.full((n_obs, ld), -7.0)
        st = np.zeros((flat.n_subjects, P), dtype=np.uint8)
        _ffi.check(L.pmx_predict(dm.handle, pop.handle, th.ctypes.data, P, out.ctypes.data, ld, st.ctypes.data))
        np.testing.assert_array_equal(out[:, :P], want)
        np.testing.assert_array_equal(st, wst)
        if ld > P:
            assert (out[:, P:] == -7.0).all()  # the padding columns are the caller's
    pinned = runtime.host_empty((n_obs, P))
    _ffi.check(L.pmx_predict(dm.handle, pop.handle, th.ctypes.data, P, pinned.ctypes.data, P, None))
    np.testing.assert_array_equal(pinned, want)
    del pinned


def test_results_are_deterministic_across_launches():
    m, flat, theta = synth.config_c3(200, 300)
    a, _ = gpu_predict(m, flat, theta)
    b, _ = gpu_predict(m, flat, theta)
    np.testing.assert_array_equal(a, b)


# --------------------------------------------------------------------------- full BASELINE sizes: properties
def test_c3_full_size_properties():
    """100k subjects x 1000 support points (5.6 GB of predictions): too big for the oracle, so check
    (1) a random sample of subjects against the oracle, (2) linearity in the dose (the system is linear:
    pred(s) / amount(s) is the same for every subject with this shared schedule), (3) subjects with equal
    (s mod 1000) have bit-identical rows."""
    import torch

    m, flat, theta = synth.config_c3(100_000, 1000)
    pop = runtime.DevicePopulation(flat, 0)
    pred, status = runtime.predict(m, pop, theta)
    torch.cuda.synchronize()
    assert int(status.max().item()) == 0
    assert bool(torch.isfinite(pred).all().item())
    P = 1000
    pr = pred.view(100_000, 7, P)
    # (3) identical dosing => identical predictions
    assert torch.equal(pr[:1000], pr[1000:2000]) and torch.equal(pr[:1000], pr[99_000:100_000])
    # (2) linearity: pred / amount constant over subjects
    amt = torch.as_tensor(500.0 * (1.0 + 0.001 * (np.arange(100_000) % 1000)), device="cuda")
    unit = pr / amt[:, None, None]
    dev = (unit - unit[0:1]).abs().amax() / unit[0].abs().amax()
    assert float(dev) < 1e-13
    # (1) sample vs oracle
    idx = np.sort(np.random.default_rng(0).choice(100_000, size=64, replace=False))
    sub = np.concatenate([np.arange(s * 8, s * 8 + 8) for s in idx])
    from pharmsol_amd.flatten import FlatPopulation

    small = FlatPopulation(subj_occ_off=np.arange(65), occ_ev_off=np.arange(65) * 8, occ_index=np.zeros(64, np.int32),
                           ev_time=flat.ev_time[sub], ev_value=flat.ev_value[sub], ev_duration=flat.ev_duration[sub],
                           ev_kind=flat.ev_kind[sub], ev_io=flat.ev_io[sub])
    want, _ = oracle.predict(m, small, theta)
    got = pr[torch.as_tensor(idx, device="cuda")].reshape(64 * 7, P).cpu().numpy()
    assert rel_err(got, want).max() <= TOL_ANALYTICAL


def test_c5_full_size_properties():
    """BASELINE configs[4] whole: 200k subjects x 512 support points, three compartments + absorption, time-varying wt
    (8.2 GB of predictions - too big for the oracle).  (1) a random sample of subjects against the oracle; (2) linearity in
    the dose: a second pass with every amount doubled gives exactly twice the predictions of the sample (the system is
    linear and the doubling is exact in binary); (3) every pair finite and OK; (4) the propagator kept across equal
    segments of a subject (pmx_compile.cpp prop cache codes) against a pass that rebuilds every one: same numbers."""
    import torch

    from pharmsol_amd import _ffi
    from pharmsol_amd.flatten import FlatPopulation

    m, flat, theta = synth.config_c5(200_000, 512)
    pop = runtime.DevicePopulation(flat, 0)
    pred, status = runtime.predict(m, pop, theta)
    torch.cuda.synchronize()
    assert runtime.last_kernel_name() == "pmx_analytical_dyn3"
    assert int(status.max().item()) == 0 and bool(torch.isfinite(pred).all().item())
    pr = pred.view(200_000, 10, 512)
    idx = np.sort(np.random.default_rng(1).choice(200_000, size=48, replace=False))

    def subset(fl):
        ev = np.concatenate([np.arange(s * 13, s * 13 + 13) for s in idx])
        k0, k1 = fl.cov_knot_off[idx], fl.cov_knot_off[idx + 1]
        kn = np.concatenate([np.arange(a, b) for a, b in zip(k0, k1)])
        return FlatPopulation(subj_occ_off=np.arange(49), occ_ev_off=np.arange(49) * 13, occ_index=np.zeros(48, np.int32),
                              ev_time=fl.ev_time[ev], ev_value=fl.ev_value[ev], ev_duration=fl.ev_duration[ev], ev_kind=fl.ev_kind[ev],
                              ev_io=fl.ev_io[ev], n_covariates=1, cov_knot_off=np.concatenate([[0], np.cumsum(k1 - k0)]),
                              cov_knot_time=fl.cov_knot_time[kn], cov_knot_value=fl.cov_knot_value[kn])

    want, _ = oracle.predict(m, subset(flat), theta)
    got = pr[torch.as_tensor(idx, device="cuda")].reshape(480, 512).cpu().numpy()
    assert rel_err(got, want).max() <= TOL_ANALYTICAL
    # (4) no propagator reuse: rebuild on every segment
    import os

    os.environ["PMX_TUNE_PROP_SLOTS"] = "0"
    _ffi.lib().pmx_debug_reload_env()
    try:
        pop0 = runtime.DevicePopulation(flat, 0)
        pred0, _ = runtime.predict(m, pop0, theta)
        torch.cuda.synchronize()
        dev = float(((pred0 - pred).abs() / pred.abs().clamp_min(1e-12)).max().item())
        assert dev < 1e-12, dev
        del pred0, pop0
    finally:
        del os.environ["PMX_TUNE_PROP_SLOTS"]
        _ffi.lib().pmx_debug_reload_env()
    # (2) doubled doses
    flat2 = synth.population_c5(200_000)
    flat2.ev_value = np.where(flat2.ev_kind == _abi.PMX_EV_BOLUS, 2.0 * flat2.ev_value, flat2.ev_value)
    pred2, _ = runtime.predict(m, runtime.DevicePopulation(flat2, 0), theta, pred=pred)
    torch.cuda.synchronize()
    got2 = pred2.view(200_000, 10, 512)[torch.as_tensor(idx, device="cuda")].reshape(480, 512).cpu().numpy()
    np.testing.assert_array_equal(got2, 2.0 * got)


def test_c4_full_size_against_closed_form():
    m, flat, theta = synth.config_c4(50_000)
    got, st = gpu_predict(m, flat, theta, batch=True)
    assert (st == 0).all()
    ma = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=2).with_nstates(1).with_ndrugs(1).with_nout(1)
    exact, _ = oracle.predict_batch(ma, flat, theta)
    assert rel_err(got, exact).max() <= TOL_ODE


def test_pmetrics_csv_population_end_to_end():
    """A Pmetrics file (ADDL/II expansion, an EVID=4 occasion, OUT=-99, a covariate column) read by
    pharmsol_amd.pmetrics, resolved through a declared model and predicted on the device."""
    from pharmsol_amd import Pow, Scaled, analytical, bolus, infusion
    from pharmsol_amd.pmetrics import from_pmetrics_csv_bytes

    eq = analytical(name="one_cmt_wt", params=["ke0", "v"], derived={"ke": Scaled("ke0", [Pow("wt", 70.0, 0.75)])},
                    covariates=["wt"], structure="one_compartment", states=["central"], outputs=["1"],
                    routes=[bolus("1", "central"), infusion("1", "central")], out={"1": Ratio("central", "v")})
    rows = ["ID,EVID,TIME,DUR,DOSE,ADDL,II,INPUT,OUT,OUTEQ,WT"]
    rng = np.random.default_rng(3)
    for i in range(40):
        wt = 50 + i
        rows.append(f"s{i:02d},1,0,0,{300 + 5 * i},{3 + i % 3},12,1,.,.,{wt}")
        for t in (1, 6, 13, 30, 47.5):
            rows.append(f"s{i:02d},0,{t},.,.,.,.,.,{-99 if i % 4 == 0 else round(float(rng.uniform(1, 9)), 3)},1,{wt + 0.05 * t}")
        if i % 2:
            rows.append(f"s{i:02d},4,72,1.5,{200 + i},.,.,1,.,.,{wt + 4}")
            rows.append(f"s{i:02d},0,75,.,.,.,.,.,2.5,1,{wt + 4.2}")
    data = from_pmetrics_csv_bytes(("\n".join(rows) + "\n").encode())
    flat = eq.flatten(data)
    assert flat.n_subjects == 40 and flat.n_occasions == 60
    th = np.stack([rng.uniform(0.05, 0.4, 48), rng.uniform(10, 60, 48)], axis=1)
    assert_parity(eq, flat, th, TOL_ANALYTICAL, expect_kernel="pmx_analytical_")


def test_small_support_grids_pick_the_measured_lane_mapping():
    """Shared designs go through the classed GRID kernel from 8 support points (64-thread blocks), ragged ones switch
    from PAIR to GRID at 48 (tools/experiments/pairgrid_sweep.sh); every mapping gives the same numbers."""
    rng = np.random.default_rng(101)
    m, flat, theta = synth.config_c3(120, 64)
    for n, kernel in ((4, "pmx_analytical_pair"), (8, "pmx_analytical_classed"), (33, "pmx_analytical_classed"),
                      (64, "pmx_analytical_classed")):
        assert_parity(m, flat, theta[:n], TOL_ANALYTICAL, expect_kernel=kernel)
    subs = [models.random_subject(rng) for _ in range(40)]
    mr = models.handwritten_analytical("two_compartments", 0, 4).with_ndrugs(1)
    fr = mr.flatten(Data(subs))
    for n, kernel in ((20, "pmx_analytical_pair"), (47, "pmx_analytical_pair"), (48, "pmx_analytical_steps"),
                      (100, "pmx_analytical_steps"), (130, "pmx_analytical_steps")):
        assert_parity(mr, fr, synth.theta_c3(n), TOL_ANALYTICAL, expect_kernel=kernel)


def test_handles_are_shared_by_concurrent_host_threads():
    """`Equation: Sync` (equation/mod.rs:377): the reference calls one model from many rayon threads.  Eight host threads
    share one model handle and one population handle (first use races on the lazily built op stream / class plan /
    log-likelihood tables / hiprtc module) and must all get the single-threaded answer."""
    import threading

    import torch

    from pharmsol_amd import AssayErrorModel, AssayErrorModels, ErrorPoly

    rng = np.random.default_rng(7)
    m, flat, theta = synth.config_c3(96, 64)
    vals = rng.uniform(1, 9, flat.n_observations)
    flat.ev_value = flat.ev_value.copy()
    flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION] = vals
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
    src = '''
PMX_DEVICE void pmx_dynamics(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                             const double* derived, double* dx) {
  dx[0] = -(p[0] + p[1]) * x[0] + p[2] * x[1] + rateiv[0];
  dx[1] = p[1] * x[0] - p[2] * x[1];
}
PMX_DEVICE void pmx_outputs(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                            const double* derived, double* y) { y[0] = x[0] / p[3]; }
'''
    from pharmsol_amd import ODE

    mc = ODE.custom(src, nstates=2, nparams=4, h_max=0.05)
    pop = runtime.DevicePopulation(flat, 0)
    thetas = [synth.theta_c3(64, synth.SplitMix64(100 + i)) for i in range(8)]
    results, errors = [None] * 8, []

    def work(i):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                p, _ = runtime.predict(m, pop, thetas[i])
                ll, _ = runtime.loglik(m, pop, em, thetas[i])
                pc, _ = runtime.predict(mc, pop, thetas[i])
            stream.synchronize()
            results[i] = (p.cpu().numpy(), ll.cpu().numpy(), pc.cpu().numpy())
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(8):
        want, _ = oracle.predict(m, flat, thetas[i])
        wll, _ = oracle.loglik(m, flat, em, thetas[i])
        assert rel_err(results[i][0], want).max() < TOL_ANALYTICAL
        assert (np.abs(results[i][1] - wll) / np.maximum(np.abs(wll), 1.0)).max() < 1e-9
        assert rel_err(results[i][2], want).max() < 1e-4  # the same two-compartment system through RK4 (h <= 0.05)


def test_state_vectors_at_the_observation_times():
    """Prediction::state (a19): the amounts in every model state at each observation, through the same kernels with
    the output equations replaced by y = x[state]; checked against the oracle run with exactly such outputs."""
    import torch

    rng = np.random.default_rng(111)
    cases = [(models.handwritten_analytical("two_compartments_with_absorption", 1, 5).with_ndrugs(1), 3,
              np.concatenate([synth.theta_c3(40)[:, :1], rng.uniform(0.8, 3.0, (40, 1)), synth.theta_c3(40)[:, 1:]], axis=1)),
             (models.handwritten_ode("two_cmt_iv", 0, 4, h_max=0.02).with_ndrugs(1), 2, synth.theta_c3(40))]
    subs = [models.random_subject(rng, multi_occasion=True) for _ in range(30)]
    for m, ns, theta in cases:
        flat = m.flatten(Data(subs))
        pop = runtime.DevicePopulation(flat, 0)
        got = runtime.predict_states(m, pop, theta)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        assert got.shape == (flat.n_observations, ns, 40)
        for st in range(ns):
            d = m.desc()
            for o in range(_abi.PMX_MAX_OUT):
                d.out[o].state, d.out[o].vol_src, d.out[o].vol_index = st, _abi.PMX_SRC_NONE, 0
            want, _ = oracle.predict(d, flat, theta)
            assert rel_err(got[:, st, :], want).max() < (TOL_ODE if "ode" in m.kernel_name or m.eq_kind == _abi.PMX_EQ_ODE else TOL_ANALYTICAL)


def test_status_bytes_cleared_by_the_kernel_itself():
    """When the classed kernel serves every subject it clears its own status bytes (no memset between passes): a dirty
    status buffer must come back clean where pairs are healthy and flagged where they are not, pass after pass."""
    import torch

    m, flat, theta = synth.config_c3(203, 64)  # 203: partial last chunk; 64 % 8 == 0
    th = theta.copy()
    th[5, 3] = 0.0   # v = 0 -> non-finite predictions for support point 5
    th[9, :3] = [1.0, -1.0, 1.0]  # (ke + kcp + kpc)^2 < 4 ke kpc: complex eigenvalues for support point 9
    pop = runtime.DevicePopulation(flat, 0)
    status = torch.full((flat.n_subjects, 64), 77, dtype=torch.uint8, device="cuda")  # dirty on purpose
    for _ in range(2):
        pred, st = runtime.predict(m, pop, th, status=status)
        torch.cuda.synchronize()
        assert runtime.last_kernel_name() == "pmx_analytical_classed"
        got = st.cpu().numpy()
        want, wst = oracle.predict(m, flat, th)
        np.testing.assert_array_equal(got, wst)
        assert (got[:, 5] == _abi.PMX_PAIR_NONFINITE).all() and (got[:, 9] == _abi.PMX_PAIR_COMPLEX_ROOTS).all()
        status.fill_(13)  # dirty again before the second pass
    # a population with an empty subject keeps the memset path (the kernel never visits that subject)
    subs = [Subject.builder("empty").build()] + [
        Subject.builder(f"s{i}").infusion(0.0, 500.0, "iv", 0.5).missing_observation(1.0, "cp").missing_observation(2.0, "cp").build()
        for i in range(16)]
    flat2 = m.flatten(Data(subs))
    pop2 = runtime.DevicePopulation(flat2, 0)
    status2 = torch.full((17, 64), 55, dtype=torch.uint8, device="cuda")
    _, st2 = runtime.predict(m, pop2, theta, status=status2)
    torch.cuda.synchronize()
    assert (st2.cpu().numpy() == 0).all()


@pytest.mark.parametrize("exhaustive", [False, True])
def test_placed_prediction_buffer(exhaustive):
    """pmx_prediction_buffer_create: an arena mapped through the HIP virtual-memory API window by window, the kernel timed
    into each (every window when exhaustive), one window kept and everything else returned.  The buffer behaves like any
    other device buffer."""
    import gc

    import torch

    m, flat, theta = synth.config_c3(300, 64)
    pop = runtime.DevicePopulation(flat, 0)
    runtime.predict(m, pop, theta)  # (first use: op stream upload, workspaces, torch's allocator pools - not what is measured)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    # the matrix is ~1 MB: chunks of 2 MiB, 128 windows
    pred = runtime.place_predictions(m, pop, theta, search_gib=0.25, exhaustive=exhaustive)
    assert pred.shape == (flat.n_observations, 64) and pred.is_cuda and pred._pmx_owner.ms_per_pass > 0
    assert free0 - torch.cuda.mem_get_info()[0] < (64 << 20)  # only the window's chunks stayed allocated
    out, st = runtime.predict(m, pop, theta, pred=pred)
    torch.cuda.synchronize()
    want, _ = oracle.predict(m, flat, theta)
    assert rel_err(out.cpu().numpy(), want).max() < TOL_ANALYTICAL
    del pred, out
    gc.collect()
    assert free0 - torch.cuda.mem_get_info()[0] < (8 << 20)


def test_placed_prediction_buffer_with_a_row_pitch_and_the_write_ceiling():
    """pmx_prediction_buffer_create_pitched + pmx_recommended_ld: rows 128-byte aligned, the padding never written;
    pmx_measure_write_ceiling on the same allocation."""
    import ctypes as C

    import torch

    from pharmsol_amd import _ffi

    m, flat, theta = synth.config_c3(300, 70)
    pop = runtime.DevicePopulation(flat, 0)
    ld = runtime.recommended_ld(70)
    assert ld == 80
    pred = runtime.place_predictions(m, pop, theta, search_gib=0.25, exhaustive=True, ld=ld)
    assert pred.shape == (flat.n_observations, 70) and pred.stride(0) == ld and pred._pmx_owner.ms_per_pass > 0
    base = pred._base
    base.fill_(-7.0)
    out, st = runtime.predict(m, pop, theta, pred=pred)
    torch.cuda.synchronize()
    want, _ = oracle.predict(m, flat, theta)
    assert rel_err(out.cpu().numpy(), want).max() < TOL_ANALYTICAL
    assert bool((base[:, 70:] == -7.0).all())  # the padding columns were not touched
    dense, _ = runtime.predict(m, pop, theta)
    assert torch.equal(dense, out)  # same numbers bit for bit, whatever the pitch
    big = torch.empty(1 << 24, dtype=torch.float64, device="cuda")
    gbs = C.c_double()
    _ffi.check(_ffi.lib().pmx_measure_write_ceiling(big.data_ptr(), big.numel(), 3, torch.cuda.current_stream().cuda_stream,
                                                    C.byref(gbs)))
    assert 500.0 < gbs.value < 8000.0 and bool((big == 0.0).all())


@pytest.mark.parametrize("structure,n_support", [("two_compartments", 70), ("two_compartments", 9),
                                                  ("three_compartments", 70)])
def test_a_failed_first_occasion_keeps_the_pair_failed(structure, n_support):
    """Covariate-derived rate constants are rebuilt per segment, so complex eigenvalues belong to an OCCASION: the next
    occasion's rows are finite again, but the pair stays PMX_PAIR_COMPLEX_ROOTS (the reference panics for the whole
    subject; the oracle keeps its status sticky).  Generic walker (3 states), classed<dyn> (2 states), PAIR."""
    from pharmsol_amd import Lin

    two = structure == "two_compartments"
    params = ["ke", "kcp0", "kpc", "v"] if two else ["k10", "k120", "k13", "k21", "k31", "v"]
    dname, src = ("kcp", "kcp0") if two else ("k12", "k120")
    m = analytical(name="sticky", params=params, derived={dname: Scaled(src, (Lin("wt", 70.0, 0.1),))}, covariates=["wt"],
                   structure=structure, states=["central", "periph"] if two else ["central", "p1", "p2"], outputs=["cp"],
                   routes=[bolus("dose", "central")], out={"cp": Ratio("central", "v")})
    subs = []
    for i in range(24):
        # occasion 0: wt = 40 -> factor 1 + 0.1 (40 - 70) = -2: a negative transfer constant, complex roots for some
        # support points; occasion 1: wt = 70 -> factor 1
        b = Subject.builder(f"s{i}").covariate("wt", 0.0, 40.0 if i % 2 == 0 else 70.0).bolus(0.0, 100.0 + i, "dose")
        for t in (1.0, 2.0 + 0.01 * i, 6.0):
            b = b.missing_observation(t, "cp")
        b = b.reset().covariate("wt", 0.0, 70.0).bolus(0.0, 50.0, "dose").missing_observation(1.0, "cp").missing_observation(3.0, "cp")
        subs.append(b.build())
    flat = m.flatten(Data(subs))
    rng = np.random.default_rng(12)
    if two:
        th = np.stack([rng.uniform(0.5, 1.5, n_support), rng.uniform(0.5, 1.0, n_support), rng.uniform(0.5, 1.5, n_support),
                       rng.uniform(10, 50, n_support)], axis=1)
    else:
        th = np.concatenate([synth.theta_c5(n_support)[:, 1:6], rng.uniform(10, 50, (n_support, 1))], axis=1)
        th[:, 1] = rng.uniform(0.5, 1.0, n_support)
    got, want = assert_parity(m, flat, th, TOL_ANALYTICAL)
    _, wst = oracle.predict(m, flat, th)
    failed = wst == _abi.PMX_PAIR_COMPLEX_ROOTS
    assert failed.any() and not failed.all()
    off = flat.subj_obs_off if hasattr(flat, "subj_obs_off") else None
    # rows of the second occasion of a failed pair are finite (statuses were compared by assert_parity)
    rows_per = 5
    for s_i in range(0, 24, 2):
        for p_i in np.nonzero(failed[s_i])[0][:3]:
            blk = got[s_i * rows_per:(s_i + 1) * rows_per, p_i]
            assert np.isnan(blk[:3]).any() and np.isfinite(blk[3:]).all()

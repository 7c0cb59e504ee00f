"""GPU parity of the fused log-likelihood (pmx_loglik_device) against the CPU oracle's
estimate_log_likelihood_dense restatement.  Tolerance 1e-9 relative on the per-(subject, support point) sums
(the predictions inside agree to ~1e-13; the sums are O(10..1e4))."""
import numpy as np
import pytest

import oracle
from pharmsol_amd import (ODE, Analytical, AssayErrorModel, AssayErrorModels, Data, ErrorPoly, Ratio, Subject, _abi,
                          runtime, synth)
from tests import models

pytestmark = pytest.mark.gpu
TOL_LL = 1e-9


def with_observed_values(model, flat, theta_true, rng, missing_frac=0.15, batch=False):
    """Fill the observation slots with 'measured' values = oracle prediction at theta_true x lognormal noise;
    a fraction stays missing (NaN)."""
    pred, _ = (oracle.predict_batch(model, flat, theta_true) if batch else oracle.predict(model, flat, theta_true))
    pred = pred.reshape(flat.n_observations, -1)[:, 0]
    vals = np.abs(pred) * np.exp(rng.normal(0, 0.2, pred.shape)) + 0.05
    vals[rng.random(pred.shape) < missing_frac] = np.nan
    is_obs = flat.ev_kind == _abi.PMX_EV_OBSERVATION
    # observation rows follow the library's per-occasion sort; these populations are already sorted per occasion
    flat.ev_value = flat.ev_value.copy()
    flat.ev_value[is_obs] = vals
    return flat


def gpu_loglik(model, flat, em, theta):
    import torch

    pop = runtime.DevicePopulation(flat, 0)
    ll, st = runtime.loglik(model, pop, em, np.ascontiguousarray(theta, dtype=np.float64))
    torch.cuda.synchronize()
    return ll.cpu().numpy(), st.cpu().numpy()


def assert_ll_parity(model, flat, em, theta, expect_kernel=None):
    got, st = gpu_loglik(model, flat, em, theta)
    if expect_kernel:
        assert runtime.last_kernel_name().startswith(expect_kernel), runtime.last_kernel_name()
    want, wst = oracle.loglik(model, flat, em, theta)
    np.testing.assert_array_equal(st, wst)
    ok = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), ok)
    err = np.abs(got[ok] - want[ok]) / np.maximum(np.abs(want[ok]), 1.0)
    assert err.max() <= TOL_LL, err.max()
    return got, want


EM_ADD = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
EM_PROP = AssayErrorModels.empty().add(0, AssayErrorModel.proportional(ErrorPoly(0.02, 0.15, 0.001, 0.0), 1.3))


@pytest.mark.parametrize("em", [EM_ADD, EM_PROP])
def test_c3_shared_design_classed_kernel(em):
    rng = np.random.default_rng(1)
    m, flat, theta = synth.config_c3(333, 1000)  # 333: the last chunk of 8 is partial
    flat = with_observed_values(m, flat, theta[:1], rng)
    assert_ll_parity(m, flat, em, theta, expect_kernel="pmx_analytical_classed_ll")


@pytest.mark.parametrize("missing_frac", [0.0, 0.02])
def test_classed_fold_without_per_member_tests(missing_frac):
    """A step whose rows are plain for every live member is folded without a test per member (pmx_ll_prepare_chunks'
    flag word, one bit per observation, 63 bits): no missing values -> every step takes that path; 2 % missing -> steps
    of both kinds in one chunk; 70 observations per subject -> observations 63.. take the tested path again."""
    rng = np.random.default_rng(11)
    times = np.sort(rng.uniform(0.1, 48.0, 70))
    subs = []
    for i in range(61):  # 61: a partial last chunk
        b = Subject.builder(f"s{i}").infusion(0.0, 300.0 + 7.0 * i, 0, 0.5)
        for t in times:
            b = b.missing_observation(float(t), 0)
        subs.append(b.build())
    m = models.handwritten_analytical("two_compartments", 0, 4).with_ndrugs(1)
    flat = m.flatten(Data(subs))
    theta = synth.theta_c3(130)
    flat = with_observed_values(m, flat, theta[:1], rng, missing_frac=missing_frac)
    assert_ll_parity(m, flat, EM_PROP, theta, expect_kernel="pmx_analytical_classed_ll")


def test_ragged_population_generic_and_pair_kernels():
    rng = np.random.default_rng(2)
    subs = [models.random_subject(rng, multi_occasion=True) for _ in range(150)]
    subs.insert(9, Subject.builder("empty").build())
    m = models.handwritten_analytical("two_compartments", 0, 4).with_ndrugs(1)
    flat = m.flatten(Data(subs))
    theta = synth.theta_c3(70)
    flat = with_observed_values(m, flat, theta[:1], rng)
    assert_ll_parity(m, flat, EM_ADD, theta, expect_kernel="pmx_analytical_steps")
    assert_ll_parity(m, flat, EM_PROP, theta[:5], expect_kernel="pmx_analytical_pair")


def test_covariate_model_c5():
    rng = np.random.default_rng(3)
    m, flat, theta = synth.config_c5(120, 64)
    flat = with_observed_values(m, flat, theta[:1], rng)
    assert_ll_parity(m, flat, EM_ADD, theta, expect_kernel="pmx_analytical_dyn3")


def test_ode_model():
    rng = np.random.default_rng(4)
    m = models.handwritten_ode("two_cmt_iv", 0, 4, h_max=0.02).with_ndrugs(1)
    subs = [models.random_subject(rng) for _ in range(60)]
    flat = m.flatten(Data(subs))
    theta = synth.theta_c3(40)
    flat = with_observed_values(m, flat, theta[:1], rng)
    assert_ll_parity(m, flat, EM_PROP, theta, expect_kernel="pmx_ode_rk4_grid")


def test_two_outputs_with_their_own_error_models():
    rng = np.random.default_rng(5)
    m = Analytical.new("two_compartments", {0: Ratio(0, 3), 1: Ratio(1, 4)}, nparams=5).with_nstates(2).with_ndrugs(
        1).with_nout(2)
    subs = []
    for i in range(40):
        b = Subject.builder(str(i)).infusion(0.0, 300.0 + i, 0, 1.0)
        for t in (0.5, 1.0, 2.0, 4.0, 8.0):
            b = b.observation(t, float(rng.uniform(0.5, 9.0)), int(rng.integers(0, 2)))
        subs.append(b.build())
    em = (AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.1, 0.1, 0.0, 0.0), 0.0))
          .add(1, AssayErrorModel.proportional(ErrorPoly(0.2, 0.05, 0.0, 0.0), 2.0)))
    theta = np.concatenate([synth.theta_c3(48), rng.uniform(20, 80, (48, 1))], axis=1)
    assert_ll_parity(m, m.flatten(Data(subs)), em, theta)


def test_missing_error_model_and_non_finite_sums():
    m, flat, theta = synth.config_c3(16, 40)
    flat = with_observed_values(m, flat, theta[:1], np.random.default_rng(6), missing_frac=0.0)
    pop = runtime.DevicePopulation(flat, 0)
    with pytest.raises(_abi.PmxError) as e:
        runtime.loglik(m, pop, AssayErrorModels.empty(), theta)
    assert e.value.status == _abi.PMX_ERR_ERROR_MODEL
    th = theta.copy()
    th[3, 3] = 0.0  # v = 0: predictions inf -> the sum is non-finite for that support point only
    got, st = gpu_loglik(m, flat, EM_ADD, th)
    want, wst = oracle.loglik(m, flat, EM_ADD, th)
    np.testing.assert_array_equal(st, wst)
    assert (st[:, 3] == _abi.PMX_PAIR_NONFINITE).all() and (np.delete(st, 3, axis=1) == 0).all()


def test_matches_predictions_then_host_reduction():
    """The fused path equals 'predict, then sum lognormpdf on the host' (what a caller without the fused entry does)."""
    rng = np.random.default_rng(7)
    m, flat, theta = synth.config_c3(64, 96)
    flat = with_observed_values(m, flat, theta[:1], rng)
    ll, _ = gpu_loglik(m, flat, EM_ADD, theta)
    pred, _ = runtime.predict_host(m, flat, theta)
    y = flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION]
    sig = np.sqrt((0.05 + 0.1 * y) ** 2 + 0.1 ** 2)
    term = -0.5 * 1.8378770664093453 - np.log(sig)[:, None] - (y[:, None] - pred) ** 2 / (2 * sig[:, None] ** 2)
    term[np.isnan(y)] = 0.0
    want = term.reshape(64, 7, 96).sum(axis=1)
    assert np.abs(ll - want).max() / np.abs(want).max() < 1e-12


def test_host_pointer_form_through_the_equation_api():
    rng = np.random.default_rng(8)
    m, flat, theta = synth.config_c3(40, 33)
    flat = with_observed_values(m, flat, theta[:1], rng)
    ll, st = m.log_likelihood_matrix(flat, theta, EM_PROP)
    want, _ = oracle.loglik(m, flat, EM_PROP, theta)
    assert ll.shape == (40, 33) and np.abs(ll - want).max() / np.abs(want).max() < TOL_LL


def _censor_some(flat, rng, frac_bloq=0.15, frac_aloq=0.1, frac_poly=0.2):
    """Mark a share of the valued observations BLOQ / ALOQ and give some their own error polynomial."""
    n = flat.n_events
    is_obs = (flat.ev_kind == _abi.PMX_EV_OBSERVATION) & ~np.isnan(flat.ev_value)
    u = rng.random(n)
    cens = np.zeros(n, dtype=np.int8)
    cens[is_obs & (u < frac_bloq)] = _abi.PMX_CENSOR_BLOQ
    cens[is_obs & (u > 1.0 - frac_aloq)] = _abi.PMX_CENSOR_ALOQ
    poly = np.full((n, 4), np.nan)
    own = is_obs & (rng.random(n) < frac_poly)
    poly[own] = np.stack([rng.uniform(0.05, 0.5, own.sum()), rng.uniform(0.0, 0.2, own.sum()), np.zeros(own.sum()),
                          np.zeros(own.sum())], axis=1)
    flat.ev_censor, flat.ev_errorpoly = cens, poly
    return flat


@pytest.mark.parametrize("n_support", [70, 5])
def test_censored_observations_and_per_observation_error_polynomials(n_support):
    """Censor::BLOQ / ALOQ rows take log CDF / log survival (distributions.rs:52-103), an observation's own ErrorPoly
    replaces the model's (error_model.rs:1051-1054).  A shared design runs the classed kernel; censored rows are
    marked in its chunk blocks and folded from their full records."""
    rng = np.random.default_rng(11)
    m, flat, theta = synth.config_c3(200, max(n_support, 8))
    theta = theta[:n_support]
    flat = _censor_some(with_observed_values(m, flat, theta[:1], rng), rng)
    assert_ll_parity(m, flat, EM_ADD, theta,
                     expect_kernel="pmx_analytical_classed_ll" if n_support >= 32 else "pmx_analytical_pair")
    # error polynomials alone keep the classed kernel
    flat.ev_censor = None
    assert_ll_parity(m, flat, EM_PROP, theta,
                     expect_kernel="pmx_analytical_classed_ll" if n_support >= 32 else "pmx_analytical_pair")


def test_censored_tail_far_from_the_prediction():
    """|z| > 37: the reference's asymptote; an upper tail that underflows before that is its Err -> flagged pair."""
    from pharmsol_amd import Censor

    m = models.handwritten_analytical("one_compartment", 0, 2).with_ndrugs(1)
    subs = [Subject.builder("lo").bolus(0.0, 100.0, 0).censored_observation(1.0, 0.01, 0, Censor.BLOQ).build(),
            Subject.builder("hi").bolus(0.0, 100.0, 0).censored_observation(1.0, 50.0, 0, Censor.ALOQ).build(),
            Subject.builder("err").bolus(0.0, 100.0, 0).censored_observation(1.0, 9.5, 0, Censor.ALOQ).build()]
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.1, 0.0, 0.0, 0.0), 0.0))
    th = np.array([[0.2, 10.0]] * 40)  # prediction 8.19, sigma 0.1: z = -82, +418, +13
    flat = m.flatten(Data(subs))
    got, st = gpu_loglik(m, flat, em, th)
    want, wst = oracle.loglik(m, flat, em, th)
    np.testing.assert_array_equal(st, wst)
    assert (st[2] == _abi.PMX_PAIR_NONFINITE).all() and (st[:2] == 0).all()
    np.testing.assert_allclose(got[:2], want[:2], rtol=1e-12)
    assert np.isnan(got[2]).all()


def test_error_models_changing_between_calls_and_invalid_sigma():
    """The sigma tables are rebuilt on the device whenever the error models change (an optimiser moving gamma / lambda
    every call): more distinct models than cache slots, revisits, and a model whose sigma goes negative."""
    rng = np.random.default_rng(21)
    m, flat, theta = synth.config_c3(100, 64)
    flat = with_observed_values(m, flat, theta[:1], rng)
    import torch

    pop = runtime.DevicePopulation(flat, 0)
    ems = [AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), lam))
           for lam in (0.05, 0.1, 0.2, 0.4, 0.8, 1.6)]
    ems += [AssayErrorModels.empty().add(0, AssayErrorModel.proportional(ErrorPoly(0.02, 0.15, 0.0, 0.0), g)) for g in (0.7, 1.3)]
    wants = [oracle.loglik(m, flat, em, theta)[0] for em in ems]
    for order in (range(len(ems)), reversed(range(len(ems))), [0, 5, 0, 7, 2, 2, 6, 1]):
        for i in order:
            ll, st = runtime.loglik(m, pop, ems[i], theta)
            torch.cuda.synchronize()
            got = ll.cpu().numpy()
            assert (np.abs(got - wants[i]) / np.maximum(np.abs(wants[i]), 1.0)).max() < TOL_LL, i
    # gamma < 0 makes sigma = gamma * alpha negative: ErrorModelError::NegativeSigma
    bad = AssayErrorModels.empty().add(0, AssayErrorModel.proportional(ErrorPoly(0.02, 0.15, 0.0, 0.0), -1.0))
    with pytest.raises(_abi.PmxError) as e:
        runtime.loglik_host(m, flat, bad, theta)
    assert e.value.status == _abi.PMX_ERR_ERROR_MODEL and "NegativeSigma" in str(e.value)
    ll, st = runtime.loglik(m, pop, bad, theta)  # device form: flagged rows instead of a call-level error
    torch.cuda.synchronize()
    assert np.isnan(ll.cpu().numpy()).all() and (st.cpu().numpy() == _abi.PMX_PAIR_NONFINITE).all()

"""Oracle log-likelihood against the reference's known answers (likelihood/mod.rs:229-360,
distributions.rs:105-140, subject.rs:183-245, error_model.rs sigma rules) and hand-derived sums."""
import math

import numpy as np
import pytest

import oracle
from pharmsol_amd import (Analytical, AssayErrorModel, AssayErrorModels, Data, ErrorPoly, Ratio, Subject, _abi, synth)
from tests import models

LOG_2PI = 1.8378770664093453


def test_lognormpdf_standard_normal():  # distributions.rs:110-118, likelihood/mod.rs:343-358
    assert abs(oracle.lognormpdf(0.0, 0.0, 1.0) - (-0.5 * LOG_2PI)) < 1e-12


def test_lognormpdf_matches_exp_pdf():  # distributions.rs:121-138
    for obs, pred, sigma in [(1.0, 0.5, 0.7), (10.0, 10.5, 10.0), (8.0, 8.2, 8.0), (-2.0, 3.0, 2.5)]:
        pdf = math.exp(-0.5 * ((obs - pred) / sigma) ** 2) / (sigma * math.sqrt(2 * math.pi))
        assert abs(oracle.lognormpdf(obs, pred, sigma) - math.log(pdf)) < 1e-12


def test_sigma_rules():  # error_model.rs:1045-1080
    add = AssayErrorModel.additive(ErrorPoly(0.1, 0.2, 0.03, 0.004), 0.5)
    y = 3.0
    alpha = 0.1 + 0.2 * y + 0.03 * y * y + 0.004 * y * y * y
    assert oracle.sigma(add, y) == math.sqrt(alpha * alpha + 0.25)
    prop = AssayErrorModel.proportional(ErrorPoly(0.1, 0.2, 0.03, 0.004), 1.5)
    assert oracle.sigma(prop, y) == 1.5 * alpha
    with pytest.raises(_abi.PmxError):  # NegativeSigma
        oracle.sigma(AssayErrorModel.proportional(ErrorPoly(-1.0, 0.0, 0.0, 0.0), 1.0), 1.0)
    with pytest.raises(_abi.PmxError):  # NonFiniteSigma
        oracle.sigma(AssayErrorModel.additive(ErrorPoly(float("inf"), 0.0, 0.0, 0.0), 0.0), 1.0)


def _one_cmt():
    return Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=2).with_nstates(1).with_ndrugs(1).with_nout(1)


def test_subject_log_likelihood_is_the_sum_over_valued_observations():
    # subject.rs:63-78 + prediction.rs:105-125: missing observations contribute 0; sigma from the observation
    m = _one_cmt()
    s = (Subject.builder("ll").bolus(0.0, 100.0, 0).observation(1.0, 7.5, 0).missing_observation(2.0, 0)
         .observation(4.0, 3.0, 0).build())
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.0, 1.0, 0.0, 0.0), 0.0))  # sigma = y
    ke, v = 0.25, 10.0
    ll, st = oracle.loglik(m, m.flatten(s), em, np.array([[ke, v]]))
    want = 0.0
    for t, y in ((1.0, 7.5), (4.0, 3.0)):
        pred = 100.0 * math.exp(-ke * t) / v
        want += -0.5 * LOG_2PI - math.log(y) - (y - pred) ** 2 / (2 * y * y)
    assert abs(ll[0, 0] - want) < 1e-12 and st[0, 0] == 0


def test_subject_without_valued_observations_is_neutral():  # likelihood/mod.rs:320-325, subject.rs:183-189
    m = _one_cmt()
    subs = [Subject.builder("none").bolus(0.0, 1.0, 0).missing_observation(1.0, 0).build(), Subject.builder("empty").build()]
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(1.0, 0.0, 0.0, 0.0), 0.0))
    ll, _ = oracle.loglik(m, m.flatten(Data(subs)), em, np.array([[0.1, 1.0], [0.2, 2.0]]))
    assert (ll == 0.0).all()


def test_log_likelihood_is_non_positive_for_unit_sigma_case():  # likelihood/mod.rs:327-340
    m = Analytical.new("one_compartment", {0: Ratio(0, None)}, nparams=1).with_nstates(1).with_ndrugs(1).with_nout(1)
    s = Subject.builder("one").bolus(0.0, 1.0, 0).observation(0.0, 1.0, 0).build()
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(1.0, 0.0, 0.0, 0.0), 0.0))
    ll, _ = oracle.loglik(m, m.flatten(s), em, np.array([[0.3]]))
    assert np.isfinite(ll[0, 0]) and ll[0, 0] <= 0.0


def test_matrix_shape_and_parameter_order():  # matrix.rs:152-238: rows = subjects, columns = support points
    m = _one_cmt()
    subs = [Subject.builder(str(i)).bolus(0.0, 100.0 + i, 0).observation(1.0, 8.0, 0).observation(3.0, 4.0, 0).build()
            for i in range(3)]
    em = AssayErrorModels.empty().add(0, AssayErrorModel.proportional(ErrorPoly(0.1, 0.1, 0.0, 0.0), 2.0))
    th = np.array([[0.2, 10.0], [0.3, 12.0], [0.5, 20.0], [0.1, 9.0]])
    ll, _ = oracle.loglik(m, m.flatten(Data(subs)), em, th)
    assert ll.shape == (3, 4)
    swapped, _ = oracle.loglik(m, m.flatten(Data(subs)), em, th[:, ::-1])
    assert not np.allclose(ll, swapped)
    one, _ = oracle.loglik(m, m.flatten(subs[1]), em, th[2:3])
    assert one[0, 0] == ll[1, 2]


def test_missing_error_model_is_an_error():  # ErrorModelError::MissingErrorModel
    m = _one_cmt()
    s = Subject.builder("x").bolus(0.0, 1.0, 0).observation(1.0, 1.0, 0).build()
    with pytest.raises(_abi.PmxError) as e:
        oracle.loglik(m, m.flatten(s), AssayErrorModels.empty(), np.array([[0.1, 1.0]]))
    assert e.value.status == _abi.PMX_ERR_ERROR_MODEL


def test_non_finite_log_likelihood_is_flagged():  # prediction.rs:119-124
    m = _one_cmt()
    s = Subject.builder("x").bolus(0.0, 1.0, 0).observation(1.0, 1.0, 0).build()
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(1.0, 0.0, 0.0, 0.0), 0.0))
    ll, st = oracle.loglik(m, m.flatten(s), em, np.array([[0.1, 0.0], [0.1, 2.0]]))  # v = 0 -> pred = inf
    assert st[0, 0] == _abi.PMX_PAIR_NONFINITE and st[0, 1] == 0 and np.isfinite(ll[0, 1])


# --------------------------------------------------------------------------- censoring (distributions.rs:52-103)
def test_lognormcdf_and_ccdf_at_the_mean():  # distributions.rs:141-163
    assert abs(oracle.lognormcdf(0.0, 0.0, 1.0) - math.log(0.5)) < 1e-10
    assert abs(oracle.lognormcdf(0.0, 0.0, 1.0, upper=True) - math.log(0.5)) < 1e-10


def test_lognormcdf_tails_stay_finite():  # distributions.rs:166-187: |z| = 40 takes the asymptote
    lo = oracle.lognormcdf(-40.0, 0.0, 1.0)
    hi = oracle.lognormcdf(40.0, 0.0, 1.0, upper=True)
    assert math.isfinite(lo) and math.isfinite(hi)
    # ln Phi(-40) ~ ln phi(40) - ln 40
    want = -0.5 * 1.8378770664093453 - 800.0 - math.log(40.0)
    assert abs(lo - want) < 1e-9 and abs(hi - want) < 1e-9
    # the asymptote is written with lognormpdf(obs, pred, sigma), i.e. it carries a -ln(sigma) the true tail does not
    # have (distributions.rs:64): restated as it is, not corrected
    assert abs(oracle.lognormcdf(-20.0, 0.0, 0.5) - (-0.5 * 1.8378770664093453 - math.log(0.5) - 800.0 - math.log(40.0))) < 1e-9
    # 1 - cdf loses the upper tail long before 37 sigma: the reference returns Err there (sf == 0, z <= 37)
    with pytest.raises(_abi.PmxError):
        oracle.lognormcdf(10.0, 0.0, 1.0, upper=True)
    with pytest.raises(_abi.PmxError):
        oracle.lognormcdf(0.0, 0.0, 0.0)  # Normal::new rejects sigma = 0


def test_lognormcdf_against_scipy():
    from scipy.stats import norm

    rng = np.random.default_rng(0)
    for _ in range(200):
        obs, pred, s = rng.uniform(0, 10), rng.uniform(0, 10), rng.uniform(0.3, 3)  # |z| < 37: no asymptote branch
        assert abs(oracle.lognormcdf(obs, pred, s) - norm.logcdf(obs, pred, s)) < 1e-11 * max(1.0, abs(norm.logcdf(obs, pred, s)))
        z = (obs - pred) / s
        if z < 5:  # (1 - cdf keeps ~1e-16 absolute: compare where the survival function is not tiny)
            assert abs(oracle.lognormcdf(obs, pred, s, upper=True) - norm.logsf(obs, pred, s)) < 1e-8


def test_censored_and_error_polynomial_observations_in_the_subject_sum():
    """Prediction::log_likelihood (prediction.rs:105-125): BLOQ -> log CDF, ALOQ -> log survival; the observation's own
    ErrorPoly replaces the model's (error_model.rs:1051-1054)."""
    from pharmsol_amd import Censor

    m = models.handwritten_analytical("one_compartment", 0, 2).with_ndrugs(1)
    s = (Subject.builder("c").bolus(0.0, 100.0, 0)
         .observation(1.0, 8.0, 0)
         .censored_observation(2.0, 5.0, 0, Censor.BLOQ)
         .censored_observation(3.0, 9.0, 0, Censor.ALOQ)
         .observation_with_error(4.0, 6.0, 0, ErrorPoly(0.5, 0.0, 0.0, 0.0))
         .observation_with_error(5.0, 4.0, 0, (0.3, 0.1, 0.0, 0.0), Censor.BLOQ)
         .build())
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.1, 0.1, 0.0, 0.0), 0.2))
    th = np.array([[0.2, 10.0]])
    flat = m.flatten(s)
    ll, st = oracle.loglik(m, flat, em, th)
    pred, _ = oracle.predict(m, flat, th)
    p = pred[:, 0]

    def sig(c0, c1, y, lam=0.2):
        return math.sqrt((c0 + c1 * y) ** 2 + lam ** 2)

    want = (oracle.lognormpdf(8.0, p[0], sig(0.1, 0.1, 8.0))
            + oracle.lognormcdf(5.0, p[1], sig(0.1, 0.1, 5.0))
            + oracle.lognormcdf(9.0, p[2], sig(0.1, 0.1, 9.0), upper=True)
            + oracle.lognormpdf(6.0, p[3], sig(0.5, 0.0, 6.0))
            + oracle.lognormcdf(4.0, p[4], sig(0.3, 0.1, 4.0)))
    assert st[0, 0] == 0 and abs(ll[0, 0] - want) < 1e-12 * abs(want)

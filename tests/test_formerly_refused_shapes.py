"""Model shapes the reference accepts that earlier rounds answered with PMX_ERR_UNSUPPORTED - each one now runs on the
device and is compared with the oracle here:

  * analytical structure with a covariate-derived rate constant AND a lag time (the README model with `lag!`:
    examples/analytical_readme.rs + structs.rs:611-666),
  * pm_* (Pmetrics 1-indexed) wrappers with a lag time, and with user closures (analytical/mod.rs:62-90),
  * more than four lagged inputs on an ODE model (built-in body, custom body),
  * more than 64 boluses in one occasion of a lagged model (covered in test_full_feature_parity.py),
  * Prediction::state read-out of a model whose outputs are user code.
"""
import ctypes as C

import numpy as np
import pytest

import oracle
from pharmsol_amd import ODE, Analytical, Data, Ratio, Subject, _abi, _ffi, runtime, synth
from tests import models
from tests.test_gpu_parity import TOL_ANALYTICAL, TOL_ODE, _lag_subjects, assert_parity, rel_err


def _create(d):
    h = C.c_void_p()
    rc = _ffi.lib().pmx_model_create(C.byref(d), C.byref(h))
    if rc == _abi.PMX_OK:
        _ffi.lib().pmx_model_destroy(h)
    return rc


# --------------------------------------------------------------------------- host side (no GPU): creation succeeds
def test_no_accepted_shape_is_refused_at_creation():
    d = models.readme_analytical().desc()
    d.lag_param[0] = 0
    assert _create(d) == _abi.PMX_OK
    d = Analytical.new("pm_two_compartments", {0: Ratio(1, 3)}, nparams=5, lag={1: 4}).with_nstates(3).with_ndrugs(2).with_nout(
        1).desc()
    assert _create(d) == _abi.PMX_OK
    d = _five_lag_ode().desc()
    assert _create(d) == _abi.PMX_OK


def _readme_with_lag():
    m = models.readme_analytical()
    m.params = list(m.params) + ["tlag", "f"]
    m.nparams = 5
    m.lag = {"oral": "tlag"}
    m.fa = {"oral": "f"}
    return m


def _five_lag_ode():
    # six routes into a three-compartment body, five of them lagged
    m = ODE.new("three_cmt_iv", {0: Ratio(0, 5)}, nparams=11, h_max=0.05).with_nstates(3).with_ndrugs(6).with_nout(1)
    m.lag = {str(i): 6 + i for i in range(5)}
    m.bolus_dest = {i: i % 3 for i in range(6)}
    return m


def _relabel(subs, rng=None, input=None, outeq=None, wt_knots=0, boluses_only=False):
    """The same designs with other input / output labels and `wt` covariate knots per occasion."""
    for s in subs:
        for occ in s.occasions:
            if boluses_only:  # (the README model declares one bolus route)
                occ.events = [ev for ev in occ.events if not hasattr(ev, "duration")]
            for ev in occ.events:
                if hasattr(ev, "input") and input is not None:
                    ev.input = input
                if hasattr(ev, "outeq") and outeq is not None:
                    ev.outeq = outeq
            for k in range(wt_knots):
                occ.covariates.add_observation("wt", 12.0 * k, float(rng.uniform(40, 110)))
    return subs


def _many_route_subjects(rng, n, n_inputs):
    subs = []
    for i in range(n):
        b = Subject.builder(f"r{i}")
        for occ in range(1 + int(rng.integers(0, 2))):
            if occ:
                b = b.reset()
            for _ in range(int(rng.integers(2, 9))):
                b = b.bolus(float(np.round(rng.uniform(0, 12), 1)), float(rng.uniform(50, 300)), int(rng.integers(0, n_inputs)))
            if rng.random() < 0.5:
                b = b.infusion(float(np.round(rng.uniform(0, 8), 1)), float(rng.uniform(50, 300)), int(rng.integers(0, n_inputs)),
                               float(np.round(rng.uniform(0.5, 3), 1)))
            for _ in range(int(rng.integers(2, 7))):
                b = b.missing_observation(float(np.round(rng.uniform(0, 20) * 2) / 2), 0)
        subs.append(b.build())
    return subs


# --------------------------------------------------------------------------- device
@pytest.mark.gpu
@pytest.mark.parametrize("n_support,batch", [(70, False), (5, False), (0, True)])
def test_covariate_derived_rate_constant_with_lag_and_bioavailability(n_support, batch):
    rng = np.random.default_rng(501)
    m = _readme_with_lag()
    subs = _relabel(_lag_subjects(rng, 30), rng, input="oral", outeq="cp", wt_knots=2, boluses_only=True)
    flat = m.flatten(Data(subs))
    n = len(subs) if batch else n_support
    th = np.stack([rng.uniform(0.8, 2.0, n), rng.uniform(0.05, 0.3, n), rng.uniform(10, 50, n),
                   np.round(rng.uniform(0, 3, n) * 2) / 2, rng.uniform(0.3, 1.0, n)], axis=1)
    assert_parity(m, flat, th, TOL_ANALYTICAL, batch=batch, expect_kernel="pmx_jit_analytical")


@pytest.mark.gpu
def test_pmetrics_wrapper_with_lag_matches_the_native_model():
    m = Analytical.new("pm_one_compartment_with_absorption", {0: Ratio(2, 2)}, nparams=5, lag={1: 3}, fa={1: 4}).with_nstates(
        3).with_ndrugs(2).with_nout(1)
    m0 = Analytical.new("one_compartment_with_absorption", {0: Ratio(1, 2)}, nparams=5, lag={0: 3}, fa={0: 4}).with_nstates(
        2).with_ndrugs(1).with_nout(1)
    subs0 = _lag_subjects(np.random.default_rng(502), 40)
    rng = np.random.default_rng(5020)
    subs = _relabel(_lag_subjects(np.random.default_rng(502), 40), input=1)  # the same design on the wrapper's 1-based input
    th = np.stack([rng.uniform(1.0, 2.0, 70), rng.uniform(0.05, 0.3, 70), rng.uniform(10, 50, 70),
                   np.round(rng.uniform(0, 3, 70) * 2) / 2, rng.uniform(0.3, 1.0, 70)], axis=1)
    got, _ = assert_parity(m, m.flatten(Data(subs)), th, TOL_ANALYTICAL, expect_kernel="pmx_jit_analytical")
    want, _ = oracle.predict(m0, m0.flatten(Data(subs0)), th)
    assert rel_err(got, want).max() <= TOL_ANALYTICAL


@pytest.mark.gpu
def test_pmetrics_wrapper_with_user_closures():
    rng = np.random.default_rng(503)
    src = """
    PMX_DEVICE void pmx_route_lag(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                                  const double* derived, double* lag) { lag[1] = p[3] * (cov[0] / 70.0); }
    PMX_DEVICE void pmx_outputs(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                                const double* derived, double* y) { y[0] = x[2] / (p[2] * cov[0] / 70.0); }
    """
    m = Analytical.user(src, eq="pm_one_compartment_with_absorption", nstates=3, nparams=4, ndrugs=2, nout=1, covariates=["wt"])
    subs = _relabel(_lag_subjects(rng, 30), rng, input=1, wt_knots=1)
    th = np.stack([rng.uniform(1.0, 2.0, 70), rng.uniform(0.05, 0.3, 70), rng.uniform(10, 50, 70),
                   np.round(rng.uniform(0, 3, 70) * 2) / 2], axis=1)
    assert_parity(m, m.flatten(Data(subs)), th, TOL_ANALYTICAL, expect_kernel="pmx_jit_analytical")


@pytest.mark.gpu
@pytest.mark.parametrize("n_support,batch", [(70, False), (4, False), (0, True)])
def test_five_lagged_inputs_on_a_built_in_ode_body(n_support, batch):
    rng = np.random.default_rng(504)
    m = _five_lag_ode()
    subs = _many_route_subjects(rng, 24, 6)
    n = len(subs) if batch else n_support
    th = np.concatenate([rng.uniform(0.05, 0.4, (n, 5)), rng.uniform(5, 50, (n, 1)), np.round(rng.uniform(0, 3, (n, 5)) * 2) / 2],
                        axis=1)
    assert_parity(m, m.flatten(Data(subs)), th, TOL_ODE, batch=batch, expect_kernel="pmx_jit_ode_user")


@pytest.mark.gpu
def test_five_lagged_inputs_on_a_custom_ode_body():
    rng = np.random.default_rng(505)
    src = """
    PMX_DEVICE void pmx_dynamics(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                                 const double* derived, double* dx) {
      dx[0] = -p[0] * x[0] + rateiv[0] + rateiv[2] + rateiv[4];
      dx[1] = p[0] * x[0] - p[1] * x[1] + rateiv[1] + rateiv[3] + rateiv[5];
    }
    PMX_DEVICE void pmx_outputs(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                                const double* derived, double* y) { y[0] = x[1] / p[2]; }
    """
    m = ODE.custom(src, nstates=2, nparams=8, ndrugs=6, nout=1, h_max=0.05)
    m.lag = {str(i): 3 + i for i in range(5)}
    m.bolus_dest = {i: i % 2 for i in range(6)}
    oracle.compile_custom(src)
    subs = _many_route_subjects(rng, 24, 6)
    th = np.concatenate([rng.uniform(0.3, 1.5, (70, 1)), rng.uniform(0.05, 0.4, (70, 1)), rng.uniform(5, 50, (70, 1)),
                         np.round(rng.uniform(0, 3, (70, 5)) * 2) / 2], axis=1)
    assert_parity(m, m.flatten(Data(subs)), th, TOL_ODE, expect_kernel="pmx_jit_ode_user")


@pytest.mark.gpu
def test_state_read_out_of_models_whose_outputs_are_user_code():
    rng = np.random.default_rng(506)
    src = """
    PMX_DEVICE void pmx_outputs(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                                const double* derived, double* y) { y[0] = x[1] / p[2]; }
    """
    m = Analytical.user(src, eq="one_compartment_with_absorption", nstates=2, nparams=3, ndrugs=1, nout=1)
    subs = _lag_subjects(rng, 10)
    flat = m.flatten(Data(subs))
    th = np.stack([rng.uniform(1.0, 2.0, 40), rng.uniform(0.05, 0.3, 40), rng.uniform(10, 50, 40)], axis=1)
    pop = runtime.DevicePopulation(flat, 0)
    states = runtime.predict_states(m, pop, th).cpu().numpy()  # [n_obs, 2, P]
    pred, _ = runtime.predict(m, pop, th)
    np.testing.assert_allclose(states[:, 1, :] / th[None, :, 2], pred.cpu().numpy(), rtol=1e-12)
    m_gut = Analytical.new("one_compartment_with_absorption", {0: Ratio(0, None)}, nparams=3).with_nstates(2).with_ndrugs(1).with_nout(1)
    want, _ = oracle.predict(m_gut, m_gut.flatten(Data(subs)), th)
    assert rel_err(states[:, 0, :], want).max() <= TOL_ANALYTICAL
    # ... and a custom ODE body
    osrc = """
    PMX_DEVICE void pmx_dynamics(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                                 const double* derived, double* dx) {
      dx[0] = -p[0] * x[0];  dx[1] = p[0] * x[0] - p[1] * x[1] + rateiv[0];  /* (infusions enter the central compartment) */
    }
    PMX_DEVICE void pmx_outputs(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                                const double* derived, double* y) { y[0] = x[1] / p[2]; }
    """
    mo = ODE.custom(osrc, nstates=2, nparams=3, h_max=0.01)
    so = runtime.predict_states(mo, pop, th).cpu().numpy()
    assert rel_err(so[:, 0, :], want).max() <= TOL_ODE
    po, _ = runtime.predict(mo, pop, th)
    np.testing.assert_allclose(so[:, 1, :] / th[None, :, 2], po.cpu().numpy(), rtol=1e-12)

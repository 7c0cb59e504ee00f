"""User ODE models compiled at run time: the same source text is compiled by hiprtc for the
device (inside libpmx_hip.so) and by gcc for the CPU oracle (oracle.compile_custom).  CPU half: the compile path
(hiprtc needs no GPU), diagnostics, and the oracle's custom-body walker against closed forms and the built-in
diffeq bodies.  GPU half: parity of every walker variant."""
import numpy as np
import pytest

import oracle
from pharmsol_amd import (ODE, Analytical, AssayErrorModel, AssayErrorModels, Data, ErrorPoly, Ratio, Subject, _abi,
                          runtime, synth)
from tests import models

SIG = ("double t, const double* x, const double* p, const double* cov, const double* rateiv, "
       "const double* derived, double* ")

ONE_CMT = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{
  const double ke = p[0];
  dx[0] = -ke * x[0] + rateiv[0];
}}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0] / p[1]; }}
"""

# Michaelis-Menten elimination + a peripheral compartment, two outputs, an initial amount: nothing built in
MM_TWO_OUT = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{
  const double vmax = p[0], km = p[1], v = p[2], q = p[3];
  const double c = x[0] / v;
  dx[0] = -vmax * c / (km + c) - q * x[0] + q * x[1] + rateiv[0];
  dx[1] = q * x[0] - q * x[1];
}}
PMX_DEVICE void pmx_outputs({SIG}y) {{
  y[0] = x[0] / p[2];
  y[1] = x[1];
}}
PMX_DEVICE void pmx_init({SIG}xi) {{ xi[1] = p[4]; }}
"""

# non-autonomous: dx = a t^3 (RK4 integrates a cubic forcing exactly) minus first-order loss
FORCED = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{ dx[0] = p[0] * t * t * t; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0] + 0.0 * t; }}
"""


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-9)


def rel_err_floor(a, b):
    """Relative error with a floor at 1e-3 of the largest value: an adaptive solver controls atol + rtol |x|, i.e.
    it does not promise relative accuracy on a concentration that has decayed by ten orders of magnitude."""
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * np.abs(b).max())


# --------------------------------------------------------------------------- compile path (no GPU needed)
def test_translation_unit_wraps_the_user_source_into_the_shared_walkers():
    m = ODE.custom(ONE_CMT, nstates=1, nparams=2)
    tu = runtime.jit_translation_unit(m)
    assert '#include "pmx_ode.hpp"' in tu and "pmx_dynamics(t, x, p, cov, r, nullptr, dx)" in tu
    assert "NS = 1, NP = 2" in tu and tu.count("extern \"C\" __global__") == 16  # grid/pair x lag x loglik x solver
    runtime.DeviceModel(m)  # hiprtc compiles for gfx950 without a device


def test_compile_errors_come_back_with_the_users_line_numbers():
    bad = ODE.custom(ONE_CMT.replace("ke * x[0]", "ke * z[0]"), nstates=1, nparams=2)
    with pytest.raises(_abi.PmxError) as e:
        runtime.DeviceModel(bad)
    assert e.value.status == _abi.PMX_ERR_INVALID_ARGUMENT
    assert "model:4" in str(e.value) and "undeclared identifier 'z'" in str(e.value)


def test_descriptor_rules_for_custom_models():
    m = ODE.custom(ONE_CMT, nstates=1, nparams=2)
    d = m.desc()
    assert d.kernel == _abi.PMX_ODE_CUSTOM
    import ctypes as C

    from pharmsol_amd import _ffi

    h = C.c_void_p()
    assert _ffi.lib().pmx_model_create(C.byref(d), C.byref(h)) == _abi.PMX_ERR_INVALID_ARGUMENT  # needs the source
    d.pmetrics_indexing = 1
    assert _ffi.lib().pmx_model_create_custom(C.byref(d), ONE_CMT.encode(), 0, C.byref(h)) == _abi.PMX_ERR_INVALID_ARGUMENT
    d = m.desc()
    d.rk4_h_max = 0.0
    assert _ffi.lib().pmx_model_create_custom(C.byref(d), ONE_CMT.encode(), 0, C.byref(h)) == _abi.PMX_ERR_INVALID_ARGUMENT


# --------------------------------------------------------------------------- oracle: custom bodies on the CPU
def test_oracle_custom_one_compartment_matches_the_closed_form():
    oracle.compile_custom(ONE_CMT)
    m = ODE.custom(ONE_CMT, nstates=1, nparams=2, h_max=0.01)
    s = (Subject.builder("a").bolus(0.0, 100.0, 0).infusion(1.0, 50.0, 0, 2.0).missing_observation(0.5, 0)
         .missing_observation(2.0, 0).missing_observation(6.0, 0).build())
    th = np.array([[0.2, 10.0], [0.5, 20.0]])
    got, _ = oracle.predict(m, m.flatten(s), th)
    ma = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=2).with_nstates(1).with_ndrugs(1).with_nout(1)
    want, _ = oracle.predict(ma, ma.flatten(s), th)
    assert rel_err(got, want).max() < 1e-9


def test_oracle_custom_body_equals_the_builtin_body():
    src = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{ const double cc = x[0] / p[2]; dx[0] = -p[0] * cc / (p[1] + cc) + rateiv[0]; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0] / p[2]; }}
"""
    oracle.compile_custom(src)
    rng = np.random.default_rng(0)
    subs = [models.random_subject(rng) for _ in range(12)]
    mc = ODE.custom(src, nstates=1, nparams=3, h_max=0.02)
    mb = ODE.new("one_cmt_mm", {0: Ratio(0, 2)}, nparams=3, h_max=0.02).with_nstates(1).with_ndrugs(1).with_nout(1)
    th = np.stack([rng.uniform(5, 30, 6), rng.uniform(1, 10, 6), rng.uniform(10, 40, 6)], axis=1)
    got, _ = oracle.predict(mc, mc.flatten(Data(subs)), th)
    want, _ = oracle.predict(mb, mb.flatten(Data(subs)), th)
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=0)


def test_oracle_time_dependent_body_sees_the_stage_times():
    oracle.compile_custom(FORCED)
    m = ODE.custom(FORCED, nstates=1, nparams=1, h_max=0.25)
    s = Subject.builder("f").missing_observation(0.0, 0).missing_observation(1.0, 0).missing_observation(3.0, 0).build()
    got, _ = oracle.predict(m, m.flatten(s), np.array([[2.0]]))
    np.testing.assert_allclose(got[:, 0], [0.0, 2.0 * 1.0 / 4.0, 2.0 * 81.0 / 4.0], rtol=1e-13, atol=1e-15)


# --------------------------------------------------------------------------- device parity
def _gpu(model, flat, theta, batch=False):
    import torch

    pop = runtime.DevicePopulation(flat, 0)
    pred, st = runtime.predict(model, pop, np.ascontiguousarray(theta), batch=batch)
    torch.cuda.synchronize()
    return pred.cpu().numpy(), st.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PMX_FUZZ_CUSTOM_ODE", "1"))))  # (more populations: set it)
@pytest.mark.parametrize("n_support,batch", [(70, False), (4, False), (0, True)])
def test_gpu_custom_model_all_lane_mappings(n_support, batch, seed):
    oracle.compile_custom(MM_TWO_OUT, has_init=True)
    rng = np.random.default_rng(21 + 1000 * seed)
    m = ODE.custom(MM_TWO_OUT, nstates=2, nparams=5, nout=2, has_init=True, h_max=0.02)
    subs = []
    for i in range(40 if seed == 0 else int(rng.integers(2, 60))):
        s = models.random_subject(rng, multi_occasion=(i % 3 == 0))
        for occ in s.occasions:  # alternate the two outputs
            for k, ev in enumerate(e for e in occ.events if hasattr(e, "outeq")):
                ev.outeq = k % 2
        subs.append(s)
    flat = m.flatten(Data(subs))
    n = len(subs) if batch else n_support
    th = np.stack([rng.uniform(5, 30, n), rng.uniform(1, 10, n), rng.uniform(10, 40, n), rng.uniform(0.05, 0.5, n),
                   rng.uniform(0, 20, n)], axis=1)
    got, st = _gpu(m, flat, th, batch=batch)
    want, wst = (oracle.predict_batch if batch else oracle.predict)(m, flat, th)
    assert runtime.last_kernel_name() == ("pmx_jit_ode_rk4_pair" if (batch or n_support < 32) else "pmx_jit_ode_rk4_grid")
    np.testing.assert_array_equal(st, wst)
    assert rel_err(got, want).max() < 1e-9


@pytest.mark.gpu
def test_gpu_custom_model_time_dependent_lag_and_loglik():
    from tests.test_gpu_parity import _lag_subjects

    rng = np.random.default_rng(22)
    # forced one-compartment: dx = -ke x + rateiv + a sin(w t)
    src = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{ dx[0] = -p[0] * x[0] + rateiv[0] + p[2] * sin(0.3 * t); }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0] / p[1]; }}
"""
    oracle.compile_custom(src)
    m = ODE.custom(src, nstates=1, nparams=5, lag={0: 3}, fa={0: 4}, h_max=0.02)
    subs = _lag_subjects(rng, 40)
    flat = m.flatten(Data(subs))
    for n in (64, 3):
        th = np.stack([rng.uniform(0.05, 0.4, n), rng.uniform(5, 40, n), rng.uniform(0, 5, n),
                       np.round(rng.uniform(0, 3, n) * 2) / 2, rng.uniform(0.3, 1.0, n)], axis=1)
        got, st = _gpu(m, flat, th)
        want, wst = oracle.predict(m, flat, th)
        assert runtime.last_kernel_name().startswith("pmx_jit_ode_rk4_" + ("grid<lag>" if n >= 32 else "pair<lag>"))
        np.testing.assert_array_equal(st, wst)
        assert rel_err(got, want).max() < 1e-9
    # fused log-likelihood through the compiled module's <ll> entry points
    m2 = ODE.custom(src, nstates=1, nparams=3, h_max=0.02)
    subs2 = [models.random_subject(rng) for _ in range(30)]
    flat2 = m2.flatten(Data(subs2))
    th2 = np.stack([rng.uniform(0.05, 0.4, 40), rng.uniform(5, 40, 40), rng.uniform(0, 5, 40)], axis=1)
    pred, _ = oracle.predict(m2, flat2, th2[:1])
    vals = np.abs(pred[:, 0]) * np.exp(rng.normal(0, 0.2, pred.shape[0])) + 0.05
    flat2.ev_value = flat2.ev_value.copy()
    flat2.ev_value[flat2.ev_kind == _abi.PMX_EV_OBSERVATION] = vals
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
    import torch

    pop = runtime.DevicePopulation(flat2, 0)
    ll, st = runtime.loglik(m2, pop, em, th2)
    torch.cuda.synchronize()
    want, wst = oracle.loglik(m2, flat2, em, th2)
    np.testing.assert_array_equal(st.cpu().numpy(), wst)
    assert (np.abs(ll.cpu().numpy() - want) / np.maximum(np.abs(want), 1.0)).max() < 1e-9


# --------------------------------------------------------------------------- adaptive solver (PMX_SOLVER_DOPRI5)
def _mm_subjects(rng, n):
    subs = []
    for i in range(n):
        b = Subject.builder(f"m{i}").bolus(0.0, float(rng.uniform(100, 600)), 0)
        if i % 2:
            b = b.infusion(float(rng.uniform(1, 6)), float(rng.uniform(100, 400)), 0, float(rng.uniform(0.5, 3)))
        for t in sorted(rng.uniform(0.1, 48, 6)):
            b = b.missing_observation(float(t), 0)
        subs.append(b.build())
    return subs


def test_oracle_dopri5_against_closed_form_and_fine_rk4():
    rng = np.random.default_rng(31)
    subs = _mm_subjects(rng, 10)
    ma = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=2).with_nstates(1).with_ndrugs(1).with_nout(1)
    mo = (ODE.new("one_cmt_iv", {0: Ratio(0, 1)}, nparams=2, h_max=4.0).with_nstates(1).with_ndrugs(1).with_nout(1)
          .with_solver("dopri5").with_tolerances(1e-9, 1e-9))
    th = np.stack([rng.uniform(0.05, 1.0, 8), rng.uniform(10, 50, 8)], axis=1)
    want, _ = oracle.predict(ma, ma.flatten(Data(subs)), th)
    got, st = oracle.predict(mo, mo.flatten(Data(subs)), th)
    assert (st == 0).all() and rel_err_floor(got, want).max() < 1e-7
    # looser tolerances -> proportionally larger error, still far inside the 1e-4 ODE budget at the reference's defaults
    mo.with_tolerances(1e-4, 1e-4)
    got, _ = oracle.predict(mo, mo.flatten(Data(subs)), th)
    assert 1e-9 < rel_err_floor(got, want).max() < 1e-3
    # nonlinear (Michaelis-Menten): against fixed-step RK4 at h = 0.002
    mm_fix = ODE.new("one_cmt_mm", {0: Ratio(0, 2)}, nparams=3, h_max=0.002).with_nstates(1).with_ndrugs(1).with_nout(1)
    mm_ad = (ODE.new("one_cmt_mm", {0: Ratio(0, 2)}, nparams=3, h_max=8.0).with_nstates(1).with_ndrugs(1).with_nout(1)
             .with_solver("dopri5").with_tolerances(1e-10, 1e-10))
    th3 = np.stack([rng.uniform(5, 30, 6), rng.uniform(1, 10, 6), rng.uniform(10, 40, 6)], axis=1)
    a, _ = oracle.predict(mm_fix, mm_fix.flatten(Data(subs)), th3)
    b, _ = oracle.predict(mm_ad, mm_ad.flatten(Data(subs)), th3)
    assert rel_err_floor(b, a).max() < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("n_support,batch", [(70, False), (4, False), (0, True)])
def test_gpu_dopri5_builtin_models(n_support, batch):
    rng = np.random.default_rng(32)
    subs = _mm_subjects(rng, 50)
    m = (ODE.new("two_cmt_iv", {0: Ratio(0, 3)}, nparams=4, h_max=6.0).with_nstates(2).with_ndrugs(1).with_nout(1)
         .with_solver("dopri5").with_tolerances(1e-8, 1e-8))
    flat = m.flatten(Data(subs))
    th = synth.theta_c3(len(subs) if batch else n_support)
    got, st = _gpu(m, flat, th, batch=batch)
    want, wst = (oracle.predict_batch if batch else oracle.predict)(m, flat, th)
    assert runtime.last_kernel_name() == ("pmx_ode_dopri5_pair" if (batch or n_support < 32) else "pmx_ode_dopri5_grid")
    np.testing.assert_array_equal(st, wst)
    # same algorithm, same tolerances; FMA contraction may move a step boundary, so agreement is at the solver's
    # tolerance rather than at rounding level
    assert rel_err_floor(got, want).max() < 1e-6
    ma = Analytical.new("two_compartments", {0: Ratio(0, 3)}, nparams=4).with_nstates(2).with_ndrugs(1).with_nout(1)
    exact, _ = (oracle.predict_batch if batch else oracle.predict)(ma, ma.flatten(Data(subs)), th)
    assert rel_err_floor(got, exact).max() < 1e-5


@pytest.mark.gpu
def test_gpu_dopri5_custom_model_with_lag_and_step_underflow_flag():
    from tests.test_gpu_parity import _lag_subjects

    rng = np.random.default_rng(33)
    src = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{ dx[0] = -p[0] * x[0] + rateiv[0] + p[2] * sin(0.3 * t); }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0] / p[1]; }}
"""
    oracle.compile_custom(src)
    m = (ODE.custom(src, nstates=1, nparams=5, lag={0: 3}, fa={0: 4}, h_max=5.0).with_solver("dopri5")
         .with_tolerances(1e-8, 1e-8))
    flat = m.flatten(Data(_lag_subjects(rng, 30)))
    for n in (64, 3):
        th = np.stack([rng.uniform(0.05, 0.4, n), rng.uniform(5, 40, n), rng.uniform(0, 5, n),
                       np.round(rng.uniform(0, 3, n) * 2) / 2, rng.uniform(0.3, 1.0, n)], axis=1)
        got, st = _gpu(m, flat, th)
        want, wst = oracle.predict(m, flat, th)
        assert runtime.last_kernel_name().startswith("pmx_jit_ode_dopri5_" + ("grid<lag>" if n >= 32 else "pair<lag>"))
        np.testing.assert_array_equal(st, wst)
        assert rel_err_floor(got, want).max() < 1e-6
    # a right-hand side that blows up in finite time: the controller runs out of step size -> flagged, NaN rows
    boom = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{ dx[0] = p[0] * x[0] * x[0]; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0]; }}
"""
    oracle.compile_custom(boom)
    mb = ODE.custom(boom, nstates=1, nparams=1, h_max=1.0).with_solver("dopri5").with_tolerances(1e-6, 1e-6)
    s = Subject.builder("b").bolus(0.0, 1.0, 0).missing_observation(0.5, 0).missing_observation(2.0, 0).build()
    th = np.array([[1.0]] * 40)  # x' = x^2, x(0) = 1: singular at t = 1
    got, st = _gpu(mb, mb.flatten(s), th)
    want, wst = oracle.predict(mb, mb.flatten(s), th)
    np.testing.assert_array_equal(st, wst)
    assert (st == _abi.PMX_PAIR_SOLVER_FAIL).all() and np.isnan(got[1]).all()
    assert rel_err_floor(got[0], np.full(40, 2.0)).max() < 1e-5  # x(0.5) = 1 / (1 - 0.5)


# --------------------------------------------------------------------------- covariates inside custom bodies
# examples/covariates.rs: one-compartment oral model whose elimination is scaled by time-varying creatinine and age
COV_SRC = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{
  const double ka = p[0], ke = p[1];
  const double scaled_ke = ke * pow(cov[0] / 75.0, 0.75) * pow(cov[1] / 25.0, 0.5);
  dx[0] = -ka * x[0];
  dx[1] = ka * x[0] - scaled_ke * x[1];
}}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[1] / p[3]; }}
"""


def _covariates_example_subject():
    return (Subject.builder("id1").bolus(0.0, 100.0, 0).repeat(2, 2.0).observation(0.5, 0.1, 0).observation(1.0, 0.4, 0)
            .observation(2.0, 1.0, 0).observation(2.5, 1.1, 0).covariate("creatinine", 0.0, 80.0)
            .covariate("creatinine", 1.0, 40.0).covariate("age", 0.0, 25.0).missing_observation(8.0, 0).build())


def test_oracle_covariates_example_against_piecewise_reasoning():
    """examples/covariates.rs through the oracle: with constant covariates the model is the closed-form
    one_compartment_with_absorption at ke' = ke (cr/75)^0.75 (age/25)^0.5; with the example's time-varying creatinine
    it must sit between the two constant-creatinine solutions after t = 0."""
    oracle.compile_custom(COV_SRC)
    m = ODE.custom(COV_SRC, nstates=2, nparams=4, covariates=["creatinine", "age"], lag={0: 2}, h_max=0.01)
    th = np.array([[1.0, 0.2, 0.0, 70.0]])
    s_var = _covariates_example_subject()
    got, st = oracle.predict(m, m.flatten(s_var), th)
    assert (st == 0).all() and np.isfinite(got).all()

    def const_cr(cr):
        s = (Subject.builder("c").bolus(0.0, 100.0, 0).repeat(2, 2.0).missing_observation(0.5, 0).missing_observation(1.0, 0)
             .missing_observation(2.0, 0).missing_observation(2.5, 0).covariate("creatinine", 0.0, cr).covariate("age", 0.0, 25.0)
             .missing_observation(8.0, 0).build())
        p, _ = oracle.predict(m, m.flatten(s), th)
        ke = 0.2 * (cr / 75.0) ** 0.75
        ma = Analytical.new("one_compartment_with_absorption", {0: Ratio(1, 2)}, nparams=3).with_nstates(2).with_ndrugs(1).with_nout(1)
        pa, _ = oracle.predict(ma, ma.flatten(s), np.array([[1.0, ke, 70.0]]))
        assert rel_err(p, pa).max() < 1e-8  # constant covariates: the closed form
        return p[:, 0]

    hi, lo = const_cr(80.0), const_cr(40.0)  # more creatinine clearance -> faster elimination -> lower curve
    assert (got[1:, 0] > hi[1:]).all() and (got[1:, 0] < lo[1:]).all()


@pytest.mark.gpu
def test_gpu_custom_model_with_time_varying_covariates():
    oracle.compile_custom(COV_SRC)
    rng = np.random.default_rng(41)
    for solver in ("rk4", "dopri5"):
        m = ODE.custom(COV_SRC, nstates=2, nparams=4, covariates=["creatinine", "age"], lag={0: 2}, h_max=0.02)
        if solver == "dopri5":
            m = m.with_step(2.0).with_solver("dopri5").with_tolerances(1e-8, 1e-8)
        subs = [_covariates_example_subject()]
        for i in range(40):
            b = Subject.builder(f"s{i}").bolus(0.0, float(rng.uniform(50, 200)), 0).repeat(int(rng.integers(0, 3)), 6.0)
            for t in sorted(rng.uniform(0.2, 30, 6)):
                b = b.missing_observation(float(t), 0)
            for t in sorted(rng.uniform(0, 24, int(rng.integers(1, 4)))):
                b = b.covariate("creatinine", float(np.round(t, 1)), float(rng.uniform(30, 120)))
            b = b.covariate("age", 0.0, float(rng.uniform(20, 80)))
            if i % 3 == 0:
                b = b.reset().bolus(0.0, 80.0, 0).missing_observation(3.0, 0).covariate("creatinine", 0.0, 60.0).covariate("age", 0.0, 50.0)
            subs.append(b.build())
        flat = m.flatten(Data(subs))
        for n in (48, 3):
            th = np.stack([rng.uniform(0.5, 2.0, n), rng.uniform(0.05, 0.5, n), np.round(rng.uniform(0, 2, n) * 2) / 2,
                           rng.uniform(20, 90, n)], axis=1)
            got, st = _gpu(m, flat, th)
            want, wst = oracle.predict(m, flat, th)
            np.testing.assert_array_equal(st, wst)
            tol = 1e-9 if solver == "rk4" else 1e-6
            assert rel_err_floor(got, want).max() < tol, (solver, n, rel_err_floor(got, want).max())

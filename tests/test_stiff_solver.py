"""PMX_SOLVER_ROS2 - the stiff option of ``ODE::with_solver`` (the role of the reference's default ``OdeSolver::Bdf`` and of
``Sdirk``, ode/mod.rs:60-77, which this build replaces: SURVEY.md §8 a23).  An L-stable second-order Rosenbrock method
with the adaptive step control of the DOPRI5 path (csrc/pmx_ode.hpp ros2_try, oracle/pmx_oracle.c ros2_try).  Pinned by
the closed forms of the analytical back-end on STIFF parameter sets (absorption / distribution 10^3-10^4 times faster than
elimination), by fixed-step RK4 on the non-linear body, and - on the GPU - by the oracle's restatement of the same method."""
import numpy as np
import pytest

import oracle
from pharmsol_amd import ODE, Analytical, Data, Ratio, Subject, _abi, runtime
from tests.test_custom_models import _gpu, _mm_subjects, rel_err_floor


def _oral_subjects(rng, n):
    subs = []
    for i in range(n):
        b = Subject.builder(f"s{i}").bolus(0.0, float(rng.uniform(100, 600)), 0)
        if i % 2:
            b = b.infusion(float(rng.uniform(1, 6)), float(rng.uniform(100, 400)), 0, float(rng.uniform(0.5, 3)))
        if i % 3 == 0:
            b = b.bolus(12.0, float(rng.uniform(100, 300)), 0)
        for t in sorted(rng.uniform(0.05, 48, 7)):
            b = b.missing_observation(float(t), 0)
        subs.append(b.build())
    return subs


def _stiff_theta(rng, n):
    # [ke, ka, kcp, kpc, v]: ka 200..5000 /h beside ke 0.05..0.3 /h
    return np.stack([rng.uniform(0.05, 0.3, n), np.exp(rng.uniform(np.log(200.0), np.log(5000.0), n)), rng.uniform(0.2, 2.0, n),
                     rng.uniform(0.1, 1.0, n), rng.uniform(10, 50, n)], axis=1)


def _models(rtol):
    mo = (ODE.new("two_cmt_oral", {0: Ratio(1, 4)}, nparams=5, h_max=48.0).with_nstates(3).with_ndrugs(1).with_nout(1)
          .with_solver("ros2").with_tolerances(rtol, rtol))
    ma = Analytical.new("two_compartments_with_absorption", {0: Ratio(1, 4)}, nparams=5).with_nstates(3).with_ndrugs(1).with_nout(1)
    return mo, ma


def test_descriptor_and_aliases():
    m = ODE.new("one_cmt_iv", {0: Ratio(0, 1)}, nparams=2).with_solver("stiff")
    assert m.desc().ode_solver == _abi.PMX_SOLVER_ROS2 == 2
    assert ODE.new("one_cmt_iv", {0: Ratio(0, 1)}, nparams=2).with_solver("ros2").desc().ode_solver == _abi.PMX_SOLVER_ROS2


def test_oracle_ros2_on_stiff_systems_against_the_closed_form():
    rng = np.random.default_rng(71)
    subs = _oral_subjects(rng, 8)
    th = _stiff_theta(rng, 6)
    for tol, bound in ((1e-5, 2e-4), (1e-7, 3e-6)):
        mo, ma = _models(tol)
        want, _ = oracle.predict(ma, ma.flatten(Data(subs)), th)
        got, st = oracle.predict(mo, mo.flatten(Data(subs)), th)
        assert (st == 0).all()
        err = rel_err_floor(got, want).max()
        assert err < bound, (tol, err)


def test_oracle_ros2_takes_far_fewer_right_hand_sides_than_the_explicit_pair():
    """Stability, not accuracy, bounds an explicit method on a stiff system: at ka = 5e5 /h DOPRI5 cannot step past
    3.3 / 5e5 h however loose the tolerance (7 x 10^6 steps over 48 h), the L-stable method steps at the accuracy the slow
    modes need.  (ROS2 estimates its error from a FIRST-order companion, so its steps go with tol^(1/2): at mild stiffness -
    ka of a few thousand - the explicit pair is still the faster of the two; the option is for the systems it cannot do.)"""
    import time

    s = (Subject.builder("s").bolus(0.0, 100.0, 0).missing_observation(24.0, 0).missing_observation(48.0, 0).build())
    th = np.array([[0.1, 5.0e5, 0.5, 0.3, 20.0]])
    mo, ma = _models(1e-5)
    md = (ODE.new("two_cmt_oral", {0: Ratio(1, 4)}, nparams=5, h_max=48.0).with_nstates(3).with_ndrugs(1).with_nout(1)
          .with_solver("dopri5").with_tolerances(1e-5, 1e-5))
    want, _ = oracle.predict(ma, ma.flatten(s), th)
    t0 = time.perf_counter()
    a, _ = oracle.predict(mo, mo.flatten(s), th)
    t_ros = time.perf_counter() - t0
    t0 = time.perf_counter()
    b, _ = oracle.predict(md, md.flatten(s), th)
    t_dp = time.perf_counter() - t0
    assert rel_err_floor(a, want).max() < 2e-4 and rel_err_floor(b, want).max() < 2e-4
    assert t_dp > 5 * t_ros, (t_dp, t_ros)


def test_oracle_ros2_nonlinear_body_against_fine_rk4():
    rng = np.random.default_rng(72)
    subs = _mm_subjects(rng, 6)
    fix = ODE.new("one_cmt_mm", {0: Ratio(0, 2)}, nparams=3, h_max=0.002).with_nstates(1).with_ndrugs(1).with_nout(1)
    ros = (ODE.new("one_cmt_mm", {0: Ratio(0, 2)}, nparams=3, h_max=8.0).with_nstates(1).with_ndrugs(1).with_nout(1)
           .with_solver("ros2").with_tolerances(1e-7, 1e-7))
    th = np.stack([rng.uniform(5, 30, 5), rng.uniform(1, 10, 5), rng.uniform(10, 40, 5)], axis=1)
    a, _ = oracle.predict(fix, fix.flatten(Data(subs)), th)
    b, st = oracle.predict(ros, ros.flatten(Data(subs)), th)
    assert (st == 0).all() and rel_err_floor(b, a).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n_support,batch", [(70, False), (4, False), (0, True)])
def test_gpu_ros2_built_in_body_on_stiff_parameters(n_support, batch):
    rng = np.random.default_rng(73)
    subs = _oral_subjects(rng, 40)
    mo, ma = _models(1e-6)
    flat = mo.flatten(Data(subs))
    th = _stiff_theta(rng, len(subs) if batch else n_support)
    got, st = _gpu(mo, flat, th, batch=batch)
    assert runtime.last_kernel_name() == ("pmx_ode_ros2_pair" if (batch or n_support < 32) else "pmx_ode_ros2_grid")
    want, wst = (oracle.predict_batch if batch else oracle.predict)(mo, flat, th)
    np.testing.assert_array_equal(st, wst)
    # same method, same tolerances; FMA contraction may move a step boundary: agreement at the solver's tolerance
    assert rel_err_floor(got, want).max() < 2e-5
    exact, _ = (oracle.predict_batch if batch else oracle.predict)(ma, ma.flatten(Data(subs)), th)
    assert rel_err_floor(got, exact).max() < 5e-5


@pytest.mark.gpu
def test_gpu_ros2_custom_body_with_lag_covariate_and_likelihood():
    """A hiprtc-compiled body (time-varying covariate in the right-hand side: the f_t term), a lag time, fused likelihood."""
    from pharmsol_amd import AssayErrorModel, AssayErrorModels, ErrorPoly

    rng = np.random.default_rng(74)
    src = """
    PMX_DEVICE void pmx_dynamics(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                                 const double* derived, double* dx) {
      const double ke = p[1] * (cov[0] / 70.0);
      dx[0] = -p[0] * x[0];
      dx[1] = p[0] * x[0] - ke * x[1] + rateiv[0];
    }
    PMX_DEVICE void pmx_outputs(double t, const double* x, const double* p, const double* cov, const double* rateiv,
                                const double* derived, double* y) { y[0] = x[1] / p[2]; }
    """
    m = (ODE.custom(src, nstates=2, nparams=4, covariates=["wt"], lag={0: 3}, h_max=24.0).with_solver("ros2")
         .with_tolerances(1e-6, 1e-6))
    m.bolus_dest = {0: 0}
    oracle.compile_custom(src)
    subs = []
    for i in range(20):
        b = Subject.builder(f"c{i}").bolus(0.0, float(rng.uniform(100, 500)), 0).infusion(5.0, 200.0, 0, 1.5)
        for t in sorted(rng.uniform(0.1, 36, 6)):
            b = b.observation(float(t), float(rng.uniform(0.5, 5.0)), 0)
        subs.append(b.covariate("wt", 0.0, float(rng.uniform(50, 100))).covariate("wt", 24.0, float(rng.uniform(50, 100))).build())
    flat = m.flatten(Data(subs))
    th = np.stack([np.exp(rng.uniform(np.log(50.0), np.log(2000.0), 64)), rng.uniform(0.05, 0.4, 64), rng.uniform(10, 50, 64),
                   rng.uniform(0.0, 2.0, 64)], axis=1)
    got, st = _gpu(m, flat, th)
    assert runtime.last_kernel_name().startswith("pmx_jit_ode_ros2_grid")
    want, wst = oracle.predict(m, flat, th)
    np.testing.assert_array_equal(st, wst)
    assert rel_err_floor(got, want).max() < 5e-5
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.1, 0.1, 0.0, 0.0), 0.0))
    pop = runtime.DevicePopulation(flat, 0)
    ll, _ = runtime.loglik(m, pop, em, th)
    wll, _ = oracle.loglik(m, flat, em, th)
    np.testing.assert_allclose(ll.cpu().numpy(), wll, rtol=2e-4, atol=1e-6)

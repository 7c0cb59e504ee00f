"""Scenarios of the reference's integration tests that were not yet re-created one by one:

  * tests/numerical_stability.rs:48-94     analytical structure == hand-written ODE on three long schedules
                                           (infusion, oral absorption + iv + late load, two-compartment multi-dose),
                                           REL 1e-2 / ABS 1e-2 (:6-7)
  * tests/test_solvers.rs:68-103           the same one-compartment model under every solver, |diff| < 0.01 (:81)
  * ode/mod.rs:1459-1538, 1541-1700        103 very short, very large infusions into a six-state non-linear model
                                           (hybrid phage): the run completes, plasma prediction finite and positive
  * tests/support/bimodal_ke.rs:10-60,     the model every run-time back-end of the reference is checked on (JIT / AOT ==
    tests/bimodal_ke_entrypoint_matrix.rs  reference predictions at 1e-10): here the hiprtc-compiled body against the
                                           built-in one, the closed form and the oracle

The reference compares diffsol solvers with each other; here the ODE side is the library's RK4 / Dormand-Prince twin and
the analytical side its closed forms, so the reference's tolerances are kept AND the tighter ones this build promises
(analytical == oracle at 1e-6, RK4 == RK4 oracle at 1e-9, ODE == closed form at 1e-4).  CPU half: oracle against
itself / closed forms.  GPU half: the HIP path against the oracle."""
import math

import numpy as np
import pytest

import oracle
from pharmsol_amd import ODE, Data, Ratio, Subject, analytical, bolus, infusion, runtime

SIG = ("double t, const double* x, const double* p, const double* cov, const double* rateiv, "
       "const double* derived, double* ")
REL_TOL = ABS_TOL = 1e-2  # tests/numerical_stability.rs:6-7

OBS_13 = [0.0, 1.0, 2.0, 4.0, 8.0, 12.0, 24.0, 25.0, 26.0, 27.0, 28.0, 32.0, 36.0]
OBS_19 = OBS_13 + [48.0, 49.0, 50.0, 52.0, 56.0, 60.0]


def _scenarios():
    """(label, analytical model, its subject, ODE source, ODE shape, index-based ODE subject, theta)"""
    out = []
    # ---- infusion_vs_analytical_is_stable (:48-61, subject :139-151, models :153-214)
    an = analytical(name="infusion_reference", params=["ke", "v"], structure="one_compartment", states=["central"],
                    outputs=["cp"], routes=[bolus("load", "central"), infusion("iv", "central")], out={"cp": Ratio("central", "v")})
    sa = Subject.builder("infusion_reference").bolus(0.0, 100.0, "load").infusion(24.0, 150.0, "iv", 3.0)
    so = Subject.builder("infusion_reference").bolus(0.0, 100.0, 0).infusion(24.0, 150.0, 0, 3.0)
    for t in OBS_13:
        sa, so = sa.missing_observation(t, "cp"), so.missing_observation(t, 0)
    src = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{ dx[0] = -p[0] * x[0] + rateiv[0]; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0] / p[1]; }}
"""
    out.append(("infusion", an, sa.build(), src, dict(nstates=1, nparams=2, ndrugs=1), so.build(), [0.1, 1.0]))
    # ---- oral_absorption_tracks_reference (:63-77, subject :216-231, models :232-291): oral -> gut, load -> central
    with pytest.warns(UserWarning, match="doses go to state"):  # (the reference's own declaration order: see below)
        an = analytical(name="absorption_reference", params=["ka", "ke", "v"], structure="one_compartment_with_absorption",
                        states=["gut", "central"], outputs=["cp"],
                        routes=[bolus("load", "central"), bolus("oral", "gut"), infusion("iv", "central")],
                        out={"cp": Ratio("central", "v")})
    sa = (Subject.builder("absorption_reference").bolus(0.0, 100.0, "oral").infusion(24.0, 150.0, "iv", 3.0)
          .bolus(48.0, 100.0, "load"))
    # hand-written closures index by INPUT: `load` is bolus input 0, `oral` bolus input 1 (per-kind declaration order,
    # metadata.rs:926-946), and x.add_bolus(input) / b[input] put them into x[0] / x[1] (equation/mod.rs:313-328) on
    # both sides of the reference's comparison, whatever the routes' to_state says
    so = Subject.builder("absorption_reference").bolus(0.0, 100.0, 1).infusion(24.0, 150.0, 0, 3.0).bolus(48.0, 100.0, 0)
    for t in OBS_19:
        sa, so = sa.missing_observation(t, "cp"), so.missing_observation(t, 0)
    src = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{
  dx[0] = -p[0] * x[0];
  dx[1] = p[0] * x[0] - p[1] * x[1] + rateiv[0];
}}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[1] / p[2]; }}
"""
    out.append(("absorption", an, sa.build(), src, dict(nstates=2, nparams=3, ndrugs=2), so.build(), [1.0, 0.1, 1.0]))
    # ---- two_compartment_multi_dose_is_well_behaved (:79-93, subject :293-305, models :306-372)
    an = analytical(name="two_comp_reference", params=["ke", "kcp", "kpc", "v"], structure="two_compartments",
                    states=["central", "peripheral"], outputs=["cp"],
                    routes=[bolus("load", "central"), infusion("iv", "central")], out={"cp": Ratio("central", "v")})
    sa = Subject.builder("two_comp_reference").bolus(0.0, 100.0, "load").infusion(24.0, 150.0, "iv", 3.0)
    so = Subject.builder("two_comp_reference").bolus(0.0, 100.0, 0).infusion(24.0, 150.0, 0, 3.0)
    for t in OBS_13:
        sa, so = sa.missing_observation(t, "cp"), so.missing_observation(t, 0)
    src = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{
  dx[0] = rateiv[0] - p[0] * x[0] - p[1] * x[0] + p[2] * x[1];
  dx[1] = p[1] * x[0] - p[2] * x[1];
}}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0] / p[3]; }}
"""
    out.append(("two_compartment", an, sa.build(), src, dict(nstates=2, nparams=4, ndrugs=2), so.build(), [0.1, 3.0, 1.0, 1.0]))
    return out


def _agree(label, reference, candidate):
    """assert_models_agree (tests/numerical_stability.rs:96-137)"""
    assert len(reference) == len(candidate), label
    for i, (r, c) in enumerate(zip(reference, candidate)):
        abs_err = abs(r - c)
        assert abs_err <= ABS_TOL or abs_err / max(abs(r), ABS_TOL) <= REL_TOL, (label, i, r, c)


def _one_cmt_infusion_closed_form(ke, v, t):
    """bolus 100 at 0 + 150 over [24, 27]: the infusion scenario in closed form (the observation AT 0 sorts in front of
    the bolus, event.rs:292-304: it sees nothing yet)"""
    if t == 0.0:
        return 0.0
    x = 100.0 * math.exp(-ke * t)
    if t > 24.0:
        te = min(t, 27.0)
        x += (50.0 / ke) * (1.0 - math.exp(-ke * (te - 24.0))) * math.exp(-ke * (t - te))
    return x / v


SCENARIOS = _scenarios()


@pytest.mark.parametrize("sc", SCENARIOS, ids=[s[0] for s in SCENARIOS])
def test_oracle_analytical_matches_its_ode_twin(sc):
    label, an, sa, src, shape, so, theta = sc
    th = np.array([theta])
    want, st = oracle.predict(an, an.flatten(sa), th)
    assert st.max() == 0 and np.isfinite(want).all()
    oracle.compile_custom(src)
    for solver in ("rk4", "dopri5"):
        m = ODE.custom(src, h_max=0.02, **shape).with_solver(solver)
        if solver == "dopri5":
            m = m.with_tolerances(1e-8, 1e-8)
        got, st2 = oracle.predict(m, m.flatten(so), th)
        assert st2.max() == 0
        _agree(label + "/" + solver, want[:, 0], got[:, 0])
        assert (np.abs(got[:, 0] - want[:, 0]) / np.maximum(np.abs(want[:, 0]), 1e-6)).max() < 1e-4
    if label == "infusion":
        cf = np.array([_one_cmt_infusion_closed_form(theta[0], theta[1], t) for t in OBS_13])
        np.testing.assert_allclose(want[:, 0], cf, rtol=1e-12)


# ---- tests/test_solvers.rs: one model, every solver
SOLVER_SRC = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{ dx[0] = -p[0] * x[0] + rateiv[0]; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0] / p[1]; }}
"""
SOLVER_OBS = [0.5, 2.0, 8.0, 12.5, 14.0, 24.0]


def _solver_subject():
    b = Subject.builder("id1").bolus(0.0, 100.0, 0).infusion(12.0, 200.0, 0, 2.0)  # tests/test_solvers.rs:8-19
    for t in SOLVER_OBS:
        b = b.observation(t, 0.0, 0)
    return b.build()


def _solver_closed_form(ke, v):
    out = []
    for t in SOLVER_OBS:
        x = 100.0 * math.exp(-ke * t)
        if t > 12.0:
            te = min(t, 14.0)
            x += (100.0 / ke) * (1.0 - math.exp(-ke * (te - 12.0))) * math.exp(-ke * (t - te))
        out.append(x / v)
    return np.array(out)


def test_oracle_solver_selection_agrees():
    oracle.compile_custom(SOLVER_SRC)
    th = np.array([[0.1, 50.0]])
    cf = _solver_closed_form(0.1, 50.0)
    preds = {}
    for solver in ("rk4", "dopri5"):
        m = ODE.custom(SOLVER_SRC, nstates=1, nparams=2, h_max=0.02).with_solver(solver)  # default tolerances 1e-4 (ode/mod.rs:126-127)
        p, st = oracle.predict(m, m.flatten(_solver_subject()), th)
        assert st.max() == 0 and np.isfinite(p).all()
        preds[solver] = p[:, 0]
        assert np.abs(p[:, 0] - cf).max() < 0.01
    assert np.abs(preds["rk4"] - preds["dopri5"]).max() < 0.01  # tests/test_solvers.rs:81


# ---- ode/mod.rs:1459-1506: the hybrid phage model; :1541-1700 its 103-infusion schedule
PHAGE_SRC = f"""
PMX_DEVICE double soft(double v) {{ const double eps = 1.0e-12; return 0.5 * (v + sqrt(v * v + eps * eps)); }}
PMX_DEVICE void pmx_dynamics({SIG}dx) {{
  const double kep = 20.799022436141968, k12 = 3.611151695251465, k21 = 0.20569434165954592, kdep = 3.674600839614868;
  const double kcl_air = 98.17452669143677, kgr = 2.072104573249817, kinf = 1.909232258796692e-6, c50 = 427933.12072753906;
  const double klysis = 0.8622971177101135, burst = 1.591451644897461, ksp = 4.387639760971069, kdp = 0.0917521107196808;
  const double kn = 1.147785520553589, va = 12.829959392547607, bmax = 1.0e10;
  const double phage_air = soft(x[2]), bacc_pos = soft(x[3]), binf_pos = soft(x[4]), bprot_pos = soft(x[5]);
  const double tb = bacc_pos + binf_pos + bprot_pos;
  const double cair = phage_air / va;
  const double inf_eff = kinf * cair / (1.0 + cair / c50);
  dx[0] = -(kep + k12 + kdep) * x[0] + k21 * x[1] + rateiv[0];
  dx[1] = k12 * x[0] - k21 * x[1];
  dx[2] = kdep * x[0] - kcl_air * x[2] - inf_eff * bacc_pos + burst * klysis * binf_pos;
  dx[3] = kgr * bacc_pos * (1.0 - tb / bmax) - inf_eff * bacc_pos - ksp * x[3] + kdp * x[5] - kn * x[3];
  dx[4] = inf_eff * bacc_pos - klysis * x[4];
  dx[5] = ksp * x[3] - kdp * x[5];
}}
PMX_DEVICE void pmx_init({SIG}xi) {{ xi[3] = 3.0 * pow(10.0, 5.5); }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0]; }}
"""


def _phage_subject():
    """The schedule's shape (ode/mod.rs:1546-1690): 0.00125 h infusions of 1e9 / 3e9 every half hour to hour for a day and
    a half, observations clustered right behind some of them, duplicates included; 103 infusions in all."""
    b = Subject.builder("long_horizon_short_infusions").infusion(0.0, 1e9, 0, 0.00125)
    for t in (0.005, 0.01791667, 0.02208333, 0.02208333, 0.03458333, 0.03833333, 0.03833333, 0.08458333, 0.18625):
        b = b.missing_observation(t, 0)
    n = 1
    for k in range(1, 16):  # 0.5 .. 7.5
        b = b.infusion(0.5 * k, 1e9, 0, 0.00125)
        n += 1
    b = b.infusion(8.490833, 1e9, 0, 0.00125).missing_observation(8.991667, 0).infusion(8.992917, 1e9, 0, 0.00125)
    n += 2
    for t in (8.995833, 9.010417, 9.074167, 9.166667):
        b = b.missing_observation(t, 0)
    for t in (9.492917, 9.992917, 10.49292, 10.99292, 11.49292):
        b = b.infusion(t, 1e9, 0, 0.00125)
        n += 1
    b = b.infusion(12.01458, 3e9, 0, 0.00125).missing_observation(12.03958, 0).missing_observation(12.03958, 0)
    b = b.missing_observation(13.01375, 0).infusion(13.01542, 3e9, 0, 0.00125)
    n += 2
    for t in (13.01792, 13.1925, 13.1925, 13.1925, 13.26875):
        b = b.missing_observation(t, 0)
    t = 14.01542
    while n < 103:  # hourly 3e9 doses for the rest of the horizon
        b = b.infusion(t, 3e9, 0, 0.00125)
        t += 1.0
        n += 1
    b = b.missing_observation(t + 0.5, 0)
    return b.build(), n


def test_oracle_many_short_infusions_complete():
    oracle.compile_custom(PHAGE_SRC, has_init=True)
    sub, n = _phage_subject()
    assert n == 103
    # (the reference model has no parameters; the ABI wants at least one, unused)
    for solver, h in (("rk4", 0.005), ("dopri5", 0.5)):
        m = ODE.custom(PHAGE_SRC, nstates=6, nparams=1, has_init=True, h_max=h).with_solver(solver)
        if solver == "dopri5":
            m = m.with_tolerances(1e-6, 1e-6)
        p, st = oracle.predict(m, m.flatten(sub), np.array([[0.0]]))
        assert st.max() == 0
        assert np.isfinite(p).all() and p[0, 0] > 0.0  # ode/mod.rs:1694-1697
        if solver == "rk4":
            ref = p[:, 0]
        else:
            assert (np.abs(p[:, 0] - ref) / np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max())).max() < 1e-3


# ---- tests/support/bimodal_ke.rs: `dx(central) = -ke * central`, `out(cp) = central / v`, infusion(iv) -> central
BIMODAL_SRC = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{ dx[0] = -p[0] * x[0] + rateiv[0]; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0] / p[1]; }}
"""
BIMODAL_T = [0.5, 1.0, 2.0, 3.0, 4.0, 6.0, 8.0]  # OBSERVATION_TIMES (:11)
BIMODAL_THETA = [1.2, 50.0]                       # SUPPORT_POINT (:12)


def _bimodal():
    b = Subject.builder("bimodal_ke").infusion(0.0, 500.0, 0, 0.5)  # subject_for_indices (:50-56)
    for t in BIMODAL_T:
        b = b.missing_observation(t, 0)
    ke, v = BIMODAL_THETA
    cf = []
    for t in BIMODAL_T:
        te = min(t, 0.5)
        cf.append((1000.0 / ke) * (1.0 - math.exp(-ke * te)) * math.exp(-ke * (t - te)) / v)
    return b.build(), np.array(cf)


def test_oracle_run_time_compiled_body_matches_the_built_in_one():
    oracle.compile_custom(BIMODAL_SRC)
    sub, cf = _bimodal()
    th = np.array([BIMODAL_THETA])
    jit = ODE.custom(BIMODAL_SRC, nstates=1, nparams=2, h_max=0.01)
    builtin = ODE.new("one_cmt_iv", {0: Ratio(0, 1)}, nparams=2, h_max=0.01).with_nstates(1).with_ndrugs(1).with_nout(1)
    a, _ = oracle.predict(jit, jit.flatten(sub), th)
    b, _ = oracle.predict(builtin, builtin.flatten(sub), th)
    assert (np.abs(a - b) / np.abs(b)).max() < 1e-10  # tests/bimodal_ke_entrypoint_matrix.rs: 1e-10
    assert (np.abs(a[:, 0] - cf) / cf).max() < 1e-8


# --------------------------------------------------------------------------- GPU half
def _gpu(model, flat, theta):
    import torch

    pop = runtime.DevicePopulation(flat, 0)
    pred, st = runtime.predict(model, pop, np.ascontiguousarray(theta))
    torch.cuda.synchronize()
    return pred.cpu().numpy(), st.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("sc", SCENARIOS, ids=[s[0] for s in SCENARIOS])
def test_gpu_numerical_stability_scenarios(sc):
    label, an, sa, src, shape, so, theta = sc
    rng = np.random.default_rng(5)
    th = np.array([theta] + [list(np.array(theta) * rng.uniform(0.7, 1.4, len(theta))) for _ in range(39)])
    flat_a = an.flatten(Data([sa] * 3))
    got_a, st = _gpu(an, flat_a, th)
    want_a, wst = oracle.predict(an, flat_a, th)
    np.testing.assert_array_equal(st, wst)
    assert (np.abs(got_a - want_a) / np.maximum(np.abs(want_a), 1e-12)).max() < 1e-6
    oracle.compile_custom(src)
    for solver in ("rk4", "dopri5"):
        m = ODE.custom(src, h_max=0.02, **shape).with_solver(solver)
        if solver == "dopri5":
            m = m.with_tolerances(1e-8, 1e-8)
        flat_o = m.flatten(Data([so] * 3))
        got_o, st_o = _gpu(m, flat_o, th)
        assert st_o.max() == 0
        for k in range(th.shape[0]):
            _agree(f"{label}/{solver}/{k}", got_a[:, k], got_o[:, k])
        assert (np.abs(got_o - got_a) / np.maximum(np.abs(got_a), 1e-6)).max() < 1e-4
        if solver == "rk4":
            want_o, _ = oracle.predict(m, flat_o, th)
            assert (np.abs(got_o - want_o) / np.maximum(np.abs(want_o), 1e-9)).max() < 1e-9


@pytest.mark.gpu
def test_gpu_solver_selection_agrees():
    oracle.compile_custom(SOLVER_SRC)
    th = np.array([[0.1, 50.0]])
    cf = _solver_closed_form(0.1, 50.0)
    preds = {}
    for solver in ("rk4", "dopri5"):
        m = ODE.custom(SOLVER_SRC, nstates=1, nparams=2, h_max=0.02).with_solver(solver)
        p, st = _gpu(m, m.flatten(_solver_subject()), th)
        assert st.max() == 0 and np.isfinite(p).all()
        preds[solver] = p[:, 0]
        assert np.abs(p[:, 0] - cf).max() < 0.01
    assert np.abs(preds["rk4"] - preds["dopri5"]).max() < 0.01
    assert np.abs(preds["rk4"] - cf).max() / cf.max() < 1e-6


@pytest.mark.gpu
def test_gpu_many_short_infusions_complete():
    oracle.compile_custom(PHAGE_SRC, has_init=True)
    sub, _ = _phage_subject()
    m = ODE.custom(PHAGE_SRC, nstates=6, nparams=1, has_init=True, h_max=0.005)
    flat = m.flatten(Data([sub] * 2))
    got, st = _gpu(m, flat, np.array([[0.0]]))
    want, wst = oracle.predict(m, flat, np.array([[0.0]]))
    np.testing.assert_array_equal(st, wst)
    assert st.max() == 0 and np.isfinite(got).all() and got[0, 0] > 0.0
    assert (np.abs(got - want) / np.maximum(np.abs(want), 1e-6 * np.abs(want).max())).max() < 1e-8
    m2 = ODE.custom(PHAGE_SRC, nstates=6, nparams=1, has_init=True, h_max=0.5).with_solver("dopri5").with_tolerances(1e-6, 1e-6)
    got2, st2 = _gpu(m2, m2.flatten(Data([sub] * 2)), np.array([[0.0]]))
    assert st2.max() == 0 and np.isfinite(got2).all()
    assert (np.abs(got2 - got) / np.maximum(np.abs(got), 1e-3 * np.abs(got).max())).max() < 1e-3


@pytest.mark.gpu
def test_gpu_run_time_compiled_body_matches_the_built_in_one():
    oracle.compile_custom(BIMODAL_SRC)
    sub, cf = _bimodal()
    th = np.array([BIMODAL_THETA])
    jit = ODE.custom(BIMODAL_SRC, nstates=1, nparams=2, h_max=0.01)
    builtin = ODE.new("one_cmt_iv", {0: Ratio(0, 1)}, nparams=2, h_max=0.01).with_nstates(1).with_ndrugs(1).with_nout(1)
    a, sa_ = _gpu(jit, jit.flatten(sub), th)
    assert runtime.last_kernel_name().startswith("pmx_jit_ode_rk4")
    b, sb_ = _gpu(builtin, builtin.flatten(sub), th)
    assert sa_.max() == 0 and sb_.max() == 0
    assert (np.abs(a - b) / np.abs(b)).max() < 1e-10
    want, _ = oracle.predict(jit, jit.flatten(sub), th)
    assert (np.abs(a - want) / np.abs(want)).max() < 1e-10
    assert (np.abs(a[:, 0] - cf) / cf).max() < 1e-8

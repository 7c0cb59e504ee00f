"""The reference's full-feature fixtures, re-created (tests/full_feature_macro_parity.rs):

  * ODE         :10-53 macro form, :55-198 hand-written `ODE::new(diffeq, lag, fa, init, out)` whose diffeq takes the
                bolus vector, subject :200-218, support point :385-400 - lag = tlag sqrt(wt/70) (90/renal)^0.1,
                fa = clamp(f_oral (renal/90)^0.1, 0, 1), covariate-dependent init and volume, TWO bolus routes
                (oral -> depot with lag + fa, load -> central) and an infusion route sharing input 0 with `oral`.
  * Analytical  :220-257 macro form, :259-332 hand-written, subject :334-353, support point :440-452 - the same route
                layout on one_compartment_with_absorption with derived ke / adjusted_v.

The reference asserts macro == hand-written at 1e-10.  Here each fixture runs (a) through the CPU oracle in both forms,
(b) against an INDEPENDENT plain-Python march of the reference's rules (no oracle code, no shared closure source),
(c) on the device through GRID, PAIR and batch against the oracle (ODE <= 1e-4, analytical <= 1e-6; measured far
tighter), the fixture pair itself also against the independent march.
"""
import math

import numpy as np
import pytest

import oracle
from pharmsol_amd import (ODE, Analytical, AssayErrorModel, AssayErrorModels, Data, ErrorPoly, Ratio, Subject, _abi, analytical,
                          bolus, infusion, ode, runtime)

SIG = ("double t, const double* x, const double* p, const double* cov, const double* rateiv, "
       "const double* derived, double* ")
SIGB = ("double t, const double* x, const double* p, const double* cov, const double* rateiv, const double* bolus, "
        "const double* derived, double* ")

# ------------------------------------------------------------------------------------------------- ODE fixture
ODE_PARAMS = ["ka", "ke", "kcp", "kpc", "v", "tlag", "f_oral", "base_depot", "base_central", "base_peripheral"]
ODE_THETA = [1.1, 0.18, 0.07, 0.04, 35.0, 0.6, 0.85, 4.0, 18.0, 9.0]  # full_feature_macro_parity.rs:385-400

# macro form (:10-53): the bodies of the diffeq / lag / fa / init / out blocks; routes are injected by `ode(...)`
ODE_MACRO_SRC = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{
  const double wt = cov[COV_wt], renal = cov[COV_renal];
  const double wt_scale = pow(wt / 70.0, 0.75);
  const double renal_scale = pow(renal / 90.0, 0.25);
  const double adjusted_ke = p[P_ke] * wt_scale * renal_scale;
  const double adjusted_kcp = p[P_kcp] * pow(wt / 70.0, 0.25);
  dx[X_depot] = -p[P_ka] * x[X_depot];
  dx[X_central] = p[P_ka] * x[X_depot] - (adjusted_ke + adjusted_kcp) * x[X_central] + p[P_kpc] * x[X_peripheral];
  dx[X_peripheral] = adjusted_kcp * x[X_central] - p[P_kpc] * x[X_peripheral];
}}
PMX_DEVICE void pmx_route_lag({SIG}lag) {{
  const double lag_scale = sqrt(cov[COV_wt] / 70.0) * pow(90.0 / cov[COV_renal], 0.1);
  lag[R_oral] = p[P_tlag] * lag_scale;
}}
PMX_DEVICE void pmx_route_bioavailability({SIG}fa) {{
  const double fa_scale = pow(cov[COV_renal] / 90.0, 0.1);
  fa[R_oral] = fmin(fmax(p[P_f_oral] * fa_scale, 0.0), 1.0);
}}
PMX_DEVICE void pmx_init({SIG}xi) {{
  xi[X_depot] = p[P_base_depot] + 0.05 * cov[COV_wt];
  xi[X_central] = p[P_base_central] + 0.1 * cov[COV_renal];
  xi[X_peripheral] = p[P_base_peripheral] + 0.02 * cov[COV_wt];
}}
PMX_DEVICE void pmx_outputs({SIG}y) {{
  const double adjusted_v = p[P_v] * (cov[COV_wt] / 70.0) * (1.0 + 0.001 * (cov[COV_renal] - 90.0));
  y[Y_cp] = x[X_central] / adjusted_v;
}}
"""

# hand-written form (:55-198): index based, the diffeq adds bolus[] and rateiv[] itself
ODE_HAND_SRC = f"""
PMX_DEVICE void pmx_dynamics_bolus({SIGB}dx) {{
  const double wt = cov[0], renal = cov[1];
  const double wt_scale = pow(wt / 70.0, 0.75);
  const double renal_scale = pow(renal / 90.0, 0.25);
  const double adjusted_ke = p[1] * wt_scale * renal_scale;
  const double adjusted_kcp = p[2] * pow(wt / 70.0, 0.25);
  dx[0] = bolus[0] - p[0] * x[0];
  dx[1] = bolus[1] + p[0] * x[0] + rateiv[0] - (adjusted_ke + adjusted_kcp) * x[1] + p[3] * x[2];
  dx[2] = adjusted_kcp * x[1] - p[3] * x[2];
}}
PMX_DEVICE void pmx_route_lag({SIG}lag) {{ lag[0] = p[5] * (sqrt(cov[0] / 70.0) * pow(90.0 / cov[1], 0.1)); }}
PMX_DEVICE void pmx_route_bioavailability({SIG}fa) {{ fa[0] = fmin(fmax(p[6] * pow(cov[1] / 90.0, 0.1), 0.0), 1.0); }}
PMX_DEVICE void pmx_init({SIG}xi) {{ xi[0] = p[7] + 0.05 * cov[0]; xi[1] = p[8] + 0.1 * cov[1]; xi[2] = p[9] + 0.02 * cov[0]; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[1] / (p[4] * (cov[0] / 70.0) * (1.0 + 0.001 * (cov[1] - 90.0))); }}
"""


def ode_macro_model(h_max=0.02):
    return ode(name="ode_full_feature_parity", params=ODE_PARAMS, covariates=["wt", "renal"],
               states=["depot", "central", "peripheral"], outputs=["cp"],
               routes=[bolus("oral", "depot"), bolus("load", "central"), infusion("iv", "central")], source=ODE_MACRO_SRC,
               h_max=h_max)


def ode_hand_model(h_max=0.02):
    return ODE.user(ODE_HAND_SRC, nstates=3, nparams=10, ndrugs=2, nout=1, covariates=["wt", "renal"], h_max=h_max)


def ode_subject(i=0, scale=1.0, labels=True):
    """build_ode_subject(), :200-218 (i, scale: variations for populations; labels=False: dense indices for the
    hand-written form's metadata-free twin)"""
    oral, load, iv, cp = ("oral", "load", "iv", "cp") if labels else (0, 1, 0, 0)
    b = (Subject.builder(f"ode-full-features-{i}").bolus(0.0, 80.0 * scale, load).bolus(1.0, 120.0 * scale, oral)
         .infusion(6.0, 150.0 * scale, iv, 2.5))
    for t in (0.25, 0.75, 1.5, 3.0, 6.5, 7.0, 8.0, 12.0):
        b = b.missing_observation(t + 0.01 * i, cp)
    return (b.covariate("wt", 0.0, 68.0 + i).covariate("wt", 8.0, 74.0 + i).covariate("renal", 0.0, 95.0 - i)
            .covariate("renal", 8.0, 72.0).build())


def _lines():
    def line(v0, v1):  # CovariateSegment: slope * t + intercept on [0, 8), carried forward after the last knot
        slope = (v1 - v0) / 8.0
        icpt = v0 - slope * 0.0
        return lambda t: v1 if t >= 8.0 else slope * t + icpt

    return line(68.0, 74.0), line(95.0, 72.0)


def independent_ode_march(theta, h_max=0.02, rk=None):
    """The ODE fixture marched in plain Python from the reference's rules: lag at the recorded bolus time, fa at the
    shifted time (structs.rs:611-666), init at 0 (ode/mod.rs:536-549), a bolus as x[dest] += amount, the solver clock
    from the occasion's recorded initial time to every next event with stops at the infusion boundaries
    (ode/mod.rs:719-739), classic RK4 with n = ceil(dt / h_max) steps per piece unless `rk` integrates a piece itself."""
    ka, ke, kcp, kpc, v, tlag, f_oral, b_dep, b_cen, b_per = theta
    wt, renal = _lines()

    def f(t, x, rate):
        a_ke = ke * (wt(t) / 70.0) ** 0.75 * (renal(t) / 90.0) ** 0.25
        a_kcp = kcp * (wt(t) / 70.0) ** 0.25
        return [-ka * x[0], ka * x[0] + rate - (a_ke + a_kcp) * x[1] + kpc * x[2], a_kcp * x[1] - kpc * x[2]]

    def piece(x, t0, t1, rate):
        if rk is not None:
            return rk(f, x, t0, t1, rate)
        n = max(1, math.ceil((t1 - t0) / h_max))
        h = (t1 - t0) / n
        for s in range(n):
            t = t0 + s * h
            k1 = f(t, x, rate)
            k2 = f(t + 0.5 * h, [a + 0.5 * h * b for a, b in zip(x, k1)], rate)
            k3 = f(t + 0.5 * h, [a + 0.5 * h * b for a, b in zip(x, k2)], rate)
            k4 = f(t + h, [a + h * b for a, b in zip(x, k3)], rate)
            x = [a + (h / 6.0) * (p + 2.0 * q + 2.0 * r + s_) for a, p, q, r, s_ in zip(x, k1, k2, k3, k4)]
        return x

    tau = 1.0 + tlag * math.sqrt(wt(1.0) / 70.0) * (90.0 / renal(1.0)) ** 0.1
    fa = min(max(f_oral * (renal(tau) / 90.0) ** 0.1, 0.0), 1.0)
    # (time, rank: observation < bolus < infusion, payload)
    events = sorted([(t, 0, None) for t in (0.25, 0.75, 1.5, 3.0, 6.5, 7.0, 8.0, 12.0)] +
                    [(0.0, 1, (1, 80.0)), (tau, 1, (0, 120.0 * fa)), (6.0, 2, None)], key=lambda e: (e[0], e[1]))
    x = [b_dep + 0.05 * wt(0.0), b_cen + 0.1 * renal(0.0), b_per + 0.02 * wt(0.0)]
    bounds, clock, preds = [6.0, 8.5], 0.0, []
    for k, (t, kind, payload) in enumerate(events):
        if kind == 1:
            x[payload[0]] += payload[1]  # depot <- oral, central <- load
        elif kind == 0:
            preds.append(x[1] / (v * (wt(t) / 70.0) * (1.0 + 0.001 * (renal(t) - 90.0))))
        if k + 1 < len(events):
            nxt = events[k + 1][0]
            while nxt > clock:
                stop = min([b for b in bounds if clock < b <= nxt] + [nxt])
                rate = 60.0 if 6.0 <= clock < 8.5 else 0.0
                x = piece(x, clock, stop, rate)
                clock = stop
    return np.array(preds)


# ------------------------------------------------------------------------------------------------- analytical fixture
AN_PARAMS = ["ka", "ke0", "v", "tlag", "f_oral", "base_gut", "base_central"]
AN_THETA = [1.0, 0.16, 32.0, 0.5, 0.8, 3.0, 14.0]  # :440-452

AN_MACRO_SRC = f"""
PMX_DEVICE void pmx_derive({SIG}d) {{
  const double wt = cov[COV_wt], renal = cov[COV_renal];
  const double wt_scale = pow(wt / 70.0, 0.75);
  const double renal_scale = pow(renal / 90.0, 0.25);
  d[D_ke] = p[P_ke0] * wt_scale * renal_scale;
  d[D_adjusted_v] = p[P_v] * (wt / 70.0) * (1.0 + 0.001 * (renal - 90.0));
}}
PMX_DEVICE void pmx_route_lag({SIG}lag) {{
  lag[R_oral] = p[P_tlag] * (sqrt(cov[COV_wt] / 70.0) * pow(90.0 / cov[COV_renal], 0.1));
}}
PMX_DEVICE void pmx_route_bioavailability({SIG}fa) {{
  fa[R_oral] = fmin(fmax(p[P_f_oral] * pow(cov[COV_renal] / 90.0, 0.1), 0.0), 1.0);
}}
PMX_DEVICE void pmx_init({SIG}xi) {{
  xi[X_gut] = p[P_base_gut] + 0.03 * cov[COV_wt];
  xi[X_central] = p[P_base_central] + 0.08 * cov[COV_renal];
}}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[Y_cp] = x[X_central] / derived[D_adjusted_v]; }}
"""

# hand-written form (:259-332): its `eq` closure projects [ka, ke] and calls the structure; here the closed form of
# one_compartment_with_absorption (one_compartment_models.rs:32-44) is written out in the user's own propagator
AN_HAND_SRC = f"""
PMX_DEVICE void pmx_eq({SIG}xn) {{
  const double ka = p[0];
  const double ke = p[1] * pow(cov[0] / 70.0, 0.75) * pow(cov[1] / 90.0, 0.25);
  const double ea = exp(-ka * t), ee = exp(-ke * t);
  xn[0] = x[0] * ea;
  xn[1] = x[1] * ee + (rateiv[0] / ke) * (1.0 - ee) + ((ka * x[0]) / (ka - ke)) * (ee - ea);
}}
PMX_DEVICE void pmx_route_lag({SIG}lag) {{ lag[0] = p[3] * (sqrt(cov[0] / 70.0) * pow(90.0 / cov[1], 0.1)); }}
PMX_DEVICE void pmx_route_bioavailability({SIG}fa) {{ fa[0] = fmin(fmax(p[4] * pow(cov[1] / 90.0, 0.1), 0.0), 1.0); }}
PMX_DEVICE void pmx_init({SIG}xi) {{ xi[0] = p[5] + 0.03 * cov[0]; xi[1] = p[6] + 0.08 * cov[1]; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[1] / (p[2] * (cov[0] / 70.0) * (1.0 + 0.001 * (cov[1] - 90.0))); }}
"""


def an_macro_model():
    return analytical(name="analytical_full_feature_parity", params=AN_PARAMS, derived=["ke", "adjusted_v"],
                      covariates=["wt", "renal"], states=["gut", "central"], outputs=["cp"],
                      routes=[bolus("oral", "gut"), bolus("load", "central"), infusion("iv", "central")],
                      structure="one_compartment_with_absorption", source=AN_MACRO_SRC)


def an_hand_model():
    return Analytical.user(AN_HAND_SRC, eq=None, nstates=2, nparams=7, ndrugs=2, nout=1, covariates=["wt", "renal"])


def an_subject(i=0, scale=1.0, labels=True):
    """build_analytical_subject(), :334-353"""
    oral, load, iv, cp = ("oral", "load", "iv", "cp") if labels else (0, 1, 0, 0)
    b = (Subject.builder(f"analytical-full-features-{i}").bolus(0.0, 60.0 * scale, load).bolus(1.0, 100.0 * scale, oral)
         .infusion(6.0, 140.0 * scale, iv, 2.0))
    for t in (0.25, 0.75, 1.5, 3.0, 6.5, 7.0, 8.0, 12.0):
        b = b.missing_observation(t + 0.01 * i, cp)
    return (b.covariate("wt", 0.0, 68.0 + i).covariate("wt", 8.0, 74.0 + i).covariate("renal", 0.0, 95.0 - i)
            .covariate("renal", 8.0, 72.0).build())


def independent_analytical_march(theta):
    """The analytical fixture in plain Python: simulate_event per event (equation/mod.rs:300-358), solve(ti, tf) split at
    the infusion end (analytical/mod.rs:313-357), derive at the segment LENGTH for eq (expand/analytical.rs:254,286) and
    at the observation time for out; a bolus goes to x[input] (oral = 0 -> gut, load = 1 -> central)."""
    ka, ke0, v, tlag, f_oral, base_gut, base_central = theta
    wt, renal = _lines()
    tau = 1.0 + tlag * math.sqrt(wt(1.0) / 70.0) * (90.0 / renal(1.0)) ** 0.1
    fa = min(max(f_oral * (renal(tau) / 90.0) ** 0.1, 0.0), 1.0)
    events = sorted([(t, 0, None) for t in (0.25, 0.75, 1.5, 3.0, 6.5, 7.0, 8.0, 12.0)] +
                    [(0.0, 1, (1, 60.0)), (tau, 1, (0, 100.0 * fa)), (6.0, 2, None)], key=lambda e: (e[0], e[1]))
    x = [base_gut + 0.03 * wt(0.0), base_central + 0.08 * renal(0.0)]
    preds, inf_on = [], False
    for k, (t, kind, payload) in enumerate(events):
        if kind == 1:
            x[payload[0]] += payload[1]
        elif kind == 2:
            inf_on = True
        else:
            preds.append(x[1] / (v * (wt(t) / 70.0) * (1.0 + 0.001 * (renal(t) - 90.0))))
        if k + 1 < len(events):
            ti, tf = t, events[k + 1][0]
            if ti == tf:
                continue
            ts = sorted([ti, tf] + ([8.0] if (inf_on and ti < 8.0 < tf) else []))
            for a, b_ in zip(ts[:-1], ts[1:]):
                dt = b_ - a
                r = 70.0 if (inf_on and a >= 6.0 and b_ <= 8.0) else 0.0
                ke = ke0 * (wt(dt) / 70.0) ** 0.75 * (renal(dt) / 90.0) ** 0.25
                ea, ee = math.exp(-ka * dt), math.exp(-ke * dt)
                x = [x[0] * ea, x[1] * ee + (r / ke) * (1.0 - ee) + (ka * x[0] / (ka - ke)) * (ee - ea)]
    return np.array(preds)


# ------------------------------------------------------------------------------------------------- CPU
def test_route_layout_of_the_fixtures():
    # `assert_eq!(oral, iv); assert_eq!(load, 1)` (:375-377, :430-432): bolus and infusion routes are numbered apart
    for m in (ode_macro_model(), an_macro_model()):
        assert m.resolve_input_label("oral", "bolus") == m.resolve_input_label("iv", "infusion") == 0
        assert m.resolve_input_label("load", "bolus") == 1 and m.ndrugs == 2
        assert m.resolve_output_label("cp") == 0


def test_ode_fixture_compiles_for_gfx950_and_selects_the_general_walker():
    m = ode_macro_model()
    assert m.user_fns == (_abi.PMX_FN_DYNAMICS | _abi.PMX_FN_ROUTE_LAG | _abi.PMX_FN_ROUTE_BIOAVAILABILITY | _abi.PMX_FN_INIT |
                          _abi.PMX_FN_OUTPUTS)
    d = m.desc()
    assert (d.bolus_dest[0], d.bolus_dest[1], d.infusion_dest[0]) == (0, 1, 1)
    tu = runtime.jit_translation_unit(m)
    assert '#include "pmx_ode_user.hpp"' in tu and "HAS_LAG = true" in tu and "BOLUS_ARG = false" in tu
    assert "dx[1] += rateiv[0];" in tu  # the macro's route injection (expand/ode.rs:380-406)
    assert tu.count('extern "C" __global__') == 8  # GRID / PAIR x prediction / log-likelihood x RK4 / Dormand-Prince
    runtime.DeviceModel(m)  # hiprtc compiles for gfx950 without a device
    h = ode_hand_model()
    assert h.user_fns & _abi.PMX_FN_DYNAMICS_BOLUS
    assert "BOLUS_ARG = true" in runtime.jit_translation_unit(h)
    runtime.DeviceModel(h)


def test_ode_fixture_oracle_matches_the_independent_march():
    m = ode_macro_model()
    got, st = oracle.predict(m, m.flatten(ode_subject()), np.array([ODE_THETA]))
    assert st[0, 0] == 0
    np.testing.assert_allclose(got[:, 0], independent_ode_march(ODE_THETA), rtol=1e-12)
    assert got[0, 0] > 0.5  # the covariate-dependent initial state + the loading dose are in the first row


def test_ode_fixture_macro_form_equals_hand_written_form():
    # the reference's assertion (:402-413), through the oracle: route injection + amount-at-destination == the DiffEq
    # that adds bolus[] / rateiv[] itself with the jump f(x, bolus) - f(x, 0)
    m, h = ode_macro_model(), ode_hand_model()
    a, _ = oracle.predict(m, m.flatten(ode_subject()), np.array([ODE_THETA]))
    b, _ = oracle.predict(h, h.flatten(ode_subject(labels=False)), np.array([ODE_THETA]))
    assert np.abs(a - b).max() <= 1e-10


def test_ode_fixture_rk4_is_within_the_ode_budget_of_a_tight_integration():
    from scipy.integrate import solve_ivp

    def rk(f, x, t0, t1, rate):
        return list(solve_ivp(lambda t, y: f(t, y, rate), (t0, t1), x, method="DOP853", rtol=1e-12, atol=1e-12).y[:, -1])

    tight = independent_ode_march(ODE_THETA, rk=rk)
    rk4 = independent_ode_march(ODE_THETA)
    assert (np.abs(rk4 - tight) / np.abs(tight)).max() < 1e-7  # (north star: 1e-4 for ODE)


def test_analytical_fixture_oracle_matches_the_independent_march_in_both_forms():
    want = independent_analytical_march(AN_THETA)
    m, h = an_macro_model(), an_hand_model()
    a, st = oracle.predict(m, m.flatten(an_subject()), np.array([AN_THETA]))
    assert st[0, 0] == 0
    np.testing.assert_allclose(a[:, 0], want, rtol=1e-12)
    b, _ = oracle.predict(h, h.flatten(an_subject(labels=False)), np.array([AN_THETA]))
    assert np.abs(a - b).max() <= 1e-10  # the reference's assertion (:454-465)


def test_ode_solver_clock_starts_at_the_recorded_initial_time():
    """ode/mod.rs:348 builds the problem with t0 = occasion.initial_time() - the occasion as recorded - and :719-721
    only advances `while next_event_time > solver.state().t`: a lagged bolus that is the FIRST event of the re-sorted
    list is applied at t0 and decays from there; one that lands after another event is reached from the clock."""
    src = f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{ dx[0] = -p[0] * x[0] + rateiv[0]; }}
PMX_DEVICE void pmx_route_lag({SIG}lag) {{ lag[0] = p[1]; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0]; }}
"""
    m = ODE.user(src, nstates=1, nparams=2, ndrugs=1, nout=1, h_max=0.005)
    ke, lag = 0.3, 0.5
    s = Subject.builder("first").bolus(0.0, 100.0, 0).missing_observation(1.0, 0).missing_observation(2.0, 0).build()
    got, _ = oracle.predict(m, m.flatten(s), np.array([[ke, lag]]))
    np.testing.assert_allclose(got[:, 0], [100.0 * math.exp(-ke * 1.0), 100.0 * math.exp(-ke * 2.0)], rtol=1e-9)
    # an observation at the recorded dose time makes the bolus the SECOND event: now it acts from its landing time
    s2 = (Subject.builder("second").missing_observation(0.0, 0).bolus(0.0, 100.0, 0).missing_observation(1.0, 0)
          .missing_observation(2.0, 0).build())
    got2, _ = oracle.predict(m, m.flatten(s2), np.array([[ke, lag]]))
    np.testing.assert_allclose(got2[:, 0], [0.0, 100.0 * math.exp(-ke * 0.5), 100.0 * math.exp(-ke * 1.5)], rtol=1e-9, atol=1e-300)
    # the same two subjects through the theta-indexed lag of the built-in bodies
    b = ODE.new("one_cmt_iv", {0: Ratio(0)}, nparams=2, lag={0: 1}, h_max=0.005).with_nstates(1).with_ndrugs(1).with_nout(1)
    for subj, want in ((s, got), (s2, got2)):
        g, _ = oracle.predict(b, b.flatten(subj), np.array([[ke, lag]]))
        np.testing.assert_allclose(g, want, rtol=1e-12, atol=1e-300)


# ------------------------------------------------------------------------------------------------- GPU
def _gpu(model, flat, theta, batch=False):
    import torch

    pop = runtime.DevicePopulation(flat, 0)
    pred, st = runtime.predict(model, pop, np.ascontiguousarray(theta, dtype=np.float64), batch=batch)
    torch.cuda.synchronize()
    return pred.cpu().numpy(), st.cpu().numpy()


def _assert_parity(model, flat, theta, batch=False, kernel=None, tol=1e-6):
    got, st = _gpu(model, flat, theta, batch)
    if kernel:
        assert runtime.last_kernel_name() == kernel, runtime.last_kernel_name()
    want, wst = (oracle.predict_batch if batch else oracle.predict)(model, flat, theta)
    np.testing.assert_array_equal(st, wst)
    ok = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), ok)
    scale = np.maximum(np.abs(want), 1e-12 * np.abs(want[ok]).max() + 1e-300)
    rel = np.where(ok, np.abs(got - want) / scale, 0.0)
    err = rel.max()
    if not err <= tol:
        w = np.unravel_index(np.argmax(rel), rel.shape)
        off = flat.observation_offsets()
        subj = int(np.searchsorted(off, w[0], side="right") - 1)
        raise AssertionError(f"max rel err {err:.3e} at row {w[0]} (subject {subj}, its row {w[0] - off[subj]}), column {w[1:]}: "
                             f"got {got[w]!r} want {want[w]!r}; theta {theta[w[1] if len(w) > 1 else subj]!r}")
    return got, want


def _theta_around(center, n, rng, spread=0.3):
    th = np.array(center)[None, :] * np.exp(rng.uniform(-spread, spread, (n, len(center))))
    th[0] = center
    return th


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["macro", "hand"])
@pytest.mark.parametrize("shape", ["grid", "pair", "batch"])
def test_ode_full_feature_fixture_on_the_device(shape, form):
    """tests/full_feature_macro_parity.rs:10-218,355-413 through the HIP path: the exact fixture (subject, support
    point) as row / column 0 of a small population x support grid, every lane mapping; <= 1e-4 of the oracle is the
    north-star bound for ODE (asserted at 1e-9: same RK4 steps on both sides), the fixture pair itself also against the
    independent Python march."""
    rng = np.random.default_rng(23)
    m = ode_macro_model() if form == "macro" else ode_hand_model()
    subs = [ode_subject(i, 1.0 + 0.05 * i, labels=form == "macro") for i in range(19)]
    flat = m.flatten(Data(subs))
    kern = "pmx_jit_ode_user_rk4_grid" if shape == "grid" else "pmx_jit_ode_user_rk4_pair"
    if shape == "batch":
        th = _theta_around(ODE_THETA, len(subs), rng)
        got, _ = _assert_parity(m, flat, th, batch=True, kernel=kern, tol=1e-9)
        fixture = got[:8]
    else:
        th = _theta_around(ODE_THETA, 70 if shape == "grid" else 5, rng)
        th[3, 6] = 1.7  # f_oral (renal/90)^0.1 > 1: the clamp is hit
        got, _ = _assert_parity(m, flat, th, kernel=kern, tol=1e-9)
        fixture = got[:8, 0]
    np.testing.assert_allclose(fixture, independent_ode_march(ODE_THETA), rtol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["macro", "hand"])
@pytest.mark.parametrize("shape", ["grid", "pair", "batch"])
def test_analytical_full_feature_fixture_on_the_device(shape, form):
    """tests/full_feature_macro_parity.rs:220-353,415-465: two bolus routes + an infusion on
    one_compartment_with_absorption, derived ke / adjusted_v; GRID, PAIR and batch <= 1e-6 of the oracle."""
    rng = np.random.default_rng(29)
    m = an_macro_model() if form == "macro" else an_hand_model()
    subs = [an_subject(i, 1.0 + 0.05 * i, labels=form == "macro") for i in range(19)]
    flat = m.flatten(Data(subs))
    kern = "pmx_jit_analytical_grid" if shape == "grid" else "pmx_jit_analytical_pair"
    if shape == "batch":
        th = _theta_around(AN_THETA, len(subs), rng)
        got, _ = _assert_parity(m, flat, th, batch=True, kernel=kern)
        fixture = got[:8]
    else:
        th = _theta_around(AN_THETA, 70 if shape == "grid" else 5, rng)
        th[3, 4] = 1.6
        got, _ = _assert_parity(m, flat, th, kernel=kern)
        fixture = got[:8, 0]
    np.testing.assert_allclose(fixture, independent_analytical_march(AN_THETA), rtol=1e-6)


@pytest.mark.gpu
def test_ode_full_feature_loglik_and_adaptive_solver():
    import torch

    rng = np.random.default_rng(31)
    m = ode_macro_model()
    subs = [ode_subject(i) for i in range(9)]
    flat = m.flatten(Data(subs))
    th = _theta_around(ODE_THETA, 64, rng)
    _, want = _assert_parity(m, flat, th, tol=1e-9)
    # fused log-likelihood through the same walker, both lane mappings
    vals = np.abs(want[:, 0]) * np.exp(rng.normal(0, 0.2, want.shape[0])) + 0.05
    vals[::5] = np.nan
    flat.ev_value = flat.ev_value.copy()
    flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION] = vals
    em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
    for n in (64, 6):
        ll, st = runtime.loglik(m, runtime.DevicePopulation(flat, 0), em, np.ascontiguousarray(th[:n]))
        torch.cuda.synchronize()
        wll, wst = oracle.loglik(m, flat, em, th[:n])
        np.testing.assert_array_equal(st.cpu().numpy(), wst)
        assert (np.abs(ll.cpu().numpy() - wll) / np.maximum(np.abs(wll), 1.0)).max() < 1e-8
    # Dormand-Prince 5(4): GPU and oracle run the same rule set; FMA contraction can move an accept / reject decision
    ad = ode_macro_model().with_solver("dopri5").with_tolerances(1e-8, 1e-8)
    flat2 = ad.flatten(Data(subs))
    for n, kern in ((64, "pmx_jit_ode_user_dopri5_grid"), (6, "pmx_jit_ode_user_dopri5_pair")):
        got, _ = _assert_parity(ad, flat2, th[:n], kernel=kern, tol=1e-6)
        assert (np.abs(got[:, 0] - want[:, 0]) / np.abs(want[:, 0])).max() < 1e-6  # == the fixed-step solution


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PMX_FUZZ_ODE_USER_LAG", "8"))))  # (more seeds: set it)
def test_ode_random_lag_closures_that_reorder_doses(seed):
    """ODE twin of test_user_analytical.py::test_random_lag_closures_that_reorder_doses: lag values that change from
    dose to dose re-order the boluses among themselves and against the fixed events, negative lags move doses before the
    occasion's first event (applied at the solver clock, no integration backwards), two inputs share the list, an infusion
    adds boundaries; both forms of the dynamics; device == oracle."""
    rng = np.random.default_rng(9000 + seed)
    body = "dx[0] = {b0}-p[0] * x[0]; dx[1] = {b1}p[0] * x[0] - p[1] * x[1] {r};"
    lagfa = f"""
PMX_DEVICE void pmx_route_lag({SIG}lag) {{ lag[0] = p[3] * cov[0]; lag[1] = p[4]; }}
PMX_DEVICE void pmx_route_bioavailability({SIG}fa) {{ fa[0] = 0.5 + 0.4 * sin(t); fa[1] = p[5]; }}
PMX_DEVICE void pmx_init({SIG}xi) {{ xi[1] = p[2] * cov[0]; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[1] / (1.0 + 0.01 * cov[0]); }}
"""
    if seed % 2:
        src = f"PMX_DEVICE void pmx_dynamics_bolus({SIGB}dx) {{ " + body.format(b0="bolus[0] ", b1="bolus[1] + ", r="+ rateiv[0]") + " }\n" + lagfa
    else:
        src = f"PMX_DEVICE void pmx_dynamics({SIG}dx) {{ " + body.format(b0="", b1="", r="+ rateiv[0]") + " }\n" + lagfa
    m = ODE.user(src, nstates=2, nparams=6, ndrugs=2, nout=1, covariates=["c"], h_max=0.05)
    subs = []
    for i in range(int(rng.integers(4, 16))):
        b = Subject.builder(f"r{i}").covariate("c", 0.0, float(rng.uniform(2, 6))).covariate("c", 24.0, float(rng.uniform(-2, 0)))
        for _ in range(int(rng.integers(2, 8))):
            b = b.bolus(float(np.round(rng.uniform(0, 24), 1)), float(rng.uniform(20, 200)), int(rng.integers(0, 2)))
        if rng.random() < 0.6:
            b = b.infusion(float(np.round(rng.uniform(0, 12), 1)), 100.0, 0, float(np.round(rng.uniform(0.5, 4), 1)))
        for _ in range(int(rng.integers(3, 10))):
            b = b.missing_observation(float(np.round(rng.uniform(0, 36) * 2) / 2), 0)
        if rng.random() < 0.4:
            b = b.reset().covariate("c", 0.0, 1.0).bolus(3.0, 50.0, 0).missing_observation(1.0, 0).missing_observation(6.0, 0)
        subs.append(b.build())
    n = int(rng.choice([3, 64, 130]))
    th = np.concatenate([rng.uniform(0.3, 2.0, (n, 1)), rng.uniform(0.05, 0.5, (n, 1)), rng.uniform(0, 3, (n, 1)),
                         rng.uniform(0.5, 1.5, (n, 1)), np.round(rng.uniform(-1, 2, (n, 1)) * 2) / 2, rng.uniform(0.3, 1.0, (n, 1))],
                        axis=1)
    _assert_parity(m, m.flatten(Data(subs)), th, tol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["analytical", "ode"])
def test_more_than_64_lagged_boluses_in_one_occasion(backend):
    """A lag closure over an occasion with more boluses than a lane keeps sorted in registers / scratch (64): the landing
    order comes from repeated scans of the occasion's list instead (pmx_userlag.hpp).  Round 2 refused such populations."""
    rng = np.random.default_rng(77)
    lag = f"PMX_DEVICE void pmx_route_lag({SIG}lag) {{ lag[0] = p[2] * (1.0 + sin(3.0 * t)); }}\n"
    if backend == "analytical":
        m = Analytical.user(lag, eq="one_compartment_with_absorption", nstates=2, nparams=4, ndrugs=1, out={0: Ratio(1, 3)})
    else:
        src = lag + f"""
PMX_DEVICE void pmx_dynamics({SIG}dx) {{ dx[0] = -p[0] * x[0]; dx[1] = p[0] * x[0] - p[1] * x[1]; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[1] / p[3]; }}
"""
        m = ODE.user(src, nstates=2, nparams=4, ndrugs=1, nout=1, h_max=0.05)
    subs = []
    for i, nb in enumerate((90, 70, 5)):
        b = Subject.builder(f"many{i}")
        for k in range(nb):
            b = b.bolus(0.25 * k + 0.01 * i, float(rng.uniform(5, 20)), 0)
        for t in np.sort(rng.uniform(0.0, 30.0, 9)):
            b = b.missing_observation(float(np.round(t, 2)), 0)
        subs.append(b.build())
    for n in (40, 3):
        th = np.stack([rng.uniform(0.8, 2.0, n), rng.uniform(0.1, 0.3, n), rng.uniform(0.0, 1.5, n), rng.uniform(10, 40, n)], axis=1)
        _assert_parity(m, m.flatten(Data(subs)), th, tol=1e-6 if backend == "analytical" else 1e-9)

"""User closures for the Analytical back-end (`Analytical::new(eq, seq_eq, lag, fa, init, out)` with arbitrary functions
of (theta, t, covariates), src/simulator/mod.rs:41-197): the same source text is compiled by hiprtc for the device
(pmx_model_create_user -> pmx_analytical.hpp) and by g++ for the CPU oracle.

Re-created from the reference:
  * tests/analytical_macro_lowering.rs:225-260  the covariate model: lag = tlag sqrt(wt/70) (90/renal)^0.1,
    fa = clamp(f_oral (renal/90)^0.1, 0, 1), init = base + c cov, derived ke / adjusted_v, subject :35-51, support point :470-483
  * analytical/mod.rs:493-527                    seq_eq accumulates within one solve -> 2.5
  * analytical/mod.rs:530-560                    a propagator reading rateiv[3] -> 4.0
CPU half: the compile path (hiprtc needs no GPU), the oracle's user-closure walker against an independent Python
restatement of the fixture and against the oracle's descriptor path.  GPU half: every lane mapping against the oracle."""
import math

import numpy as np
import pytest

import oracle
from pharmsol_amd import (Analytical, AssayErrorModel, AssayErrorModels, Data, ErrorPoly, Lin, Pow, Ratio, Scaled, Subject,
                          _abi, analytical, bolus, infusion, runtime)

SIG = ("double t, const double* x, const double* p, const double* cov, const double* rateiv, "
       "const double* derived, double* ")

# the bodies of the macro's derive / lag / fa / init / out blocks (tests/analytical_macro_lowering.rs:236-258) live in
# pharmsol_amd.synth (bench.py --workload user runs the same model at scale)
from pharmsol_amd import synth  # noqa: E402

COVARIATE_SRC = synth.USER_COVARIATE_SRC
COVARIATE_THETA = [1.0, 0.16, 32.0, 0.5, 0.8, 3.0, 14.0]  # ka ke0 v tlag f_oral base_gut base_central (:470-483)
covariate_model = synth.model_user_covariates


def covariate_subject(i=0, scale=1.0):
    """covariate_subject(), tests/analytical_macro_lowering.rs:35-51 (i, scale: variations for populations)"""
    b = (Subject.builder(f"analytical-macro-covariates-{i}").bolus(1.0, 100.0 * scale, "oral")
         .infusion(6.0, 140.0 * scale, "iv", 2.0))
    for t in (0.25, 0.75, 1.5, 3.0, 6.5, 7.0, 8.0):
        b = b.missing_observation(t + 0.01 * i, "cp")
    return (b.covariate("wt", 0.0, 68.0 + i).covariate("wt", 8.0, 74.0 + i).covariate("renal", 0.0, 95.0 - i)
            .covariate("renal", 8.0, 72.0).build())


def independent_fixture_predictions(theta):
    """The fixture subject marched in plain Python from the reference's rules (no oracle code): covariate lines as
    slope * t + intercept (covariate.rs:50-65), lag at the recorded bolus time, fa at the shifted time, init at 0,
    derive at the segment LENGTH for eq (expand/analytical.rs:254,286) and at the observation time for out."""
    ka, ke0, v, tlag, f_oral, base_gut, base_central = theta

    def line(v0, v1):
        slope = (v1 - v0) / 8.0
        icpt = v0 - slope * 0.0
        return lambda t: v1 if t >= 8.0 else slope * t + icpt

    wt, renal = line(68.0, 74.0), line(95.0, 72.0)
    lag = tlag * math.sqrt(wt(1.0) / 70.0) * (90.0 / renal(1.0)) ** 0.1
    tau = 1.0 + lag
    fa = min(max(f_oral * (renal(tau) / 90.0) ** 0.1, 0.0), 1.0)
    events = sorted([(t, 0, None) for t in (0.25, 0.75, 1.5, 3.0, 6.5, 7.0, 8.0)] + [(tau, 1, 100.0 * fa), (6.0, 2, None)])
    x = [base_gut + 0.03 * wt(0.0), base_central + 0.08 * renal(0.0)]
    preds, inf_on = [], False
    for k, (t, kind, amt) in enumerate(events):
        if kind == 1:
            x[0] += amt
        elif kind == 2:
            inf_on = True
        else:
            adjusted_v = v * (wt(t) / 70.0) * (1.0 + 0.001 * (renal(t) - 90.0))
            preds.append(x[1] / adjusted_v)
        if k + 1 < len(events):
            ti, tf = t, events[k + 1][0]
            if ti == tf:
                continue
            ts = [ti, tf] + ([8.0] if (inf_on and ti < 8.0 < tf) else [])
            ts.sort()
            for a, b_ in zip(ts[:-1], ts[1:]):
                dt = b_ - a
                r = 70.0 if (inf_on and a >= 6.0 and b_ <= 8.0) else 0.0
                ke = ke0 * (wt(dt) / 70.0) ** 0.75 * (renal(dt) / 90.0) ** 0.25
                ea, ee = math.exp(-ka * dt), math.exp(-ke * dt)
                x = [x[0] * ea, x[1] * ee + (r / ke) * (1.0 - ee) + (ka * x[0] / (ka - ke)) * (ee - ea)]
    return np.array(preds)


# seq_eq / eq closures of the reference's two known-answer tests (analytical/mod.rs:494-501, 531-535)
SEQ_SRC = f"""
PMX_DEVICE void pmx_eq({SIG}xn) {{ xn[0] = x[0] + p[0] * t; }}           // next[0] += p[0] * dt
PMX_DEVICE void pmx_seq_eq({SIG}pw) {{ pw[0] += 1.0; }}                  // params[0] += 1.0
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0]; }}
"""
RATEIV3_SRC = f"""
PMX_DEVICE void pmx_eq({SIG}xn) {{ xn[0] = x[0] + rateiv[3] * t; }}      // next[0] += rateiv[3] * dt
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[0] = x[0]; }}
"""


def seq_model():
    return Analytical.user(SEQ_SRC, eq=None, nstates=1, nparams=1, ndrugs=1, nout=1)


def seq_subject():
    return Subject.builder("seq").bolus(0.0, 0.0, 0).infusion(0.25, 1.0, 0, 0.25).observation(1.0, 0.0, 0).build()


def rateiv3_model():
    return Analytical.user(RATEIV3_SRC, eq=None, nstates=4, nparams=1, ndrugs=4, nout=1)


def rateiv3_subject():
    return Subject.builder("inf").infusion(0.0, 4.0, 3, 1.0).observation(1.0, 0.0, 0).build()


# --------------------------------------------------------------------------- CPU: compile path + oracle
def test_function_mask_and_generated_policy():
    m = covariate_model()
    assert m.user_fns == (_abi.PMX_FN_DERIVE | _abi.PMX_FN_ROUTE_LAG | _abi.PMX_FN_ROUTE_BIOAVAILABILITY | _abi.PMX_FN_INIT |
                          _abi.PMX_FN_OUTPUTS)
    d = m.desc()
    assert d.n_derived == 2 and d.n_bind == 2  # [ka <- theta[0], ke <- derived[0]] (Mixed projection, analysis.rs:295-297)
    assert (d.bind[0].src, d.bind[0].index, d.bind[1].src, d.bind[1].index) == (_abi.PMX_SRC_PRIMARY, 0, _abi.PMX_SRC_DERIVED, 0)
    tu = runtime.jit_translation_unit(m)
    assert '#include "pmx_analytical.hpp"' in tu and "KID = 3" in tu and "HAS_LAG = true" in tu
    assert "kp[1] = der[0];" in tu and tu.count('extern "C" __global__') == 4  # GRID / PAIR x prediction / log-likelihood
    runtime.DeviceModel(m)  # hiprtc compiles for gfx950 without a device


def test_closures_left_out_fall_back_to_the_descriptor_forms():
    # only `derive` in the source: lag / fa / init / out come from the index forms of Analytical.new
    src = f"PMX_DEVICE void pmx_derive({SIG}d) {{ d[0] = p[1] * pow(cov[0] / 70.0, 0.75); }}"
    m = Analytical.user(src, eq="one_compartment_with_absorption", nstates=2, nparams=6, covariates=["wt"], n_derived=1,
                        bind=[("p", 0), ("d", 0)], out={0: Ratio(1, 2)}, init={1: 5}, lag={0: 3}, fa={0: 4})
    tu = runtime.jit_translation_unit(m)
    for piece in ("lag[0] = p[3];", "fa[0] = p[4];", "x[1] = p[5];", "y[0] = x[1] / p[2];", "kp[1] = der[0];"):
        assert piece in tu, piece
    runtime.DeviceModel(m)


def test_descriptor_rules_for_user_models():
    import ctypes as C

    from pharmsol_amd import _ffi

    L = _ffi.lib()
    m = seq_model()
    d = m.desc()
    assert d.kernel == _abi.PMX_K_CUSTOM
    h = C.c_void_p()
    assert L.pmx_model_create(C.byref(d), C.byref(h)) == _abi.PMX_ERR_INVALID_ARGUMENT  # needs pmx_model_create_user
    assert L.pmx_model_create_user(C.byref(d), m.source.encode(), _abi.PMX_FN_OUTPUTS, C.byref(h)) == _abi.PMX_ERR_INVALID_ARGUMENT
    assert b"PMX_FN_EQ" in L.pmx_last_error()
    bad = Analytical.user(SEQ_SRC.replace("p[0] * t", "q[0] * t"), eq=None, nstates=1, nparams=1)
    with pytest.raises(_abi.PmxError) as e:
        runtime.DeviceModel(bad)
    assert "undeclared identifier 'q'" in str(e.value) and "model:" in str(e.value)
    with pytest.raises(ValueError):
        Analytical.user(RATEIV3_SRC, eq="one_compartment", nstates=1, nparams=1)  # pmx_eq AND a structure


def test_oracle_user_walker_matches_an_independent_restatement_of_the_reference_fixture():
    m = covariate_model()
    flat = m.flatten(covariate_subject())
    got, st = oracle.predict(m, flat, np.array([COVARIATE_THETA]))
    want = independent_fixture_predictions(COVARIATE_THETA)
    assert st[0, 0] == 0
    np.testing.assert_allclose(got[:, 0], want, rtol=1e-12)
    # the pre-dose rows carry the covariate-dependent initial state: (14 + 0.08 * 95) decayed / adjusted_v
    assert got[0, 0] > 0.5


def test_oracle_known_answers_through_user_closures():
    # analytical/mod.rs:493-527 -> 2.5 and :530-560 -> 4.0, the closures as compiled source instead of oracle built-ins
    got, _ = oracle.predict(seq_model(), seq_model().flatten(seq_subject()), np.array([[1.0]]))
    assert abs(got[0, 0] - 2.5) < 1e-12
    got, _ = oracle.predict(rateiv3_model(), rateiv3_model().flatten(rateiv3_subject()), np.array([[0.0]]))
    assert got[0, 0] == 4.0


def _declarative_twin():
    """A model the descriptor path can express, and the same model as user closures."""
    decl = analytical(name="twin", params=["ka", "ke0", "v0", "tlag", "f"], derived={"ke": Scaled("ke0", (Pow("wt", 70.0, 0.75),)),
                                                                                   "v": Scaled("v0", (Lin("wt", 70.0, 0.004),))},
                      covariates=["wt"], states=["gut", "central"], outputs=["cp"], routes=[bolus("oral", "gut"), infusion("iv", "central")],
                      structure="one_compartment_with_absorption", out={"cp": Ratio("central", "v")}, fa={"oral": "f"})
    src = f"""
PMX_DEVICE void pmx_derive({SIG}d) {{
  d[D_ke] = p[P_ke0] * pow(cov[COV_wt] / 70.0, 0.75);
  d[D_v] = p[P_v0] * (1.0 + 0.004 * (cov[COV_wt] - 70.0));
}}
PMX_DEVICE void pmx_route_bioavailability({SIG}fa) {{ fa[R_oral] = p[P_f]; }}
PMX_DEVICE void pmx_outputs({SIG}y) {{ y[Y_cp] = x[X_central] / derived[D_v]; }}
"""
    user = analytical(name="twin", params=["ka", "ke0", "v0", "tlag", "f"], derived=["ke", "v"], covariates=["wt"],
                      states=["gut", "central"], outputs=["cp"], routes=[bolus("oral", "gut"), infusion("iv", "central")],
                      structure="one_compartment_with_absorption", source=src)
    return decl, user


def _twin_population(n, rng):
    subs = []
    for i in range(n):
        b = Subject.builder(f"t{i}").covariate("wt", 0.0, float(rng.uniform(50, 100))).covariate("wt", 20.0, float(rng.uniform(50, 100)))
        b = b.bolus(0.0, float(rng.uniform(50, 200)), "oral").bolus(12.0, 80.0, "oral")
        if i % 2:
            b = b.infusion(float(rng.uniform(1, 5)), 120.0, "iv", float(rng.uniform(0.5, 3)))
        for t in np.sort(rng.uniform(0.2, 30.0, 6)):
            b = b.missing_observation(float(t), "cp")
        if i % 3 == 0:
            b = b.reset().covariate("wt", 0.0, 77.0).bolus(0.0, 60.0, "oral").missing_observation(2.0, "cp")
        subs.append(b.build())
    return subs


def test_oracle_user_closures_equal_the_descriptor_path_on_an_expressible_model():
    decl, user = _declarative_twin()
    rng = np.random.default_rng(5)
    subs = _twin_population(12, rng)
    th = np.stack([rng.uniform(0.8, 2.0, 9), rng.uniform(0.05, 0.3, 9), rng.uniform(10, 50, 9), rng.uniform(0, 1, 9),
                   rng.uniform(0.4, 1.0, 9)], axis=1)
    a, _ = oracle.predict(decl, decl.flatten(Data(subs)), th)
    b, _ = oracle.predict(user, user.flatten(Data(subs)), th)
    np.testing.assert_allclose(b, a, rtol=1e-14)


# --------------------------------------------------------------------------- GPU parity
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
    scale = np.maximum(np.abs(want[ok]), 1e-12 * np.abs(want[ok]).max() + 1e-300)
    err = (np.abs(got[ok] - want[ok]) / scale).max()
    assert err <= tol, f"max rel err {err:.3e}"
    return got, want


def _theta_around(center, n, rng, spread=0.3):
    th = np.array(center)[None, :] * np.exp(rng.uniform(-spread, spread, (n, len(center))))
    th[0] = center
    return th


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["grid", "pair", "batch"])
def test_reference_covariate_fixture_on_the_device(shape):
    """tests/analytical_macro_lowering.rs:225-260 + :462-505 through the HIP path: the exact fixture (subject, support
    point) as row/column 0 of a small population x support grid, every lane mapping, <= 1e-6 of the oracle; the fixture
    pair itself also against the independent Python restatement."""
    rng = np.random.default_rng(17)
    m = covariate_model()
    subs = [covariate_subject(i, 1.0 + 0.05 * i) for i in range(21)]
    flat = m.flatten(Data(subs))
    if shape == "batch":
        th = _theta_around(COVARIATE_THETA, len(subs), rng)
        got, _ = _assert_parity(m, flat, th, batch=True, kernel="pmx_jit_analytical_pair")
        fixture = got[:7]
    else:
        th = _theta_around(COVARIATE_THETA, 70 if shape == "grid" else 5, rng)
        th[3, 4] = 1.6  # f_oral (renal/90)^0.1 > 1: the clamp is hit
        got, _ = _assert_parity(m, flat, th, kernel="pmx_jit_analytical_grid" if shape == "grid" else "pmx_jit_analytical_pair")
        fixture = got[:7, 0]
    np.testing.assert_allclose(fixture, independent_fixture_predictions(COVARIATE_THETA), rtol=1e-6)


@pytest.mark.gpu
def test_reference_covariate_fixture_absolute_covariate_time_and_loglik():
    import torch

    rng = np.random.default_rng(18)
    m = covariate_model(cov_time="segment_end_abs")  # the DSL runtime's rule (src/dsl/native.rs:1907-1916)
    subs = [covariate_subject(i) for i in range(9)]
    flat = m.flatten(Data(subs))
    th = _theta_around(COVARIATE_THETA, 64, rng)
    _, want = _assert_parity(m, flat, th, kernel="pmx_jit_analytical_grid")
    # fused log-likelihood through the same walker
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
        assert (np.abs(ll.cpu().numpy() - wll) / np.maximum(np.abs(wll), 1.0)).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n_support", [1, 64])
def test_seq_eq_and_custom_propagators_on_the_device(n_support):
    # analytical/mod.rs:493-527: solve(0, 0.25): p = 2, x = 0.5; solve(0.25, 1) splits at the infusion end 0.5 and its
    # parameter vector lives across both sub-segments: p = 2, x = 1.0; p = 3, x = 2.5
    m = seq_model()
    th = np.linspace(1.0, 3.0, n_support).reshape(-1, 1)
    got, _ = _assert_parity(m, m.flatten(seq_subject()), th, tol=1e-12)
    assert abs(got[0, 0] - 2.5) < 1e-12
    # a second solve starts from the support point again (parameters_v is rebuilt per solve, :331)
    s2 = (Subject.builder("seq2").bolus(0.0, 0.0, 0).infusion(0.25, 1.0, 0, 0.25).observation(1.0, 0.0, 0)
          .observation(2.0, 0.0, 0).build())
    got2, _ = _assert_parity(m, m.flatten(s2), th, tol=1e-12)
    assert abs(got2[1, 0] - (2.5 + 2.0)) < 1e-12
    # analytical/mod.rs:530-560: the propagator reads rateiv[3]
    m3 = rateiv3_model()
    got3, _ = _assert_parity(m3, m3.flatten(rateiv3_subject()), np.zeros((n_support, 1)), tol=0.0)
    assert got3[0, 0] == 4.0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PMX_FUZZ_USER_TWIN", "1"))))  # (more populations: set it)
@pytest.mark.parametrize("n_support", [72, 7])
def test_user_closures_equal_the_descriptor_kernels_on_an_expressible_model(n_support, seed):
    """The same model through two device paths: the library's own covariate kernels (host-evaluated factors) and the user
    walker (device-side covariate lookup), both against the oracle, and against each other at rounding level."""
    decl, user = _declarative_twin()
    rng = np.random.default_rng(6 + 1000 * seed)
    subs = _twin_population(40 if seed == 0 else int(rng.integers(3, 70)), rng)
    th = np.stack([rng.uniform(0.8, 2.0, n_support), rng.uniform(0.05, 0.3, n_support), rng.uniform(10, 50, n_support),
                   rng.uniform(0, 1, n_support), rng.uniform(0.4, 1.0, n_support)], axis=1)
    a, _ = _assert_parity(decl, decl.flatten(Data(subs)), th)
    b, _ = _assert_parity(user, user.flatten(Data(subs)), th)
    assert (np.abs(a - b) / np.maximum(np.abs(a), 1e-9)).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PMX_FUZZ_USER_LAG", "12"))))  # (more seeds: set the variable)
def test_random_lag_closures_that_reorder_doses(seed):
    """Lag closures whose value changes from dose to dose (a covariate falling steeply) re-order the boluses of one
    input among themselves and against the fixed events; negative lags move doses before the occasion's first event;
    several inputs share the occasion's list.  Device (per-lane sort of landing times) == oracle (re-sorted event list)."""
    rng = np.random.default_rng(4000 + seed)
    src = f"""
PMX_DEVICE void pmx_route_lag({SIG}lag) {{
  lag[0] = p[3] * cov[0];            // falls from ~+6 h to ~-2 h along the occasion: later doses overtake earlier ones
  lag[1] = p[4];
}}
PMX_DEVICE void pmx_route_bioavailability({SIG}fa) {{ fa[0] = 0.5 + 0.4 * sin(t); fa[1] = p[5]; }}
"""
    m = Analytical.user(src, eq="two_compartments", nstates=2, nparams=6, ndrugs=2, covariates=["c"], out={0: Ratio(0)}, init={1: 5})
    subs = []
    for i in range(int(rng.integers(4, 20))):
        b = Subject.builder(f"r{i}").covariate("c", 0.0, float(rng.uniform(2, 6))).covariate("c", 24.0, float(rng.uniform(-2, 0)))
        for _ in range(int(rng.integers(2, 9))):
            b = b.bolus(float(np.round(rng.uniform(0, 24), 1)), float(rng.uniform(20, 200)), int(rng.integers(0, 2)))
        if rng.random() < 0.6:
            b = b.infusion(float(np.round(rng.uniform(0, 12), 1)), 100.0, 0, float(np.round(rng.uniform(0.5, 4), 1)))
        for _ in range(int(rng.integers(3, 10))):
            b = b.missing_observation(float(np.round(rng.uniform(0, 36) * 2) / 2), 0)
        if rng.random() < 0.4:
            b = b.reset().covariate("c", 0.0, 1.0).bolus(3.0, 50.0, 0).missing_observation(1.0, 0).missing_observation(6.0, 0)
        subs.append(b.build())
    n = int(rng.choice([3, 64, 130]))
    from pharmsol_amd import synth

    th = np.concatenate([synth.theta_c3(n, synth.SplitMix64(seed + 1))[:, :3], rng.uniform(0.5, 1.5, (n, 1)),
                         np.round(rng.uniform(-1, 2, (n, 1)) * 2) / 2, rng.uniform(0.3, 1.0, (n, 1))], axis=1)
    _assert_parity(m, m.flatten(Data(subs)), th)


@pytest.mark.gpu
@pytest.mark.parametrize("n_support", [40, 5])
def test_user_lag_closure_with_observations_an_ulp_apart(n_support):
    """The user walker's twin of tests/test_gpu_parity.py::test_lagged_bolus_landing_between_observations_an_ulp_apart:
    a lag closure returning 0 leaves the bolus between two observations one ulp apart, where the solve has no propagation
    step (analytical/mod.rs:327); other lanes' lags put the second bolus exactly on an observation."""
    src = f"""
PMX_DEVICE void pmx_route_lag({SIG}lag) {{ lag[0] = p[3]; }}
"""
    m = Analytical.user(src, eq="two_compartments", nstates=2, nparams=4, ndrugs=1, out={0: Ratio(0)})
    t = 10.6
    t_next = float(np.nextafter(t, 20.0))
    s = (Subject.builder("u").missing_observation(10.3, 0).infusion(10.3, 286.0, 0, 0.3).missing_observation(t, 0)
         .bolus(t, 309.0, 0).missing_observation(t_next, 0).missing_observation(t_next, 0).missing_observation(14.3, 0)
         .bolus(13.3, 50.0, 0).missing_observation(13.8, 0).missing_observation(float(np.nextafter(13.8, 20.0)), 0).build())
    from pharmsol_amd import synth

    th = np.concatenate([synth.theta_c3(n_support)[:, :3], np.zeros((n_support, 1))], axis=1)
    th[1::4, 3] = -0.0
    th[2::4, 3] = 0.5
    th[3::4, 3] = 1.25
    got, want = _assert_parity(m, m.flatten(s), th)
    zero = th[:, 3] == 0.0
    assert (want[2, zero] > want[1, zero] + 1.0).all()

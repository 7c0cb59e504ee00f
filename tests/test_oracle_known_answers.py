"""Pin the CPU oracle to every known answer the reference's own tests hold for this path
(SURVEY.md §8c).  The reference has no golden prediction vectors; these are its closed-form /
exact expectations, with the reference's tolerances (or tighter)."""
import math

import numpy as np
import pytest

import oracle
from pharmsol_amd import ODE, Analytical, Parameters, Ratio, Subject, _abi
from tests import models


def _desc_test_kernel(kid: int, nstates: int, ndrugs: int, nparams: int):
    """An oracle-only analytical model (ids >= 100) with out y = x[0]."""
    m = Analytical.new("one_compartment", {0: Ratio(0, None)}, nparams=nparams).with_nstates(nstates).with_ndrugs(
        ndrugs).with_nout(1)
    d = m.desc()
    d.kernel = kid
    return m, d


def test_secondary_equations_accumulate_within_single_solve():
    # analytical/mod.rs:493-527: seq_eq mutates ONE parameter copy cumulatively across the
    # sub-segments of a single solve (bolus@0, infusion 0.25..0.5, obs@1 -> 3 sub-segments) => 2.5
    m, d = _desc_test_kernel(oracle.K_TEST_SEQ_ACCUM, 1, 1, 1)
    s = Subject.builder("seq").bolus(0.0, 0.0, 0).infusion(0.25, 1.0, 0, 0.25).observation(1.0, 0.0, 0).build()
    pred, _ = oracle.predict(d, m.flatten(s), np.array([[1.0]]))
    assert abs(pred[0, 0] - 2.5) < 1e-12


def test_infusion_inputs_match_state_dimension():
    # analytical/mod.rs:530-560: rateiv[3] is filled for an infusion on input 3 => exactly 4.0
    m, d = _desc_test_kernel(oracle.K_TEST_RATEIV3, 4, 4, 1)
    s = Subject.builder("inf").infusion(0.0, 4.0, 3, 1.0).observation(1.0, 0.0, 0).build()
    pred, _ = oracle.predict(d, m.flatten(s), np.array([[0.0]]))
    assert pred[0, 0] == 4.0


PM_CASES = [  # analytical/mod.rs:805-889 (x, p, rateiv) at t = 1.5
    ("one_compartment", [100.0], [0.2], [5.0]),
    ("one_compartment_cl", [100.0], [0.2, 2.0], [5.0]),
    ("one_compartment_with_absorption", [10.0, 20.0], [1.1, 0.2], [5.0]),
    ("one_compartment_cl_with_absorption", [10.0, 20.0], [1.1, 0.2, 2.0], [5.0]),
    ("two_compartments", [100.0, 40.0], [0.1, 0.3, 0.2], [3.0]),
    ("two_compartments_cl", [100.0, 40.0], [0.1, 0.3, 1.0, 2.0], [3.0]),
    ("two_compartments_with_absorption", [10.0, 100.0, 40.0], [0.1, 1.0, 0.3, 0.2], [3.0]),
    ("two_compartments_cl_with_absorption", [10.0, 100.0, 40.0], [1.0, 0.1, 0.3, 1.0, 2.0], [3.0]),
    ("three_compartments", [100.0, 40.0, 20.0], [0.1, 3.0, 2.0, 1.0, 0.5], [2.0]),
    ("three_compartments_cl", [100.0, 40.0, 20.0], [0.1, 3.0, 2.0, 1.0, 3.0, 4.0], [2.0]),
    ("three_compartments_with_absorption", [10.0, 100.0, 40.0, 20.0], [1.0, 0.1, 3.0, 2.0, 1.0, 0.5], [2.0]),
    ("three_compartments_cl_with_absorption", [10.0, 100.0, 40.0, 20.0], [1.0, 0.1, 3.0, 2.0, 1.0, 3.0, 4.0], [2.0]),
]


@pytest.mark.parametrize("name,x,p,r", PM_CASES)
def test_pmetrics_wrappers_match_native_helpers(name, x, p, r):
    native = oracle.kernel(name, x, p, 1.5, r)
    wrapped = oracle.kernel(name, [1234.0] + x, p, 1.5, [5678.0] + r, pm=True)
    assert wrapped[0] == 0.0 and len(wrapped) == len(native) + 1
    np.testing.assert_allclose(wrapped[1:], native, rtol=1e-10, atol=1e-10)
    assert np.isfinite(native).all()


@pytest.mark.parametrize("name,x,p,r", PM_CASES)
def test_cl_kernels_equal_their_micro_constant_form(name, x, p, r):
    # *_cl_models.rs: convert then delegate — the CL kernel must equal the native kernel on converted params
    if "_cl" not in name:
        pytest.skip("native form")
    base = name.replace("_cl", "")
    p = list(p)
    if base == "one_compartment":
        q = [p[0] / p[1]]
    elif base == "one_compartment_with_absorption":
        q = [p[0], p[1] / p[2]]
    elif base == "two_compartments":
        q = [p[0] / p[2], p[1] / p[2], p[1] / p[3]]
    elif base == "two_compartments_with_absorption":
        q = [p[1] / p[3], p[0], p[2] / p[3], p[2] / p[4]]
    elif base == "three_compartments":
        q = [p[0] / p[3], p[1] / p[3], p[2] / p[3], p[1] / p[4], p[2] / p[5]]
    else:
        q = [p[0], p[1] / p[4], p[2] / p[4], p[3] / p[4], p[2] / p[5], p[3] / p[6]]
    np.testing.assert_array_equal(oracle.kernel(name, x, p, 1.5, r), oracle.kernel(base, x, q, 1.5, r))


def test_c1_analytical_readme_values():
    # examples/analytical_readme.rs; expected values from SURVEY.md §8c (hand-restated closed form)
    m = models.readme_analytical()
    th = Parameters.with_model(m, [("ka", 1.2), ("ke0", 0.08), ("v", 194.0)])
    pred, st = oracle.predict(m, m.flatten(models.readme_subject()), th.as_slice())
    want = [1.1363216631314599, 1.7130756583758835, 2.0906323551896495, 1.956103669112038]
    np.testing.assert_allclose(pred[:, 0], want, rtol=4e-16)
    # and the single closed form from t = 0 (one_compartment_models.rs:32-44 with x = [500, 0])
    ke = 0.08 * (75.0 / 70.0) ** 0.75
    ka = 1.2
    for t, p in zip([0.5, 1.0, 2.0, 4.0], pred[:, 0]):
        exact = (ka * 500.0 / (ka - ke)) * (math.exp(-ke * t) - math.exp(-ka * t)) / 194.0
        assert abs(p - exact) / exact < 1e-14


def _run_infusions(subject, ke=0.0, h_max=0.01):
    # ode/mod.rs:981-1011: dx = rateiv[0] (- ke x), y = x[0]
    m = ODE.new("one_cmt_iv", {0: Ratio(0, None)}, nparams=1, h_max=h_max).with_nstates(1).with_ndrugs(1).with_nout(1)
    m.has_metadata = True
    m.routes = [__import__("pharmsol_amd").infusion("iv", 0)]
    m.states = ["central"]
    m.outputs = ["cp"]
    pred, _ = oracle.predict(m, m.flatten(subject), np.array([[ke]]))
    return pred[:, 0]


def test_ode_short_infusion_dose_conserved():  # ode/mod.rs:1274-1284
    s = Subject.builder("short").infusion(0.0, 100.0, "iv", 0.1).observation(0.5, 0.0, "cp").build()
    assert abs(_run_infusions(s)[0] - 100.0) / 100.0 < 1e-12


def test_ode_observation_at_infusion_end_uses_active_left_rate():  # ode/mod.rs:1286-1298
    s = (Subject.builder("end").infusion(0.0, 100.0, "iv", 0.1).observation(0.1, 0.0, "cp").observation(0.5, 0.0, "cp")
         .build())
    p = _run_infusions(s)
    assert abs(p[0] - 100.0) / 100.0 < 1e-12 and abs(p[1] - 100.0) / 100.0 < 1e-12


def test_ode_very_short_infusion():  # ode/mod.rs:1300-1321
    s = Subject.builder("vshort").infusion(0.0, 100.0, "iv", 0.01).observation(0.01, 0.0, "cp").build()
    assert abs(_run_infusions(s)[0] - 100.0) / 100.0 < 1e-12


def test_ode_delayed_short_infusion():  # ode/mod.rs:1323-1334
    s = (Subject.builder("delayed").observation(0.0, 0.0, "cp").infusion(0.5, 100.0, "iv", 0.01)
         .observation(0.52, 0.0, "cp").build())
    assert abs(_run_infusions(s)[1] - 100.0) / 100.0 < 1e-12


def test_ode_back_to_back_infusions_conserve_dose():  # ode/mod.rs:1336-1347
    s = (Subject.builder("b2b").infusion(0.0, 100.0, "iv", 0.5).infusion(0.5, 100.0, "iv", 0.5)
         .observation(1.0, 0.0, "cp").build())
    assert abs(_run_infusions(s)[0] - 200.0) / 200.0 < 1e-12


def test_ode_infusion_end_one_ulp_after_observation():  # ode/mod.rs:1349-1395, closed form :1389-1394
    dur = np.nextafter(10.0, np.inf) - 5.0
    s = (Subject.builder("ulp").infusion(5.0, 100.0, "iv", dur).observation(10.0, 0.0, "cp")
         .observation(20.0, 0.0, "cp").build())
    p = _run_infusions(s, ke=0.5, h_max=0.005)
    delivered = 100.0 * (1.0 - math.exp(-2.5)) / 2.5
    expected = delivered * math.exp(-5.0)
    assert abs(p[1] - expected) / expected < 1e-8  # reference tolerance: 1e-3


def test_ode_observations_a_few_ulps_from_a_bolus():  # ode/mod.rs:1398-1450
    ulp = np.nextafter(12.0, np.inf) - 12.0
    s = (Subject.builder("dense").bolus(0.0, 200.0, 0).bolus(12.0, 100.0, 0).missing_observation(0.0, 0)
         .missing_observation(12.0 - 16.0 * ulp, 0).missing_observation(12.0 + 16.0 * ulp, 0)
         .missing_observation(24.0, 0).build())
    m = ODE.new("one_cmt_iv", {0: Ratio(0, None)}, nparams=1, h_max=0.01).with_nstates(1).with_ndrugs(1).with_nout(1)
    pred, _ = oracle.predict(m, m.flatten(s), np.array([[0.3]]))
    p = pred[:, 0]
    assert len(p) == 4
    # observation at t=0 sorts BEFORE the bolus at t=0 (event.rs:292-304): pre-dose state
    assert p[0] == 0.0
    a = 200.0 * math.exp(-0.3 * 12.0)
    assert abs(p[1] - a) / a < 1e-9 and abs(p[2] - (a + 100.0)) / (a + 100.0) < 1e-9
    b = (a + 100.0) * math.exp(-0.3 * 12.0)
    assert abs(p[3] - b) / b < 1e-9


@pytest.mark.parametrize("structure,central,theta,subject_fn,diffeq", [c for c in models.KERNEL_CASES if c[4]])
def test_analytical_kernel_matches_its_ode(structure, central, theta, subject_fn, diffeq):
    # the reference's kernel unit tests (e.g. two_compartment_models.rs:125-180): analytical ~ hand-written ODE,
    # there at max_relative 1e-4 / epsilon 1.0 against BDF; here against converged RK4 at 1e-7.
    subj = subject_fn()
    ma = models.handwritten_analytical(structure, central, len(theta))
    mo = models.handwritten_ode(diffeq, central, len(theta), h_max=0.005)
    th = np.array([theta])
    pa, _ = oracle.predict(ma, ma.flatten(subj), th)
    po, _ = oracle.predict(mo, mo.flatten(subj), th)
    scale = np.max(np.abs(pa))
    assert np.max(np.abs(pa - po)) / scale < 1e-7


def test_observation_at_dose_time_sees_pre_dose_state():
    # Appendix B rule 1; tests/ode_optimizations.rs:786-843
    m = models.handwritten_analytical("one_compartment", 0, 2).with_ndrugs(1)
    s = Subject.builder("tie").bolus(1.0, 100.0, 0).missing_observation(1.0, 0).missing_observation(2.0, 0).build()
    pred, _ = oracle.predict(m, m.flatten(s), np.array([[0.2, 10.0]]))
    assert pred[0, 0] == 0.0
    assert abs(pred[1, 0] - 100.0 * math.exp(-0.2) / 10.0) < 1e-13


def test_overlapping_infusions_sum_rates():
    # Appendix B rule 5; tests/ode_optimizations.rs:523
    m = models.handwritten_analytical("one_compartment", 0, 2).with_ndrugs(1)
    s = (Subject.builder("ovl").infusion(0.0, 100.0, 0, 2.0).infusion(1.0, 60.0, 0, 2.0).missing_observation(4.0, 0)
         .build())
    ke, v = 0.3, 5.0
    pred, _ = oracle.predict(m, m.flatten(s), np.array([[ke, v]]))

    def seg(x, r, dt):
        e = math.exp(-ke * dt)
        return x * e + r / ke * (1 - e)

    x = seg(0.0, 50.0, 1.0)
    x = seg(x, 80.0, 1.0)
    x = seg(x, 30.0, 1.0)
    x = seg(x, 0.0, 1.0)
    assert abs(pred[0, 0] - x / v) / (x / v) < 1e-14


def test_state_resets_every_occasion_and_init_only_first():
    # Appendix B rule 2 (analytical/mod.rs:409-426)
    m = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=3, init={0: 2}).with_nstates(1).with_ndrugs(
        1).with_nout(1)
    s = (Subject.builder("occ").missing_observation(0.0, 0).missing_observation(1.0, 0).reset()
         .missing_observation(0.0, 0).bolus(0.0, 10.0, 0).missing_observation(1.0, 0).build())
    ke, v, x0 = 0.5, 2.0, 40.0
    pred, _ = oracle.predict(m, m.flatten(s), np.array([[ke, v, x0]]))
    want = [x0 / v, x0 * math.exp(-ke) / v, 0.0, 10.0 * math.exp(-ke) / v]
    np.testing.assert_allclose(pred[:, 0], want, rtol=1e-14, atol=0)


def test_lag_then_bioavailability_rewrite():
    # Appendix B rule 3 (structs.rs:611-666): lag first (re-sort), then fa; bolus only
    m = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=4, lag={0: 2}, fa={0: 3}).with_nstates(
        1).with_ndrugs(1).with_nout(1)
    s = (Subject.builder("lag").bolus(0.0, 100.0, 0).missing_observation(0.5, 0).missing_observation(2.0, 0).build())
    ke, v, tlag, f = 0.4, 4.0, 1.0, 0.6
    pred, _ = oracle.predict(m, m.flatten(s), np.array([[ke, v, tlag, f]]))
    assert pred[0, 0] == 0.0  # dose has not arrived at t = 0.5
    want = 100.0 * f * math.exp(-ke * (2.0 - tlag)) / v
    assert abs(pred[1, 0] - want) / want < 1e-14


def test_complex_roots_flagged_not_trapped():
    # two_compartment_models.rs:20-22 panics; here the pair is flagged and its rows are NaN.
    # disc = (ke+kcp+kpc)^2 - 4 ke kpc < 0 needs a negative rate constant.
    m = models.handwritten_analytical("two_compartments", 0, 4).with_ndrugs(1)
    s = Subject.builder("cx").bolus(0.0, 10.0, 0).missing_observation(0.0, 0).missing_observation(1.0, 0).build()
    th = np.array([[1.0, -1.9, 1.0, 1.0], [0.1, 0.3, 0.2, 50.0]])
    pred, st = oracle.predict(m, m.flatten(s), th)
    assert st[0, 0] == _abi.PMX_PAIR_COMPLEX_ROOTS and st[0, 1] == _abi.PMX_PAIR_OK
    assert pred[0, 0] == 0.0 and np.isnan(pred[1, 0]) and np.isfinite(pred[:, 1]).all()


def test_input_out_of_range_is_an_error():
    # equation/mod.rs:322-327
    m = models.handwritten_analytical("one_compartment", 0, 2).with_ndrugs(1)
    s = Subject.builder("oor").bolus(0.0, 1.0, 1).missing_observation(1.0, 0).build()
    with pytest.raises(_abi.PmxError) as e:
        oracle.predict(m, m.flatten(s), np.array([[0.1, 1.0]]))
    assert e.value.status == _abi.PMX_ERR_INPUT_OUT_OF_RANGE


COV_KNOTS = ([0.0, 10.0, 20.0], [70.0, 80.0, 60.0])


@pytest.mark.parametrize("t,want", [(-5.0, 70.0), (0.0, 70.0), (5.0, 75.0), (10.0, 80.0), (15.0, 70.0), (20.0, 60.0),
                                    (100.0, 60.0)])
def test_covariate_interpolation(t, want):
    # covariate.rs:216-241: linear inside, first value before, last value at/after the last knot
    assert abs(oracle.cov_interpolate(*COV_KNOTS, t) - want) < 1e-12


def test_covariate_carry_forward_when_fixed():
    assert oracle.cov_interpolate(*COV_KNOTS, 5.0, fixed=True) == 70.0
    assert oracle.cov_interpolate(*COV_KNOTS, 10.0, fixed=True) == 80.0


def test_covariate_value_is_slope_times_t_plus_intercept():
    # covariate.rs:198-208 stores slope/intercept (NOT the lerp form): reproduce the exact rounding
    kt, kv = [0.3, 7.7], [68.123, 74.9]
    slope = (kv[1] - kv[0]) / (kt[1] - kt[0])
    icpt = kv[0] - slope * kt[0]
    for t in (0.3, 1.234567, 5.5, 7.699999):
        assert oracle.cov_interpolate(kt, kv, t) == slope * t + icpt


def test_macro_derive_sees_covariates_at_segment_length():
    # SURVEY §3.1 / tests/analytical_macro_lowering.rs:264-273: `derive` inside eq gets t = dt, not absolute time
    from pharmsol_amd import Pow, Scaled, analytical, bolus

    def make(mode):
        return analytical(name="m", params=["ke0", "v"], derived={"ke": Scaled("ke0", (Pow("wt", 70.0, 0.75),))},
                          covariates=["wt"], states=["central"], outputs=["cp"], routes=[bolus("iv", "central")],
                          structure="one_compartment", out={"cp": Ratio("central", "v")}, cov_time=mode)

    s = (Subject.builder("cov").bolus(0.0, 100.0, "iv").missing_observation(4.0, "cp").missing_observation(6.0, "cp")
         .covariate("wt", 0.0, 60.0).covariate("wt", 10.0, 90.0).build())
    wt = lambda t: 60.0 + 3.0 * t
    ke0, v = 0.2, 10.0
    # SEGMENT_DT: segments [0,4] (dt=4 -> wt(4)) and [4,6] (dt=2 -> wt(2)!)
    x = 100.0 * math.exp(-ke0 * (wt(4.0) / 70.0) ** 0.75 * 4.0)
    p0 = x / v
    x = x * math.exp(-ke0 * (wt(2.0) / 70.0) ** 0.75 * 2.0)
    m = make("segment_dt")
    pred, _ = oracle.predict(m, m.flatten(s), np.array([[ke0, v]]))
    np.testing.assert_allclose(pred[:, 0], [p0, x / v], rtol=1e-14)
    # SEGMENT_END_ABS: wt at absolute segment end (4, then 6)
    x = 100.0 * math.exp(-ke0 * (wt(4.0) / 70.0) ** 0.75 * 4.0)
    x = x * math.exp(-ke0 * (wt(6.0) / 70.0) ** 0.75 * 2.0)
    m = make("segment_end_abs")
    pred2, _ = oracle.predict(m, m.flatten(s), np.array([[ke0, v]]))
    np.testing.assert_allclose(pred2[:, 0], [p0, x / v], rtol=1e-14)
    assert pred[1, 0] != pred2[1, 0]

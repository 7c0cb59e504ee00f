"""CPU-side tests of the product's host logic (no GPU, no compute calls into the device path):
  * the C-ABI library loads and exports every symbol include/pmx.h declares,
  * the population compiler's op stream (csrc/pmx_compile.cpp), interpreted here step by step with the
    oracle's single-kernel function, reproduces the oracle's predictions — i.e. flattening the
    reference's per-(subject, theta) event rewrite + Analytical::solve splitting once per subject is exact,
  * model / population validation and error codes,
  * the product fails loudly without a GPU (no silent CPU fallback).
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from pharmsol_amd import (Analytical, Data, Pow, Ratio, Scaled, Subject, _abi, _ffi, analytical, bolus, infusion,
                          runtime, synth)
from tests import models

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_of_the_header():
    hdr = open(os.path.join(ROOT, "include", "pmx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(pmx_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 20
    L = C.CDLL(_ffi.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, f"not exported: {missing}"
    bound = {n for n, _, _ in _ffi.SYMBOLS}
    assert names == bound, f"binding drift: {names ^ bound}"
    assert _ffi.lib().pmx_abi_version() == _abi.PMX_ABI_VERSION


def test_descriptor_layouts_match_the_library():
    L = _ffi.lib()
    assert L.pmx_sizeof_model_desc() == C.sizeof(_abi.pmx_model_desc)
    assert L.pmx_sizeof_population_desc() == C.sizeof(_abi.pmx_population_desc)


def test_no_device_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m, flat, theta = synth.config_c3(4, 2)
    assert runtime.device_count() == 0
    with pytest.raises(_abi.PmxError) as e:
        runtime.predict_host(m, flat, theta)
    assert e.value.status == _abi.PMX_ERR_NO_DEVICE
    with pytest.raises(_abi.PmxError):
        m.estimate_predictions(Subject.builder("x").infusion(0, 1, "iv", 1.0).build(), [0.1, 0.1, 0.1, 1.0])


def interpret_analytical(model, ops, theta):
    """Walk the op stream exactly as the device does, one (subject, theta) at a time."""
    d = model.desc()
    kname = model.kernel_name
    ns = _abi.KERNEL_STATE_COUNT[kname]
    out_state = d.out[0].state
    assert d.out[0].vol_src == _abi.PMX_SRC_PRIMARY
    v = theta[d.out[0].vol_index]
    preds = []
    for s in range(ops["n_subjects"]):
        x = np.zeros(ns)
        for o in range(ops["subj_op_off"][s], ops["subj_op_off"][s + 1]):
            k, io, a, b = ops["kind"][o], ops["io"][o], ops["a"][o], ops["b"][o]
            if k == _abi.PMX_OP_RESET:
                x[:] = 0.0
            elif k == _abi.PMX_OP_BOLUS:
                x[io] += a
            elif k == _abi.PMX_OP_OBS:
                preds.append(x[out_state] / v)
            else:
                x = oracle.kernel(kname, x, theta, a, [b])
    return np.array(preds)


@pytest.mark.parametrize("structure,central,theta,subject_fn,diffeq", models.KERNEL_CASES)
def test_op_stream_reproduces_oracle_on_reference_fixtures(structure, central, theta, subject_fn, diffeq):
    m = models.handwritten_analytical(structure, central, len(theta))
    flat = m.flatten(subject_fn())
    ops = runtime.compile_ops(m, flat)
    got = interpret_analytical(m, ops, np.array(theta))
    want, _ = oracle.predict(m, flat, np.array([theta]))
    np.testing.assert_array_equal(got, want[:, 0])  # same arithmetic on the same values: bit-exact


def test_op_stream_reproduces_oracle_on_ragged_random_subjects():
    rng = np.random.default_rng(7)
    subjects = [models.random_subject(rng, n_bolus_inputs=2, multi_occasion=True) for _ in range(60)]
    subjects.append(Subject.builder("empty").build())  # no events at all
    subjects.append(Subject.builder("only_dose").bolus(1.0, 5.0, 0).build())  # no observations
    m = models.handwritten_analytical("two_compartments", 0, 4)
    flat = m.flatten(Data(subjects))
    theta = np.array([0.17, 0.4, 0.25, 12.0])
    ops = runtime.compile_ops(m, flat)
    got = interpret_analytical(m, ops, theta)
    want, _ = oracle.predict(m, flat, theta.reshape(1, -1))
    assert got.shape[0] == flat.n_observations
    np.testing.assert_array_equal(got, want[:, 0])


def test_op_stream_subsegments_follow_analytical_solve():
    # analytical/mod.rs:313-357: breakpoints strictly inside, rate only where the piece is inside [s, s+dur]
    m = models.handwritten_analytical("one_compartment", 0, 2).with_ndrugs(1)
    s = (Subject.builder("seg").bolus(0.0, 0.0, 0).infusion(0.25, 1.0, 0, 0.25).observation(1.0, 0.0, 0).build())
    ops = runtime.compile_ops(m, m.flatten(s))
    prop = ops["kind"] == _abi.PMX_OP_PROP
    # bolus@0 -> infusion@0.25: one piece (0.25, rate 0); infusion@0.25 -> obs@1: [0.25,0.5] rate 4, [0.5,1] rate 0
    np.testing.assert_allclose(ops["a"][prop], [0.25, 0.25, 0.5])
    np.testing.assert_allclose(ops["b"][prop], [0.0, 4.0, 0.0])


def test_op_stream_covariates_follow_cov_time_mode():
    def make(mode):
        return analytical(name="m", params=["ke0", "v"], derived={"ke": Scaled("ke0", (Pow("wt", 70.0, 0.75),))},
                          covariates=["wt"], states=["central"], outputs=["cp"], routes=[bolus("iv", "central")],
                          structure="one_compartment", out={"cp": Ratio("central", "v")}, cov_time=mode)

    s = (Subject.builder("cov").bolus(0.0, 100.0, "iv").missing_observation(4.0, "cp").missing_observation(6.0, "cp")
         .covariate("wt", 0.0, 60.0).covariate("wt", 10.0, 90.0).build())
    for mode, want in (("segment_dt", [72.0, 66.0]), ("segment_end_abs", [72.0, 78.0])):
        m = make(mode)
        ops = runtime.compile_ops(m, m.flatten(s))
        prop = ops["kind"] == _abi.PMX_OP_PROP
        np.testing.assert_allclose(ops["cov"][prop, 0], want)
        obs = ops["kind"] == _abi.PMX_OP_OBS
        np.testing.assert_allclose(ops["cov"][obs, 0], [72.0, 78.0])  # out sees the absolute observation time


def test_ode_op_stream_pieces_and_step_counts():
    m = models.handwritten_ode("one_cmt_iv", 0, 2, h_max=0.02).with_ndrugs(1)
    s = (Subject.builder("ode").infusion(0.0, 100.0, 0, 0.5).missing_observation(0.5, 0).missing_observation(1.0, 0)
         .infusion(0.75, 40.0, 0, 1.0).missing_observation(2.0, 0).build())
    ops = runtime.compile_ops(m, m.flatten(s))
    prop = ops["kind"] == _abi.PMX_OP_PROP
    # pieces: [0,.5] r=200 | [.5,.75] r=0 | [.75,1] r=40 | [1,1.75] r=40 | [1.75,2] r=0
    np.testing.assert_allclose(ops["a"][prop], [0.5, 0.25, 0.25, 0.75, 0.25])
    np.testing.assert_allclose(ops["rate"][prop, 0], [200.0, 0.0, 40.0, 40.0, 0.0])
    n = ops["n"][prop]
    np.testing.assert_array_equal(n, np.ceil(ops["a"][prop] / 0.02).astype(int))
    np.testing.assert_allclose(ops["b"][prop] * n, ops["a"][prop], rtol=1e-15)


def test_lagged_boluses_leave_the_op_stream():
    # structs.rs:611-643: a lagged bolus is re-timed per support point, so it cannot sit in the shared stream
    m = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=3, lag={0: 2}).with_nstates(1).with_ndrugs(2).with_nout(1)
    s = (Subject.builder("lag").bolus(0.0, 100.0, 0).bolus(0.0, 7.0, 1).missing_observation(0.5, 0)
         .missing_observation(2.0, 0).build())
    ops = runtime.compile_ops(m, m.flatten(s))
    bol = ops["kind"] == _abi.PMX_OP_BOLUS
    assert bol.sum() == 1 and ops["io"][bol][0] == 1 and ops["a"][bol][0] == 7.0  # only the un-lagged input stays
    assert ops["max_input_used"] == 1


def test_pair_kernel_lane_order_sorts_subjects_by_work():
    m, flat, theta = synth.config_c4(200)
    ops = runtime.compile_ops(m, flat)
    work = np.array([(1 + ops["n"][ops["subj_op_off"][s]:ops["subj_op_off"][s + 1]]).sum() for s in range(200)])
    order = ops["subj_order"]
    assert sorted(order.tolist()) == list(range(200))
    assert (np.diff(work[order]) <= 0).all()


def test_input_and_outeq_range_metadata():
    m = models.handwritten_analytical("one_compartment", 0, 2).with_ndrugs(1)
    s = Subject.builder("oor").bolus(0.0, 1.0, 3).missing_observation(1.0, 2).build()
    ops = runtime.compile_ops(m, m.flatten(s))
    assert ops["max_input_used"] == 3 and ops["max_outeq"] == 2


def test_model_validation_errors():
    L = _ffi.lib()

    def create(d):
        h = C.c_void_p()
        rc = L.pmx_model_create(C.byref(d), C.byref(h))
        if h:
            L.pmx_model_destroy(h)
        return rc

    good = models.handwritten_analytical("two_compartments", 0, 4).desc()
    assert create(good) == _abi.PMX_OK
    d = models.handwritten_analytical("two_compartments", 0, 4).desc()
    d.kernel = 99
    assert create(d) == _abi.PMX_ERR_INVALID_ARGUMENT
    d = models.handwritten_analytical("two_compartments", 0, 4).desc()
    d.nstates = 1
    assert create(d) == _abi.PMX_ERR_INVALID_ARGUMENT
    d = models.handwritten_analytical("two_compartments", 0, 2).desc()  # too few params for [ke,kcp,kpc]
    assert create(d) == _abi.PMX_ERR_INVALID_ARGUMENT
    d = models.handwritten_analytical("two_compartments", 0, 4).desc()
    d.lag_param[0] = 1
    assert create(d) == _abi.PMX_OK  # lag time: merged per lane on the device
    d.lag_param[1] = 9
    assert create(d) == _abi.PMX_ERR_INVALID_ARGUMENT
    d = models.readme_analytical().desc()  # covariate-derived rate constant + lag: the closure walker, derive written out
    d.lag_param[0] = 0
    assert create(d) == _abi.PMX_OK
    d = models.handwritten_ode("one_cmt_iv", 0, 2).desc()
    d.lag_param[0] = 1
    assert create(d) == _abi.PMX_OK  # ODE lag: RK4 pieces split per lane on the device
    d.lag_param[0] = 2
    assert create(d) == _abi.PMX_ERR_INVALID_ARGUMENT
    d = models.handwritten_ode("one_cmt_iv", 0, 2).desc()
    d.rk4_h_max = 0.0
    assert create(d) == _abi.PMX_ERR_INVALID_ARGUMENT


def test_population_validation_errors():
    m = models.handwritten_analytical("one_compartment", 0, 2)
    flat = m.flatten(models.infusion_dosing_subject())
    L = _ffi.lib()
    d = flat.desc()
    md = m.desc()
    v = _abi.pmx_op_stream_view()
    assert L.pmx_debug_compile(C.byref(d), C.byref(md), C.byref(v)) == _abi.PMX_OK
    L.pmx_debug_free(C.byref(v))
    bad = flat.desc()
    bad.n_events = flat.n_events + 1  # offsets no longer span the events
    assert L.pmx_debug_compile(C.byref(bad), C.byref(md), C.byref(v)) == _abi.PMX_ERR_INVALID_ARGUMENT
    assert b"occ_ev_off" in L.pmx_last_error()
    flat.ev_kind[0] = 7
    assert L.pmx_debug_compile(C.byref(flat.desc()), C.byref(md), C.byref(v)) == _abi.PMX_ERR_INVALID_ARGUMENT


def test_library_sorts_like_occasion_sort():
    # events handed over unsorted with ties: Observation < Bolus < Infusion at equal times, stable
    m = models.handwritten_analytical("one_compartment", 0, 2).with_ndrugs(1)
    s = (Subject.builder("ties").infusion(1.0, 10.0, 0, 1.0).bolus(1.0, 5.0, 0).missing_observation(1.0, 0)
         .missing_observation(0.0, 0).build())
    flat = m.flatten(s)
    # scramble the stored order; the library must restore it
    perm = np.array([2, 0, 3, 1])
    for name in ("ev_time", "ev_value", "ev_duration", "ev_kind", "ev_io"):
        setattr(flat, name, np.ascontiguousarray(getattr(flat, name)[perm]))
    ops = runtime.compile_ops(m, flat)
    kinds = ops["kind"].tolist()
    # RESET, OBS(0), PROP(0->1), OBS(1), BOLUS(1), [infusion: no op], (no PROP after the last event)
    assert kinds == [_abi.PMX_OP_RESET, _abi.PMX_OP_OBS, _abi.PMX_OP_PROP, _abi.PMX_OP_OBS, _abi.PMX_OP_BOLUS]
    want, _ = oracle.predict(m, flat, np.array([[0.3, 2.0]]))
    assert want[0, 0] == 0.0 and want[1, 0] == 0.0


def test_graft_entry_build_is_consistent_with_the_library():
    """`__graft_entry__.build()` is the driver's "does it build" check: it must pass against the ABI the tree has."""
    import __graft_entry__ as g

    g.build()


def test_class_planner_exact_loose_and_generic_subjects():
    """pmx_debug_class_plan: a shared design -> exact classes; the same shape with individual times -> loose classes;
    shapes nobody shares and empty subjects -> the generic walker; covariate models -> loose classes only."""
    from pharmsol_amd import Analytical, Data, Ratio, Subject, runtime, synth

    m = synth.model_two_cpt_iv()
    exact = runtime.class_plan(m, synth.population_c23(1003))
    assert exact == dict(chunks_exact=126, chunks_loose=0, classed_subjects=1003, generic_subjects=0, members_per_chunk=8)
    loose = runtime.class_plan(m, synth.population_c23(1003, ragged=True))
    assert loose == dict(chunks_exact=0, chunks_loose=126, classed_subjects=1003, generic_subjects=0, members_per_chunk=8)
    # covariate-derived constants: the three-compartment rebuild gains nothing from batching and stays generic ...
    assert runtime.class_plan(synth.model_three_cpt_abs_wt(), synth.population_c5(50))["members_per_chunk"] == 0
    # ... the one- and two-state structures are classed by program shape (loose chunks), each member with its own factors
    from pharmsol_amd import Pow, Scaled, analytical, bolus

    small = analytical(name="wt1", params=["ka", "ke0", "v"], derived={"ke": Scaled("ke0", (Pow("wt", 70.0, 0.75),))},
                       covariates=["wt"], structure="one_compartment_with_absorption", states=["gut", "central"],
                       outputs=["cp"], routes=[bolus("oral", "gut")], out={"cp": Ratio("central", "v")})
    cov = runtime.class_plan(small, synth.population_c5(50))
    assert cov == dict(chunks_exact=0, chunks_loose=7, classed_subjects=50, generic_subjects=0, members_per_chunk=8)

    model = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=2).with_nstates(1).with_ndrugs(1).with_nout(1)
    rng = np.random.default_rng(0)
    subs = []
    for i in range(20):  # shared design
        subs.append(Subject.builder(f"e{i}").bolus(0.0, 100.0 + i, 0).missing_observation(1.0, 0).missing_observation(4.0, 0).build())
    for i in range(13):  # same shape, own times
        t = np.sort(rng.uniform(0.5, 9.0, 2))
        subs.append(Subject.builder(f"l{i}").bolus(0.0, 100.0, 0).missing_observation(float(t[0]), 0).missing_observation(float(t[1]), 0).build())
    for i in range(3):  # one more shape, too few members for a class of its own (min = G / 2 = 4)
        subs.append(Subject.builder(f"g{i}").bolus(0.0, 10.0, 0).missing_observation(float(rng.uniform(1, 2)), 0).build())
    subs.append(Subject.builder("empty").build())
    plan = runtime.class_plan(model, model.flatten(Data(subs)))
    # 20 exact -> 3 chunks of 8; 13 loose -> 2 chunks; 3 + the empty subject stay generic
    assert plan == dict(chunks_exact=3, chunks_loose=2, classed_subjects=33, generic_subjects=4, members_per_chunk=8)


def test_prediction_stores_of_the_write_bound_kernels_stay_fire_and_forget():
    """tools/isa_guard.py: on gfx950 loads and stores share one in-order counter, so a `s_waitcnt vmcnt(..)` that lands in
    the emit path (a rarely taken branch leaving a vector load pending at a join did it once: C3 0.83 -> 0.98 ms) makes
    every observation step wait for all earlier prediction stores.  The compiled assembly of every exact / loose prediction
    instantiation of the classed kernel must have no vmcnt wait in a block that holds a 16-byte prediction store.
    (Compiles pmx_kernels.hip to assembly once - about 90 s - and reuses it while the sources do not change.)"""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("isa_guard", os.path.join(root, "tools", "isa_guard.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    n, bad = g.check(g.assembly())
    assert n == 24 and not bad, bad


def test_recommended_row_pitch():
    # include/pmx.h pmx_recommended_ld: rows start on 128-byte boundaries
    from pharmsol_amd import runtime

    assert [runtime.recommended_ld(n) for n in (0, 1, 16, 17, 512, 1000, 1008)] == [0, 16, 16, 32, 512, 1008, 1008]

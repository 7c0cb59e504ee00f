"""API items around the hot path that a pharmsol caller uses to drive it:
``ParameterOrder`` (parameter_order.rs:12-116, parameters.rs:104-165), ``Prediction`` fields / ``PopulationPredictions``
(prediction.rs:18-27, subject.rs:140-165), per-subject ``estimate_log_likelihood`` / ``simulate_subject``
(equation/mod.rs:468-477,569-576), ``log_likelihood_batch`` with prediction-based ``ResidualErrorModels`` and its
failure -> -inf rule (likelihood/mod.rs:119-177, residual_error.rs:178-271), the host-pointer entry points' persistent
workspace (pinned or pageable outputs)."""
import math

import numpy as np
import pytest

import oracle
from pharmsol_amd import (Analytical, AssayErrorModel, AssayErrorModels, Censor, Data, ErrorPoly, ParameterError,
                          ParameterOrder, Parameters, Ratio, ResidualErrorModel, ResidualErrorModels, Subject, _abi, runtime,
                          synth)
from tests import models


# --------------------------------------------------------------------------- ParameterOrder (no GPU)
def test_parameter_order_permutes_values_and_matrices_into_model_order():
    m = models.readme_analytical()  # params ka, ke0, v
    order = ParameterOrder.with_model(m, ["v", "ka", "ke0"])
    assert order.permutation() == [1, 2, 0] and order.width() == 3 and not order.is_identity()
    np.testing.assert_array_equal(order.values([50.0, 1.0, 0.2]), [1.0, 0.2, 50.0])
    mat = order.matrix(np.array([[50.0, 1.0, 0.2], [60.0, 1.5, 0.3]]))
    np.testing.assert_array_equal(mat, [[1.0, 0.2, 50.0], [1.5, 0.3, 60.0]])
    assert mat.flags["C_CONTIGUOUS"]  # the row-major theta the library takes (matrix.rs:62-65)
    assert ParameterOrder.with_model(m, ["ka", "ke0", "v"]).is_identity()
    np.testing.assert_array_equal(order.parameters([50.0, 1.0, 0.2]).as_slice(),
                                  Parameters.with_model(m, [("ka", 1.0), ("ke0", 0.2), ("v", 50.0)]).as_slice())


def test_parameter_order_errors():
    m = models.readme_analytical()
    with pytest.raises(ParameterError, match="UnknownParameter"):
        ParameterOrder.with_model(m, ["ka", "ke0", "volume"])
    with pytest.raises(ParameterError, match="DuplicateParameter"):
        ParameterOrder.with_model(m, ["ka", "ka", "v"])
    with pytest.raises(ParameterError, match="MissingParameters"):
        ParameterOrder.with_model(m, ["ka", "v"])
    with pytest.raises(ParameterError, match="WidthMismatch"):
        ParameterOrder.with_model(m, ["v", "ka", "ke0"]).matrix(np.zeros((2, 4)))


def test_residual_error_model_sigma_rules():
    # residual_error.rs:178-191 (the doc examples :57-66) and the floor at sqrt(f64::EPSILON)
    assert ResidualErrorModel.constant(0.5).sigma(100.0) == 0.5
    assert abs(ResidualErrorModel.proportional(0.1).sigma(-100.0) - 10.0) < 1e-12
    assert abs(ResidualErrorModel.combined(0.5, 0.1).sigma(100.0) - math.sqrt(100.25)) < 1e-12
    assert ResidualErrorModel.proportional(0.1).sigma(0.0) == math.sqrt(2.220446049250313e-16)


def _batch_case(n=37, seed=3):
    rng = np.random.default_rng(seed)
    m = models.handwritten_analytical("two_compartments", 0, 4).with_nout(2)
    m.out = {0: Ratio(0, 3), 1: Ratio(1, None)}
    subs = []
    for i in range(n):
        b = Subject.builder(f"b{i}").infusion(0.0, 300.0 + i, 0, 0.5).bolus(6.0, 50.0, 0)
        for k, t in enumerate(np.sort(rng.uniform(0.6, 24.0, 6))):
            b = b.observation(float(t), float(rng.uniform(0.5, 6.0)), k % 2) if k != 2 else b.missing_observation(float(t), 0)
        subs.append(b.build())
    theta = synth.theta_c3(n, synth.SplitMix64(seed))
    rem = (ResidualErrorModels.new().add(0, ResidualErrorModel.combined(0.3, 0.1)).add(1, ResidualErrorModel.proportional(0.2)))
    return m, subs, theta, rem


def test_oracle_residual_log_likelihood_matches_the_reference_formula():
    m, subs, theta, rem = _batch_case(5)
    flat = m.flatten(Data(subs))
    ll, _ = oracle.loglik(m, flat, rem, theta[:1])
    pred, _ = oracle.predict(m, flat, theta[:1])
    row = 0
    for s, sub in enumerate(subs):
        want = 0.0
        for ev in sub.occasions[0].events:
            if hasattr(ev, "outeq"):
                if ev.value is not None:
                    em = rem._m[int(ev.outeq)]
                    sg = em.sigma(pred[row, 0])
                    want += -0.5 * (math.log(2 * math.pi) + 2 * math.log(sg) + ((ev.value - pred[row, 0]) / sg) ** 2)
                row += 1
        assert abs(ll[s, 0] - want) < 1e-10 * max(1.0, abs(want))


# --------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_log_likelihood_batch_and_the_failure_to_minus_infinity_rule():
    m, subs, theta, rem = _batch_case()
    theta[5, 1], theta[5, 0], theta[5, 2] = -3.0, 1.0, 1.0  # complex eigenvalues for subject 5 (the reference panics; Err -> -inf)
    got = m.log_likelihood_batch(Data(subs), theta, rem)
    assert runtime.last_kernel_name().startswith("pmx_analytical_pair")
    assert got.shape == (len(subs),) and got[5] == -np.inf
    flat = m.flatten(Data(subs))
    for i in (0, 1, 7, 20, 36):
        want, _ = oracle.loglik(m, flat.subject_slice(i, i + 1), rem, theta[i:i + 1])
        assert abs(got[i] - want[0, 0]) <= 1e-9 * max(1.0, abs(want[0, 0]))
    with pytest.raises(ValueError):
        m.log_likelihood_batch(Data(subs), theta[:-1], rem)


@pytest.mark.gpu
@pytest.mark.parametrize("n_support,shared", [(70, True), (70, False), (5, False)])
def test_residual_error_models_through_the_matrix_entry(n_support, shared):
    """Prediction-based sigma inside the fused kernels: classed (exact / loose chunks fold from the full records),
    generic GRID and PAIR, against the oracle."""
    import torch

    m, subs, theta, rem = _batch_case(43, seed=4)
    if shared:  # a shared design: exact classes
        proto = subs[0]
        subs = []
        for i in range(43):
            b = Subject.builder(f"s{i}").infusion(0.0, 300.0 + i, 0, 0.5).bolus(6.0, 50.0, 0)
            for ev in proto.occasions[0].events:
                if hasattr(ev, "outeq"):
                    b = b.observation(ev.time, ev.value * (1 + 0.01 * i), ev.outeq) if ev.value is not None else b.missing_observation(ev.time, ev.outeq)
            subs.append(b.build())
    flat = m.flatten(Data(subs))
    th = synth.theta_c3(n_support)
    ll, st = runtime.loglik(m, runtime.DevicePopulation(flat, 0), rem, th)
    torch.cuda.synchronize()
    want, wst = oracle.loglik(m, flat, rem, th)
    np.testing.assert_array_equal(st.cpu().numpy(), wst)
    assert (np.abs(ll.cpu().numpy() - want) / np.maximum(np.abs(want), 1.0)).max() < 1e-9


@pytest.mark.gpu
def test_per_subject_entry_points_and_result_containers():
    m = models.readme_analytical()
    subj = (Subject.builder("one").bolus(0.0, 500.0, "oral").observation(0.5, 1.2, "cp")
            .censored_observation(1.0, 0.3, "cp", Censor.BLOQ).missing_observation(2.0, "cp").covariate("wt", 0.0, 75.0).build())
    p = Parameters.with_model(m, [("ka", 1.0), ("ke0", 0.08), ("v", 150.0)])
    em = AssayErrorModels.empty().add("cp", AssayErrorModel.additive(ErrorPoly(0.1, 0.1, 0.0, 0.0), 0.0))
    preds = m.estimate_predictions(subj, p, with_state=True)
    flat = m.flatten(subj)
    want, _ = oracle.predict(m, flat, p.as_slice())
    np.testing.assert_allclose(preds.flat_predictions(), want[:, 0], rtol=1e-9)
    # Prediction.state = the amounts at the observation: central / v is the prediction
    for pr in preds.predictions():
        assert len(pr.state) == 2 and abs(pr.state[1] / 150.0 - pr.prediction) < 1e-9 * max(1.0, pr.prediction)
    assert preds.predictions()[0].censoring == Censor.NONE and preds.predictions()[1].censoring == Censor.BLOQ
    assert preds.predictions()[2].observation is None
    assert abs(preds.squared_error() - sum((o - q) ** 2 for o, q in zip(preds.flat_observations(), preds.flat_predictions()) if o is not None)) < 1e-12
    ll = m.estimate_log_likelihood(subj, p, em)
    wll, _ = oracle.loglik(m, flat, em, p.as_slice())
    assert abs(ll - wll[0, 0]) < 1e-9 * max(1.0, abs(ll))
    preds2, lik = m.simulate_subject(subj, p, em)
    assert abs(lik - math.exp(ll)) < 1e-12 * max(1.0, lik) and preds2.flat_predictions() == preds.flat_predictions()
    assert m.simulate_subject(subj, p)[1] is None
    # PopulationPredictions: subjects x support points from one device pass
    th = np.stack([p.as_slice(), p.as_slice() * 1.1])
    pp = m.population_predictions(Data([subj, subj]), th)
    assert pp.shape == (2, 2)
    np.testing.assert_allclose(pp[1, 0].flat_predictions(), preds.flat_predictions(), rtol=1e-12)
    w2, _ = oracle.predict(m, flat, th)
    np.testing.assert_allclose(pp[0, 1].flat_predictions(), w2[:, 1], rtol=1e-9)


@pytest.mark.gpu
def test_host_pointer_entry_points_reuse_their_workspace_and_take_pinned_or_pageable_outputs():
    """pmx_predict / pmx_loglik keep device buffers, streams and pinned staging on the population handle: growing and
    shrinking calls, pageable and page-locked outputs, padded leading dimensions - the same numbers every time."""
    import ctypes as C

    from pharmsol_amd import _ffi

    L = _ffi.lib()
    m, flat, theta = synth.config_c3(3000, 96)
    dm, pop = runtime._as_model(m), runtime.DevicePopulation(flat, 0)
    want, _ = oracle.predict(m, flat, theta)
    NO, P = pop.n_observations, 96
    for trial, (npts, ld, pinned) in enumerate([(96, 96, False), (17, 17, True), (96, 100, False), (96, 128, True), (96, 96, True)]):
        out = (runtime.host_empty((NO, ld)) if pinned else np.empty((NO, ld)))
        out[:] = -7.0
        st = np.zeros((pop.n_subjects, npts), dtype=np.uint8)
        th = np.ascontiguousarray(theta[:npts])
        _ffi.check(L.pmx_predict(dm.handle, pop.handle, th.ctypes.data, npts, out.ctypes.data, ld, st.ctypes.data))
        err = np.abs(out[:, :npts] - want[:, :npts]) / np.maximum(np.abs(want[:, :npts]), 1e-12)
        assert err.max() < 1e-9 and not st.any(), trial
        assert (out[:, npts:] == -7.0).all()  # the caller's padding columns are left alone
    # a matrix larger than the 32 MB bounce buffers through the pageable path (several pieces, split at row ends)
    m2, flat2, theta2 = synth.config_c3(9000, 128)
    pop2 = runtime.DevicePopulation(flat2, 0)
    out = np.empty((pop2.n_observations, 130))
    _ffi.check(L.pmx_predict(runtime._as_model(m2).handle, pop2.handle, theta2.ctypes.data, 128, out.ctypes.data, 130, None))
    w2, _ = oracle.predict(m2, flat2.subject_slice(8990, 9000), theta2)
    np.testing.assert_allclose(out[-w2.shape[0]:, :128], w2, rtol=1e-9)
    # failures: the flag comes from a device-side reduction, the status array is optional
    bad = theta.copy()
    bad[3, 1] = -1.5
    bad[3, 0], bad[3, 2] = 1.0, 1.0  # (ke + kcp + kpc)^2 < 4 ke kpc: complex eigenvalues
    sink = np.empty((NO, 96))
    rc = L.pmx_predict(dm.handle, pop.handle, bad.ctypes.data, 96, sink.ctypes.data, 96, None)
    assert rc == _abi.PMX_ERR_PAIR_FAILED and np.isnan(sink[:, 3]).all() and np.isfinite(np.delete(sink, 3, axis=1)).all()

"""ctypes binding of the CPU ORACLE (``oracle/pmx_oracle.c``).

TEST INFRASTRUCTURE ONLY — see ``oracle/pmx_oracle.h``.  Allowed importers:
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
The product package ``pharmsol_amd`` never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

from pharmsol_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libpmx_oracle.so")
_lib = None

K_TEST_SEQ_ACCUM = 1000
K_TEST_RATEIV3 = 1001


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "pmx_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH)
        for p in (src, os.path.join(_HERE, "pmx_oracle.h"), os.path.join(_HERE, "..", "include", "pmx.h")))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.pmx_oracle_predict.restype = C.c_int32
        L.pmx_oracle_predict.argtypes = [C.POINTER(_abi.pmx_model_desc), C.POINTER(_abi.pmx_population_desc),
                                         C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        L.pmx_oracle_predict_batch.restype = C.c_int32
        L.pmx_oracle_predict_batch.argtypes = [C.POINTER(_abi.pmx_model_desc), C.POINTER(_abi.pmx_population_desc),
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.pmx_oracle_kernel.restype = C.c_int32
        L.pmx_oracle_kernel.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                        C.c_void_p]
        L.pmx_oracle_cov_interpolate.restype = C.c_int32
        L.pmx_oracle_cov_interpolate.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double,
                                                 C.POINTER(C.c_double)]
        L.pmx_oracle_loglik.restype = C.c_int32
        L.pmx_oracle_loglik.argtypes = [C.POINTER(_abi.pmx_model_desc), C.POINTER(_abi.pmx_population_desc),
                                        C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        L.pmx_oracle_sigma.restype = C.c_int32
        L.pmx_oracle_sigma.argtypes = [C.POINTER(_abi.pmx_error_model), C.c_double, C.POINTER(C.c_double)]
        L.pmx_oracle_lognormpdf.restype = C.c_double
        L.pmx_oracle_lognormpdf.argtypes = [C.c_double, C.c_double, C.c_double]
        L.pmx_oracle_lognormcdf.restype = C.c_int32
        L.pmx_oracle_lognormcdf.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_double)]
        L.pmx_oracle_last_error.restype = C.c_char_p
        L.pmx_oracle_max_threads.restype = C.c_int32
        L.pmx_oracle_sizeof_model_desc.restype = C.c_int64
        L.pmx_oracle_sizeof_population_desc.restype = C.c_int64
        assert L.pmx_oracle_sizeof_model_desc() == C.sizeof(_abi.pmx_model_desc), "pmx_model_desc layout drift"
        assert L.pmx_oracle_sizeof_population_desc() == C.sizeof(_abi.pmx_population_desc)
        _lib = L
    return _lib


def compile_custom(source: str, has_init: bool = False) -> None:
    """Build the SAME user source the device compiles (``ODE.custom``) with gcc (``-ffp-contract=off``) and register
    its bodies with the oracle.  The handles stay alive in this module (one custom model at a time: tests only)."""
    import hashlib
    import tempfile

    global _custom_lib
    h = hashlib.sha1(source.encode()).hexdigest()[:16]
    d = os.path.join(tempfile.gettempdir(), "pmx_oracle_custom")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, f"m{h}.so")
    if not os.path.exists(so):
        src = os.path.join(d, f"m{h}.cpp")
        with open(src, "w") as f:
            f.write('#include <cmath>\nusing namespace std;\n#define PMX_DEVICE extern "C"\n#line 1 "model"\n' + source)
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src], check=True)
    cl = C.CDLL(so)
    fn = lambda n: C.cast(getattr(cl, n), C.c_void_p)  # noqa: E731
    L = lib()
    L.pmx_oracle_set_custom.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pmx_oracle_set_custom.restype = None
    L.pmx_oracle_set_custom(fn("pmx_dynamics"), fn("pmx_outputs"), fn("pmx_init") if has_init else None)
    _custom_lib = cl


_custom_lib = None
_user_lib = None
_user_key = None


def _register_user(model) -> None:
    """User closures of an Analytical or ODE model: build the model's source with g++ (the SAME text the device
    compiles) and register its bodies (pmx_oracle_set_user); a descriptor model clears the registration.  One model at
    a time: tests only."""
    import hashlib
    import tempfile

    global _user_lib, _user_key
    L = lib()
    L.pmx_oracle_set_user.argtypes = [C.c_uint32, C.c_void_p]
    L.pmx_oracle_set_user.restype = None
    mask = int(getattr(model, "user_fns", 0) or 0)
    if not mask or getattr(model, "eq_kind", None) not in (_abi.PMX_EQ_ANALYTICAL, _abi.PMX_EQ_ODE):
        if _user_key is not None:
            L.pmx_oracle_set_user(0, None)
            _user_key = None
        return
    source = model.source
    key = hashlib.sha1(source.encode()).hexdigest()[:16]
    if key == _user_key:
        return
    d = os.path.join(tempfile.gettempdir(), "pmx_oracle_custom")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, f"u{key}.so")
    if not os.path.exists(so):
        src = os.path.join(d, f"u{key}.cpp")
        with open(src, "w") as f:
            f.write('#include <cmath>\nusing namespace std;\n#define PMX_DEVICE extern "C"\n#line 1 "model"\n' + source)
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src], check=True)
    cl = C.CDLL(so)
    fns = (C.c_void_p * 9)()
    for name, bit in _abi.USER_FUNCTION_BITS.items():
        if mask & bit:
            fns[bit.bit_length() - 1] = C.cast(getattr(cl, name), C.c_void_p)
    L.pmx_oracle_set_user(mask, fns)
    _user_lib, _user_key = cl, key


def max_threads() -> int:
    return int(lib().pmx_oracle_max_threads())


def _model_desc(model):
    return model if isinstance(model, _abi.pmx_model_desc) else model.desc()


def predict(model, flat, theta: np.ndarray, nthreads: int = 0, allow_pair_failures: bool = True
            ) -> Tuple[np.ndarray, np.ndarray]:
    """Oracle twin of ``pmx_predict``: returns ``(pred[n_obs, P], status[S, P])``."""
    L = lib()
    _register_user(model)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    if theta.ndim == 1:
        theta = theta.reshape(1, -1)
    md = _model_desc(model)
    assert theta.shape[1] == md.nparams, (theta.shape, md.nparams)
    P = theta.shape[0]
    pd = flat.desc()
    pred = np.full((flat.n_observations, P), np.nan, dtype=np.float64)
    status = np.zeros((flat.n_subjects, P), dtype=np.uint8)
    rc = L.pmx_oracle_predict(C.byref(md), C.byref(pd), theta.ctypes.data, P, pred.ctypes.data, P,
                              status.ctypes.data, nthreads)
    if rc != _abi.PMX_OK and not (rc == _abi.PMX_ERR_PAIR_FAILED and allow_pair_failures):
        raise _abi.PmxError(rc, L.pmx_oracle_last_error().decode())
    return pred, status


def predict_batch(model, flat, theta: np.ndarray, nthreads: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Oracle twin of ``pmx_predict_batch`` (subject s with theta row s)."""
    L = lib()
    _register_user(model)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    md = _model_desc(model)
    assert theta.shape == (flat.n_subjects, md.nparams)
    pd = flat.desc()
    pred = np.full((flat.n_observations,), np.nan, dtype=np.float64)
    status = np.zeros((flat.n_subjects,), dtype=np.uint8)
    rc = L.pmx_oracle_predict_batch(C.byref(md), C.byref(pd), theta.ctypes.data, pred.ctypes.data,
                                    status.ctypes.data, nthreads)
    if rc not in (_abi.PMX_OK, _abi.PMX_ERR_PAIR_FAILED):
        raise _abi.PmxError(rc, L.pmx_oracle_last_error().decode())
    return pred, status


def kernel(name_or_id, x, p, t: float, rateiv, pm: bool = False) -> np.ndarray:
    """One closed-form kernel call (``AnalyticalEq``)."""
    kid = _abi.ANALYTICAL_KERNELS[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
    x = np.ascontiguousarray(x, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    r = np.ascontiguousarray(rateiv, dtype=np.float64)
    out = np.zeros_like(x)
    rc = lib().pmx_oracle_kernel(kid, 1 if pm else 0, x.ctypes.data, p.ctypes.data, float(t), r.ctypes.data,
                                 out.ctypes.data)
    if rc != 0:
        raise _abi.PmxError(_abi.PMX_ERR_PAIR_FAILED, f"kernel status {rc}")
    return out


def cov_interpolate(knots_t, knots_v, t: float, fixed: bool = False) -> float:
    kt = np.ascontiguousarray(knots_t, dtype=np.float64)
    kv = np.ascontiguousarray(knots_v, dtype=np.float64)
    v = C.c_double()
    rc = lib().pmx_oracle_cov_interpolate(kt.ctypes.data, kv.ctypes.data, kt.shape[0], 1 if fixed else 0, float(t),
                                          C.byref(v))
    if rc != 0:
        raise _abi.PmxError(rc, "MissingSegments")
    return v.value


def loglik(model, flat, error_models, theta: np.ndarray, nthreads: int = 0, allow_pair_failures: bool = True
           ) -> Tuple[np.ndarray, np.ndarray]:
    """Oracle twin of ``pmx_loglik``: returns ``(ll[S, P], status[S, P])``."""
    L = lib()
    _register_user(model)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    if theta.ndim == 1:
        theta = theta.reshape(1, -1)
    md = _model_desc(model)
    P = theta.shape[0]
    pd = flat.desc()
    em = error_models.to_c(model)
    ll = np.full((flat.n_subjects, P), np.nan, dtype=np.float64)
    status = np.zeros((flat.n_subjects, P), dtype=np.uint8)
    rc = L.pmx_oracle_loglik(C.byref(md), C.byref(pd), C.cast(em, C.c_void_p), theta.ctypes.data, P, ll.ctypes.data, P,
                             status.ctypes.data, nthreads)
    if rc != _abi.PMX_OK and not (rc == _abi.PMX_ERR_PAIR_FAILED and allow_pair_failures):
        raise _abi.PmxError(rc, L.pmx_oracle_last_error().decode())
    return ll, status


def sigma(em, observation: float) -> float:
    """``AssayErrorModel::sigma`` for one observed value (em: pharmsol_amd.error_model.AssayErrorModel)."""
    c = _abi.pmx_error_model()
    c.kind, c.scalar = em.kind, em.scalar
    c.c[0], c.c[1], c.c[2], c.c[3] = em.poly.c0, em.poly.c1, em.poly.c2, em.poly.c3
    out = C.c_double()
    rc = lib().pmx_oracle_sigma(C.byref(c), float(observation), C.byref(out))
    if rc != 0:
        raise _abi.PmxError(rc, "ErrorModelError")
    return out.value


def lognormpdf(obs: float, pred: float, sigma_: float) -> float:
    return float(lib().pmx_oracle_lognormpdf(float(obs), float(pred), float(sigma_)))


def lognormcdf(obs: float, pred: float, sigma_: float, upper: bool = False) -> float:
    """``lognormcdf`` / ``lognormccdf`` (upper=True); raises on the reference's Err branches."""
    out = C.c_double()
    rc = lib().pmx_oracle_lognormcdf(float(obs), float(pred), float(sigma_), 1 if upper else 0, C.byref(out))
    if rc != 0:
        raise _abi.PmxError(rc, "ErrorModelError::NegativeSigma")
    return out.value

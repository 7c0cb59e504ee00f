"""Device-resident handles over the C ABI.

``DevicePopulation`` = ``pmx_population*`` (flattened events uploaded once, like the
reference's ``Data`` living across NPAG cycles); ``DeviceModel`` = ``pmx_model*``.
``predict`` takes/returns torch tensors resident in HBM and enqueues on torch's
current stream (PyTorch is plumbing here: device memory + streams).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _abi, _ffi
from .flatten import FlatPopulation


def device_count() -> int:
    return int(_ffi.lib().pmx_device_count())


class DeviceModel:
    def __init__(self, model):
        self.model = model
        self.desc = model.desc() if not isinstance(model, _abi.pmx_model_desc) else model
        h = C.c_void_p()
        src = getattr(model, "source", None)
        if src is not None and getattr(model, "user_fns", 0):  # user closures of an analytical model (hiprtc)
            _ffi.check(_ffi.lib().pmx_model_create_user(C.byref(self.desc), src.encode(), int(model.user_fns), C.byref(h)))
        elif src is not None:  # user ODE body: compiled for gfx950 with hiprtc inside the library
            _ffi.check(_ffi.lib().pmx_model_create_custom(C.byref(self.desc), src.encode(), 1 if model.has_init else 0,
                                                          C.byref(h)))
        else:
            _ffi.check(_ffi.lib().pmx_model_create(C.byref(self.desc), C.byref(h)))
        self.handle = h

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and _ffi is not None and _ffi._lib is not None:  # (module globals vanish at interpreter exit)
            _ffi._lib.pmx_model_destroy(h)
            self.handle = None


class DevicePopulation:
    def __init__(self, flat: FlatPopulation, device: int = 0, subjects: Optional[Tuple[int, int]] = None):
        """``subjects=(s0, s1)``: only that subject range of ``flat`` (``pmx_population_create_shard``: one rank's shard
        of a population every rank holds on the host, no re-based copy)."""
        L = _ffi.lib()
        self.flat = flat
        self.device = int(device)
        d = flat.desc()
        h = C.c_void_p()
        if subjects is None:
            _ffi.check(L.pmx_population_create(C.byref(d), self.device, C.byref(h)))
        else:
            _ffi.check(L.pmx_population_create_shard(C.byref(d), int(subjects[0]), int(subjects[1]), self.device, C.byref(h)))
        self.handle = h
        self.n_subjects = int(L.pmx_population_n_subjects(h))
        self.n_observations = int(L.pmx_population_n_observations(h))
        self.n_events = int(L.pmx_population_n_events(h))

    def observation_offsets(self) -> np.ndarray:
        off = np.zeros(self.n_subjects + 1, dtype=np.int64)
        _ffi.check(_ffi.lib().pmx_population_observation_offsets(self.handle, off.ctypes.data))
        return off

    def observation_info(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        t = np.zeros(self.n_observations, dtype=np.float64)
        o = np.zeros(self.n_observations, dtype=np.int32)
        s = np.zeros(self.n_observations, dtype=np.int64)
        _ffi.check(_ffi.lib().pmx_population_observation_info(self.handle, t.ctypes.data, o.ctypes.data, s.ctypes.data))
        return t, o, s

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and _ffi is not None and _ffi._lib is not None:
            _ffi._lib.pmx_population_destroy(h)
            self.handle = None


def alloc_predictions(model, pop: DevicePopulation, theta, tries: int = 4, reps: int = 5, log=None):
    """A prediction matrix ``[n_observations, n_support]`` placed where the kernel writes fastest.

    The prediction stream is a row-strided scatter; on MI355X its rate depends on WHICH allocation it lands in
    (tools/experiments/store_pattern_probe.hip ``alloc``, tools/experiments/alloc_tune.py: the same kernel takes 0.94-0.97 ms in most 5.6 GB
    allocations and 1.10-1.12 ms in others, stable for the life of the allocation).  This helper allocates up to
    ``tries`` candidates (all alive at once, so that each sits on different memory), times ``reps`` passes of the real
    kernel into each, keeps the fastest and frees the rest.  A caller reuses the returned buffer across passes."""
    import torch

    dev = torch.device("cuda", pop.device)
    if not (isinstance(theta, torch.Tensor) and theta.is_cuda):
        theta = torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64), device=dev)
    theta = theta.contiguous()
    P = int(theta.shape[0])
    best, best_ms, held = None, float("inf"), []
    for _ in range(max(1, tries)):
        try:
            cand = torch.empty((pop.n_observations, P), dtype=torch.float64, device=dev)
        except RuntimeError:  # out of memory: keep what we have
            break
        held.append(cand)
        ms_c = C.c_double()
        if len(held) == 1:  # bring the device clocks up first, or the first candidate is judged on a cold device
            _ffi.check(_ffi.lib().pmx_time_predict_device(_as_model(model).handle, pop.handle, theta.data_ptr(), P,
                                                          cand.data_ptr(), P, 40, torch.cuda.current_stream(dev).cuda_stream,
                                                          C.byref(ms_c)))
        _ffi.check(_ffi.lib().pmx_time_predict_device(_as_model(model).handle, pop.handle, theta.data_ptr(), P, cand.data_ptr(),
                                                      P, reps, torch.cuda.current_stream(dev).cuda_stream, C.byref(ms_c)))
        ms = ms_c.value
        if log is not None:
            log.append((int(cand.data_ptr()), ms))
        if ms < best_ms:
            best, best_ms = cand, ms
        if ms * 1.0e3 / max(cand.numel() * 8, 1) > 2.0e-6:  # < ~0.5 TB/s of output: not write-bound, placement is moot
            break
    held.clear()
    torch.cuda.empty_cache()
    return best


class _PlacedBuffer:
    """Device memory owned by ``pmx_prediction_buffer_create``, exposed to torch through ``__cuda_array_interface__``."""

    def __init__(self, ptr: int, shape, ms: float):
        self.ptr, self.shape, self.ms_per_pass = int(ptr), tuple(int(x) for x in shape), float(ms)
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": "<f8", "data": (self.ptr, False), "version": 3,
                                         "strides": None}

    def __del__(self):
        if getattr(self, "ptr", 0) and _ffi is not None and _ffi._lib is not None:
            _ffi._lib.pmx_prediction_buffer_destroy(C.c_void_p(self.ptr))
            self.ptr = 0


def place_predictions(model, pop: DevicePopulation, theta, search_gib: float = 48.0, exhaustive: bool = False, ld: int = 0):
    """A prediction matrix ``[n_observations, n_support]`` in a fast window of an arena of up to ``search_gib``
    (``pmx_prediction_buffer_create``: physical chunks mapped through the HIP virtual-memory API window by window, the
    real kernel timed into each; the search stops inside the first fast plateau, or - ``exhaustive`` - times every window
    and keeps the best; everything outside the chosen window is returned to the device).  ``ld`` > n_support: rows that
    many doubles apart (``pmx_prediction_buffer_create_pitched``), the tensor returned is the ``[:, :n_support]`` view.
    Returns a CUDA tensor; the memory lives as long as the tensor (``tensor._pmx_owner``)."""
    import torch

    dev = torch.device("cuda", pop.device)
    if not (isinstance(theta, torch.Tensor) and theta.is_cuda):
        theta = torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64), device=dev)
    theta = theta.contiguous()
    P = int(theta.shape[0])
    out, ms = C.c_void_p(), C.c_double()
    pitch = max(int(ld), P)
    with torch.cuda.device(dev):
        _ffi.check(_ffi.lib().pmx_prediction_buffer_create_pitched(_as_model(model).handle, pop.handle, theta.data_ptr(), P, pitch,
                                                                   int(search_gib * (1 << 30)) * (-1 if exhaustive else 1),
                                                                   torch.cuda.current_stream(dev).cuda_stream, C.byref(out),
                                                                   C.byref(ms)))
    owner = _PlacedBuffer(out.value, (pop.n_observations, pitch), ms.value)
    base = torch.as_tensor(owner, device=dev)
    base._pmx_owner = owner
    if pitch == P:
        return base
    t = base[:, :P]
    t._pmx_owner = owner
    return t


def recommended_ld(n_support: int) -> int:
    """Row pitch (doubles) at which the prediction kernels write fastest (``pmx_recommended_ld``: 128-byte row starts)."""
    return int(_ffi.lib().pmx_recommended_ld(int(n_support)))


def predict_states(model, pop: DevicePopulation, theta, states=None):
    """``Prediction::state`` (likelihood/prediction.rs:18-27) for every observation: a CUDA tensor
    ``[n_observations, len(states), n_support]`` (default: all model states), one ``pmx_predict_state_device`` call
    per state on torch's current stream."""
    import torch

    L = _ffi.lib()
    dm = _as_model(model)
    dev = torch.device("cuda", pop.device)
    if not (isinstance(theta, torch.Tensor) and theta.is_cuda):
        theta = torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64), device=dev)
    theta = theta.contiguous()
    P = int(theta.shape[0])
    states = list(range(dm.desc.nstates)) if states is None else [int(s) for s in states]
    out = torch.empty((len(states), pop.n_observations, P), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for k, st in enumerate(states):
        _ffi.check(L.pmx_predict_state_device(dm.handle, pop.handle, theta.data_ptr(), P, st, out[k].data_ptr(), P, None,
                                              stream))
    return out.permute(1, 0, 2)


def jit_translation_unit(model) -> str:
    """The source text the library hands to hiprtc for a custom ODE model (``pmx_debug_jit_source``)."""
    L = _ffi.lib()
    d = model.desc()
    out = C.c_void_p()
    if getattr(model, "user_fns", 0):
        _ffi.check(L.pmx_debug_jit_source_user(C.byref(d), model.source.encode(), int(model.user_fns), C.byref(out)))
    else:
        _ffi.check(L.pmx_debug_jit_source(C.byref(d), model.source.encode(), 1 if model.has_init else 0, C.byref(out)))
    try:
        return C.cast(out, C.c_char_p).value.decode()
    finally:
        L.pmx_free_text(out)


def _as_model(model) -> DeviceModel:
    if isinstance(model, DeviceModel):
        return model
    dm = getattr(model, "_handle", None)
    if dm is None:
        dm = DeviceModel(model)
        model._handle = dm
    return dm


def predict_host(model, flat: FlatPopulation, theta: np.ndarray, device: int = 0, batch: bool = False,
                 raise_on_pair_failure: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """Host-pointer ABI form (``pmx_predict`` / ``pmx_predict_batch``): numpy in, numpy out."""
    L = _ffi.lib()
    dm = _as_model(model)
    pop = DevicePopulation(flat, device)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    if theta.ndim == 1:
        theta = theta.reshape(1, -1)
    if theta.shape[1] != dm.desc.nparams:
        raise ValueError(f"theta has {theta.shape[1]} columns, model declares {dm.desc.nparams} parameters")
    if batch:
        if theta.shape[0] != pop.n_subjects:
            raise ValueError("batch form needs one theta row per subject")
        pred = np.full((pop.n_observations,), np.nan)
        status = np.zeros((pop.n_subjects,), dtype=np.uint8)
        rc = L.pmx_predict_batch(dm.handle, pop.handle, theta.ctypes.data, pred.ctypes.data, status.ctypes.data)
    else:
        P = theta.shape[0]
        pred = np.full((pop.n_observations, P), np.nan)
        status = np.zeros((pop.n_subjects, P), dtype=np.uint8)
        rc = L.pmx_predict(dm.handle, pop.handle, theta.ctypes.data, P, pred.ctypes.data, P, status.ctypes.data)
    _ffi.check(rc, allow_pair_failures=not raise_on_pair_failure)
    return pred, status


def predict(model, pop: DevicePopulation, theta, pred=None, status=None, batch: bool = False, want_status: bool = True):
    """Device-pointer ABI form (``pmx_predict_device``): torch CUDA tensors, enqueued on torch's
    current stream, not synchronised.  Returns ``(pred, status)`` tensors."""
    import torch

    L = _ffi.lib()
    dm = _as_model(model)
    dev = torch.device("cuda", pop.device)
    if not (isinstance(theta, torch.Tensor) and theta.is_cuda):
        theta = torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64), device=dev)
    theta = theta.contiguous()
    assert theta.dtype == torch.float64 and theta.dim() == 2 and theta.shape[1] == dm.desc.nparams
    stream = torch.cuda.current_stream(dev).cuda_stream
    if batch:
        assert theta.shape[0] == pop.n_subjects
        if pred is None:
            pred = torch.empty((pop.n_observations,), dtype=torch.float64, device=dev)
        if status is None and want_status:
            status = torch.zeros((pop.n_subjects,), dtype=torch.uint8, device=dev)
        rc = L.pmx_predict_batch_device(dm.handle, pop.handle, theta.data_ptr(), pred.data_ptr(),
                                        status.data_ptr() if status is not None else None, stream)
    else:
        P = int(theta.shape[0])
        if pred is None:
            pred = torch.empty((pop.n_observations, P), dtype=torch.float64, device=dev)
        assert pred.is_contiguous() or pred.stride(1) == 1
        ld = int(pred.stride(0)) if pred.dim() == 2 and pred.shape[0] > 1 else P
        if status is None and want_status:
            status = torch.zeros((pop.n_subjects, P), dtype=torch.uint8, device=dev)
        rc = L.pmx_predict_device(dm.handle, pop.handle, theta.data_ptr(), P, pred.data_ptr(), ld,
                                  status.data_ptr() if status is not None else None, stream)
    _ffi.check(rc)
    return pred, status


def loglik_host(model, flat: FlatPopulation, error_models, theta: np.ndarray, device: int = 0,
                raise_on_pair_failure: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """Host-pointer form of the fused log-likelihood (``pmx_loglik``): ``(ll[S, P], status[S, P])``."""
    L = _ffi.lib()
    dm = _as_model(model)
    pop = DevicePopulation(flat, device)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    if theta.ndim == 1:
        theta = theta.reshape(1, -1)
    P = theta.shape[0]
    em = error_models.to_c(model)
    ll = np.full((pop.n_subjects, P), np.nan)
    status = np.zeros((pop.n_subjects, P), dtype=np.uint8)
    rc = L.pmx_loglik(dm.handle, pop.handle, C.cast(em, C.c_void_p), theta.ctypes.data, P, ll.ctypes.data, P,
                      status.ctypes.data)
    _ffi.check(rc, allow_pair_failures=not raise_on_pair_failure)
    return ll, status


def loglik_batch_host(model, flat: FlatPopulation, error_models, theta: np.ndarray, device: int = 0
                      ) -> Tuple[np.ndarray, np.ndarray]:
    """Host-pointer batch form (``pmx_loglik_batch``): subject s under theta row s; ``(ll[S], status[S])``, failed
    subjects = -inf (likelihood/mod.rs:137-140)."""
    L = _ffi.lib()
    dm = _as_model(model)
    pop = DevicePopulation(flat, device)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    assert theta.shape == (pop.n_subjects, dm.desc.nparams)
    em = error_models.to_c(model)
    ll = np.full((pop.n_subjects,), np.nan)
    status = np.zeros((pop.n_subjects,), dtype=np.uint8)
    _ffi.check(L.pmx_loglik_batch(dm.handle, pop.handle, C.cast(em, C.c_void_p), theta.ctypes.data, ll.ctypes.data,
                                  status.ctypes.data))
    return ll, status


def host_empty(shape, dtype=np.float64) -> np.ndarray:
    """A numpy array in page-locked host memory (``pmx_host_alloc``): outputs of the host-pointer entry points land in
    it by one DMA at link rate instead of going through bounce buffers.  The memory lives as long as the array."""
    import weakref

    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    p = C.c_void_p()
    _ffi.check(_ffi.lib().pmx_host_alloc(n, C.byref(p)))
    buf = (C.c_char * max(n, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    weakref.finalize(buf, _ffi.lib().pmx_host_free, p.value)
    return arr


def loglik(model, pop: DevicePopulation, error_models, theta, ll=None, status=None, want_status: bool = True):
    """Device-pointer form (``pmx_loglik_device``): torch CUDA tensors on torch's current stream, not
    synchronised.  Returns ``(ll[S, P], status[S, P])`` — the matrix ``log_likelihood_matrix`` returns
    (likelihood/matrix.rs:52-106), support point fastest."""
    import torch

    L = _ffi.lib()
    dm = _as_model(model)
    dev = torch.device("cuda", pop.device)
    if not (isinstance(theta, torch.Tensor) and theta.is_cuda):
        theta = torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64), device=dev)
    theta = theta.contiguous()
    P = int(theta.shape[0])
    if ll is None:
        ll = torch.empty((pop.n_subjects, P), dtype=torch.float64, device=dev)
    if status is None and want_status:
        status = torch.zeros((pop.n_subjects, P), dtype=torch.uint8, device=dev)
    em = error_models if not hasattr(error_models, "to_c") else error_models.to_c(model)
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = L.pmx_loglik_device(dm.handle, pop.handle, C.cast(em, C.c_void_p), theta.data_ptr(), P, ll.data_ptr(),
                             int(ll.stride(0)) if ll.shape[0] > 1 else P,
                             status.data_ptr() if status is not None else None, stream)
    _ffi.check(rc)
    return ll, status


def last_kernel_name() -> str:
    return _ffi.lib().pmx_last_kernel_name().decode()


def class_plan(model, flat: FlatPopulation) -> dict:
    """Host-side introspection (``pmx_debug_class_plan``): how the analytical GRID kernels would batch this
    population.  Needs no GPU."""
    md = model.desc() if not isinstance(model, _abi.pmx_model_desc) else model
    pd = flat.desc()
    counts = (C.c_int64 * 5)()
    _ffi.check(_ffi.lib().pmx_debug_class_plan(C.byref(pd), C.byref(md), counts))
    return dict(chunks_exact=int(counts[0]), chunks_loose=int(counts[1]), classed_subjects=int(counts[2]),
                generic_subjects=int(counts[3]), members_per_chunk=int(counts[4]))


def compile_ops(model, flat: FlatPopulation) -> dict:
    """Host-side introspection (``pmx_debug_compile``): the op stream the device would walk, as numpy
    arrays.  Needs no GPU."""
    L = _ffi.lib()
    md = model.desc() if not isinstance(model, _abi.pmx_model_desc) else model
    pd = flat.desc()
    v = _abi.pmx_op_stream_view()
    _ffi.check(L.pmx_debug_compile(C.byref(pd), C.byref(md), C.byref(v)))
    try:
        n, S = int(v.n_ops), int(v.n_subjects)

        def arr(ptr, count, dtype):
            if not ptr or count == 0:
                return np.zeros((0,), dtype=dtype)
            return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)

        meta = arr(v.op_meta, n, np.uint32)
        out = dict(
            n_ops=n, n_subjects=S, n_cov=int(v.n_cov), n_rate=int(v.n_rate), max_input_used=int(v.max_input_used),
            max_outeq=int(v.max_outeq), subj_op_off=arr(v.subj_op_off, S + 1, np.int64), kind=(meta & 0xFF).astype(np.int32),
            io=((meta >> 8) & 0xFFFF).astype(np.int32), flags=(meta >> 24).astype(np.int32),
            a=arr(v.op_a, n, np.float64), b=arr(v.op_b, n, np.float64),
            n=arr(v.op_n, n, np.int32), rate=arr(v.op_rate, n * int(v.n_rate), np.float64).reshape(-1, max(int(v.n_rate), 1)),
            cov=arr(v.op_cov, n * int(v.n_cov), np.float64).reshape(-1, max(int(v.n_cov), 1)),
            subj_order=arr(v.subj_order, S, np.int32))
    finally:
        L.pmx_debug_free(C.byref(v))
    return out

"""PCIe-inclusive rates of the HOST-pointer ABI forms on a resident population (what an NPAG loop calls per cycle):
pmx_loglik (theta H2D + kernel + S x P log-likelihoods D2H) and pmx_predict (+ the whole prediction matrix D2H), with
the output array in page-locked memory (pmx_host_alloc) and in ordinary pageable memory.  `value` of bench.py is
never this number (inputs and outputs there are device-resident)."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np
from pharmsol_amd import AssayErrorModel, AssayErrorModels, ErrorPoly, _abi, _ffi, runtime, synth

L = _ffi.lib()
S, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 1000)
m, flat, theta = synth.config_c3(S, P)
rng = np.random.default_rng(1)
flat.ev_value = flat.ev_value.copy()
flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION] = rng.uniform(0.5, 8.0, int((flat.ev_kind == _abi.PMX_EV_OBSERVATION).sum()))
em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1)).to_c(m)
dm, pop = runtime._as_model(m), runtime.DevicePopulation(flat, 0)
steps = flat.n_events * P

def rate(label, call, out_bytes, reps=5):
    call(); call()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:58s} {dt*1e3:8.2f} ms/call  {steps/dt:.3e} steps/s  {out_bytes/dt/1e9:6.1f} GB/s of output to the host", flush=True)

for pinned in (True, False):
    kind = "page-locked (pmx_host_alloc)" if pinned else "pageable (numpy)"
    ll = runtime.host_empty((S, P)) if pinned else np.empty((S, P))
    rate(f"pmx_loglik  C3 {S}x{P}, ll in {kind}",
         lambda: _ffi.check(L.pmx_loglik(dm.handle, pop.handle, C.cast(em, C.c_void_p), theta.ctypes.data, P, ll.ctypes.data, P, None)),
         ll.nbytes)
    del ll
for pinned in (True, False):
    kind = "page-locked (pmx_host_alloc)" if pinned else "pageable (numpy)"
    n = pop.n_observations
    pred = runtime.host_empty((n, P)) if pinned else np.empty((n, P))
    rate(f"pmx_predict C3 {S}x{P}, pred in {kind}",
         lambda: _ffi.check(L.pmx_predict(dm.handle, pop.handle, theta.ctypes.data, P, pred.ctypes.data, P, None)),
         pred.nbytes, reps=2)
    del pred

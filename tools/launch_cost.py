"""Launch-bound shape (C2: 10k subjects x 1 support point): time per pass when the passes are issued from Python
(runtime.predict), from a C++ loop inside the library (pmx_time_predict_device), and the kernel alone."""
import ctypes as C
import time
import numpy as np
import torch
from pharmsol_amd import _ffi, runtime, synth

S = 10_000
model = synth.model_two_cpt_iv()
theta = synth.theta_c2()
flat = synth.population_c23(S)
pop = runtime.DevicePopulation(flat, 0)
d_theta = torch.as_tensor(np.ascontiguousarray(theta), device="cuda")
pred = torch.empty((pop.n_observations, 1), dtype=torch.float64, device="cuda")
status = torch.zeros((S, 1), dtype=torch.uint8, device="cuda")
for _ in range(200):
    runtime.predict(model, pop, d_theta, pred=pred, status=status)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    runtime.predict(model, pop, d_theta, pred=pred, status=status)
torch.cuda.synchronize()
print("python loop        : %.2f us/pass" % ((time.perf_counter() - t0) / n * 1e6))
ms = C.c_double()
L = _ffi.lib()
dm = runtime._as_model(model)
for reps in (200, 2000):
    _ffi.check(L.pmx_time_predict_device(dm.handle, pop.handle, d_theta.data_ptr(), 1, pred.data_ptr(), 1, reps, None, C.byref(ms)))
    print("C++ loop (%4d)    : %.2f us/pass (device time between events)" % (reps, ms.value * 1e3))
t0 = time.perf_counter()
_ffi.check(L.pmx_time_predict_device(dm.handle, pop.handle, d_theta.data_ptr(), 1, pred.data_ptr(), 1, 2000, None, C.byref(ms)))
print("C++ loop wall      : %.2f us/pass" % ((time.perf_counter() - t0) / 2001 * 1e6))
if hasattr(runtime, "PredictPlan"):
    plan = runtime.PredictPlan(model, pop, d_theta, pred=pred, status=status)
    for _ in range(200):
        plan.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        plan.run()
    torch.cuda.synchronize()
    print("plan (graph) loop  : %.2f us/pass" % ((time.perf_counter() - t0) / n * 1e6))

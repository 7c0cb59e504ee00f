"""One process, one device buffer set: alternate the current build and round 1's build is not possible in-process (one
library per process), so this script times ONE library (PMX_LIB) into (a) a plain torch allocation, (b) a second one,
(c) the placed buffer, several rounds each, to separate box / allocation / time effects."""
import os, sys, time
import numpy as np, torch
from pharmsol_amd import runtime, synth, _ffi
import ctypes as C
m, flat, theta = synth.config_c3(100_000, 1000)
pop = runtime.DevicePopulation(flat, 0)
d_theta = torch.as_tensor(theta, device="cuda")
dm = runtime._as_model(m)
stream = torch.cuda.current_stream().cuda_stream
def t(ptr, reps=20):
    ms = C.c_double()
    _ffi.check(_ffi.lib().pmx_time_predict_device(dm.handle, pop.handle, d_theta.data_ptr(), 1000, ptr, 1000, reps, stream, C.byref(ms)))
    return ms.value
a = torch.empty((pop.n_observations, 1000), dtype=torch.float64, device="cuda")
b = torch.empty((pop.n_observations, 1000), dtype=torch.float64, device="cuda")
for _ in range(5): t(a.data_ptr(), 20)
out = []
for rnd in range(6):
    out.append("A %.4f  B %.4f" % (t(a.data_ptr()), t(b.data_ptr())))
print(os.environ.get("PMX_LIB", "current").split("/")[-1], " | ".join(out), flush=True)

"""Row pitch x allocation: does another leading dimension rescue a slow plain allocation?  (HBM channel
hashing vs the 8000-byte row stride.)  One process; eight plain allocations sized for the widest pitch, all kept alive
(so they differ); the C3 kernel timed into each with ld = 1000 (dense), 1008, 1016, 1024, 1032, 1040, 1056, 1088, 1152;
plus a flat streaming fill of the same bytes into the same allocation (pmx_measure_write_ceiling) as the yardstick.
Writes a table: rows = allocations, columns = pitches, cells = TB/s of algorithmic bytes (5.62 GB)."""
import ctypes as C
import sys

import numpy as np
import torch

from pharmsol_amd import _ffi, runtime, synth

LDS = [1000, 1008, 1016, 1024, 1032, 1040, 1056, 1088, 1152]
m, flat, theta = synth.config_c3(100_000, 1000)
pop = runtime.DevicePopulation(flat, 0)
d_theta = torch.as_tensor(theta, device="cuda")
dm = runtime._as_model(m)
stream = torch.cuda.current_stream().cuda_stream
L = _ffi.lib()
n_obs = pop.n_observations
b_alg = 8 * n_obs * 1000 + 8 * theta.size + 26 * flat.n_events


def t(ptr, ld, reps=10):
    ms = C.c_double()
    _ffi.check(L.pmx_time_predict_device(dm.handle, pop.handle, d_theta.data_ptr(), 1000, ptr, ld, reps, stream, C.byref(ms)))
    return ms.value


bufs = [torch.empty(n_obs * max(LDS), dtype=torch.float64, device="cuda") for _ in range(8)]
for _ in range(3):
    t(bufs[0].data_ptr(), 1000, 20)  # clocks up
print("alloc  address          " + "  ".join("ld=%4d" % ld for ld in LDS) + "   flat fill", flush=True)
for i, b in enumerate(bufs):
    cells = []
    for ld in LDS:
        best = min(t(b.data_ptr(), ld) for _ in range(2))
        cells.append(b_alg / (best * 1e-3) / 1e12)
    g = C.c_double()
    _ffi.check(L.pmx_measure_write_ceiling(b.data_ptr(), n_obs * 1000, 5, stream, C.byref(g)))
    print("%5d  0x%012x  " % (i, b.data_ptr()) + "  ".join("%7.3f" % c for c in cells) + "   %7.3f" % (g.value / 1e3), flush=True)

"""Chunks-per-block sweep of the classed kernel into ONE prediction buffer (the allocation decides the speed class, so
the sweep must not reallocate)."""
import os
import numpy as np, torch
from pharmsol_amd import _ffi, runtime, synth
m, flat, theta = synth.config_c3(100_000, 1000)
pop = runtime.DevicePopulation(flat, 0)
d_theta = torch.as_tensor(theta, device="cuda")
pred = runtime.alloc_predictions(m, pop, d_theta, tries=6)
def t(n=10):
    for _ in range(2): runtime.predict(m, pop, d_theta, pred=pred, want_status=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): runtime.predict(m, pop, d_theta, pred=pred, want_status=False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(2):
    for cpb in (1, 2, 3, 4, 6, 8, 12, 16, 24, 48):
        os.environ["PMX_TUNE_CPB"] = str(cpb)
        _ffi.lib().pmx_debug_reload_env()  # (the library reads its switches once per process)
        print(f"rep{rep} cpb={cpb:3d}  {t():.4f} ms")

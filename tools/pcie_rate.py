"""PCIe-inclusive rate of the host-pointer ABI form (pmx_predict): theta H2D + kernel + pred/status D2H."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from pharmsol_amd import runtime, synth
m, flat, theta = synth.config_c3(20000, 1000)
runtime.predict_host(m, flat, theta)  # warm-up (population compile + upload)
t0 = time.perf_counter()
for _ in range(3):
    pred, st = runtime.predict_host(m, flat, theta)
dt = (time.perf_counter() - t0) / 3
steps = flat.n_events * 1000
print(f"host-pointer form: {dt*1e3:.1f} ms per call for {steps:.3g} steps -> {steps/dt:.3e} steps/s "
      f"({pred.nbytes/1e9:.2f} GB of predictions copied back, incl. population re-upload each call)")

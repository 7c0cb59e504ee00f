"""Covariate-derived rate constants on the smaller structures (ke = ke0 (wt/70)^0.75, v = v0 wt/70; C5's population:
time-varying wt, 13 events per subject): ms per pass of the per-PROP coefficient rebuild."""
import sys
import numpy as np
import torch
from pharmsol_amd import Pow, Ratio, Scaled, analytical, bolus, runtime, synth

S, P = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, 512
flat = synth.population_c5(S)
rng = np.random.default_rng(1)
th = {"ke0": rng.uniform(0.05, 0.4, P), "v0": rng.uniform(10, 60, P), "ka": rng.uniform(1.5, 3.0, P),
      "kcp": rng.uniform(0.1, 0.6, P), "kpc": rng.uniform(0.05, 0.3, P)}
for structure, params, states in (("one_compartment_with_absorption", ["ka", "ke0", "v0"], ["gut", "central"]),
                                  ("two_compartments_with_absorption", ["ke0", "ka", "kcp", "kpc", "v0"], ["gut", "central", "periph"])):
    m = analytical(name="wt_" + structure, params=params,
                   derived={"ke": Scaled("ke0", (Pow("wt", 70.0, 0.75),)), "v": Scaled("v0", (Pow("wt", 70.0, 1.0),))},
                   covariates=["wt"], structure=structure, states=states, outputs=["cp"],
                   routes=[bolus("oral", states[0])], out={"cp": Ratio("central", "v")})
    theta = np.stack([th[p] for p in params], axis=1)
    pop = runtime.DevicePopulation(flat, 0)
    d_theta = torch.as_tensor(np.ascontiguousarray(theta), device="cuda")
    pred = torch.empty((pop.n_observations, P), dtype=torch.float64, device="cuda")
    for _ in range(3):
        runtime.predict(m, pop, d_theta, pred=pred)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        runtime.predict(m, pop, d_theta, pred=pred)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{structure:36s} S={S} P={P}  {ms:7.3f} ms  {pop.n_events * P / ms / 1e6:8.1f} G steps/s  {runtime.last_kernel_name()}")

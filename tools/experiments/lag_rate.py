"""Oral one-compartment model with and without a lag time (and with bioavailability) on a C5-shaped dosing design
(3 boluses q24h + 10 samples, 100k subjects x 512 support points, no covariates): ms per pass and the kernel taken."""
import numpy as np
import torch
from pharmsol_amd import Analytical, Ratio, runtime, synth

S, P = 100_000, 512
base = synth.population_c5(S)
from pharmsol_amd.flatten import FlatPopulation
flat = FlatPopulation(subj_occ_off=base.subj_occ_off, occ_ev_off=base.occ_ev_off, occ_index=base.occ_index, ev_time=base.ev_time,
                      ev_value=base.ev_value, ev_duration=base.ev_duration, ev_kind=base.ev_kind, ev_io=base.ev_io, presorted=False)
rng = np.random.default_rng(2)
theta = np.stack([rng.uniform(1.0, 3.0, P), rng.uniform(0.05, 0.4, P), rng.uniform(10, 60, P), rng.uniform(0.0, 2.0, P),
                  rng.uniform(0.4, 1.0, P)], 1)
cases = {
    "plain": Analytical.new("one_compartment_with_absorption", {0: Ratio(1, 2)}, nparams=5),
    "fa": Analytical.new("one_compartment_with_absorption", {0: Ratio(1, 2)}, nparams=5, fa={0: 4}),
    "lag": Analytical.new("one_compartment_with_absorption", {0: Ratio(1, 2)}, nparams=5, lag={0: 3}),
    "lag+fa": Analytical.new("one_compartment_with_absorption", {0: Ratio(1, 2)}, nparams=5, lag={0: 3}, fa={0: 4}),
}
for name, m in cases.items():
    m = m.with_nstates(2).with_ndrugs(1).with_nout(1)
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
    print(f"{name:8s} {ms:7.3f} ms  {pop.n_events * P / ms / 1e6:8.1f} G steps/s  {runtime.last_kernel_name()}")

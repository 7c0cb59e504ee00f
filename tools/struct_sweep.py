"""C3-shaped population (100k subjects x 1000 support points, shared design) through every closed-form structure:
ms per pass and achieved write bandwidth (the prediction stream is the same 5.6 GB for all of them)."""
import sys
import numpy as np
import torch
from pharmsol_amd import Analytical, Ratio, runtime, synth

S, P = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, 1000
flat = synth.population_c23(S)
rng = np.random.default_rng(0)
th3 = synth.theta_c5(P)  # ka, k10, k12, k13, k21, k31, v with real eigenvalues
th2 = synth.theta_c3(P)  # ke, kcp, kpc, v
cases = {
    "one_compartment": (np.stack([th2[:, 0], th2[:, 3]], 1), 0),
    "one_compartment_with_absorption": (np.stack([th3[:, 0], th2[:, 0], th2[:, 3]], 1), 1),
    "two_compartments": (th2, 0),
    "two_compartments_with_absorption": (np.concatenate([th2[:, :1], th3[:, :1], th2[:, 1:]], 1), 1),
    "three_compartments": (th3[:, 1:], 0),
    "three_compartments_with_absorption": (th3, 1),
}
for name, (theta, central) in cases.items():
    k = theta.shape[1]
    m = Analytical.new(name, {0: Ratio(central, k - 1)}, nparams=k).with_nstates(
        {"one_compartment": 1, "one_compartment_with_absorption": 2, "two_compartments": 2,
         "two_compartments_with_absorption": 3, "three_compartments": 3, "three_compartments_with_absorption": 4}[name]
    ).with_ndrugs(2).with_nout(1)
    # the infusion goes to input 0; for absorption models that is the depot -> rate into rateiv[0] (central)
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
    print(f"{name:38s} {ms:7.3f} ms  {pop.n_observations * P * 8 / ms / 1e9:6.2f} TB/s  {S * 8 * P / ms / 1e9:8.1f} Gsteps/s  {runtime.last_kernel_name()}")
    del pred, pop

"""C4 population (50k irregular subjects, one theta each) through the adaptive Dormand-Prince 5(4) solver at a few
tolerances, next to the fixed-step RK4 pass: ms per pass and max rel err against the closed form."""
import numpy as np
import torch
import oracle
from pharmsol_amd import Analytical, Ratio, runtime, synth

model, flat, theta = synth.config_c4(50_000)
ma = Analytical.new("one_compartment", {0: Ratio(0, 1)}, nparams=2).with_nstates(1).with_ndrugs(1).with_nout(1)
exact, _ = oracle.predict_batch(ma, flat, theta)
pop = runtime.DevicePopulation(flat, 0)
d_theta = torch.as_tensor(np.ascontiguousarray(theta), device="cuda")
cases = [("rk4 h<=0.02", model)]
for tol in (1e-4, 1e-6, 1e-8):
    cases.append((f"dopri5 rtol=atol={tol:g}", synth.model_one_cmt_iv_ode(0.02).with_step(4.0).with_solver("dopri5").with_tolerances(tol, tol)))
for name, m in cases:
    pred = torch.empty((pop.n_observations,), dtype=torch.float64, device="cuda")
    for _ in range(3):
        runtime.predict(m, pop, d_theta, pred=pred, batch=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        runtime.predict(m, pop, d_theta, pred=pred, batch=True)
    e1.record()
    torch.cuda.synchronize()
    got = pred.cpu().numpy()
    scale = np.maximum(np.abs(exact), 1e-6 * np.abs(exact).max())
    print(f"{name:28s} {e0.elapsed_time(e1) / 10:7.3f} ms   max rel err vs closed form {np.max(np.abs(got - exact) / scale):.2e}   {runtime.last_kernel_name()}")

"""ODE GRID kernel (lane = support point, wave-uniform ops): RK4 steps per second on the C4 population walked with a
shared support grid, against the chip's FP64 issue rate (39.3e12 lane-instructions/s / instructions per step)."""
import sys
import numpy as np
import torch
from pharmsol_amd import runtime, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for P in (64, 256, 1024):
    model, flat, theta_all = synth.config_c4(S)
    theta = theta_all[:P] if P <= S else np.tile(theta_all, (P // S + 1, 1))[:P]
    pop = runtime.DevicePopulation(flat, 0)
    d_theta = torch.as_tensor(np.ascontiguousarray(theta), device="cuda")
    pred = torch.empty((pop.n_observations, P), dtype=torch.float64, device="cuda")
    for _ in range(2):
        runtime.predict(model, pop, d_theta, pred=pred)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        runtime.predict(model, pop, d_theta, pred=pred)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    t, off = flat.ev_time, flat.occ_ev_off
    steps = 0
    for s in range(S):
        d = np.diff(np.sort(t[off[s]:off[s + 1]]))
        steps += np.ceil(d[d > 0] / 0.02).sum()
    print(f"S={S} P={P:5d}  {ms:8.3f} ms  {steps * P / ms / 1e9:8.1f} G RK4 steps/s  {runtime.last_kernel_name()}")

"""Pass time with and without the status array, into one (tuned) prediction buffer."""
import numpy as np, torch
from pharmsol_amd import runtime, synth
m, flat, theta = synth.config_c3(100_000, 1000)
pop = runtime.DevicePopulation(flat, 0)
d_theta = torch.as_tensor(theta, device="cuda")
log = []
pred = runtime.alloc_predictions(m, pop, d_theta, tries=6, log=log)
print("candidates:", ", ".join(f"{ms:.3f}" for _, ms in log))
status = torch.zeros((pop.n_subjects, 1000), dtype=torch.uint8, device="cuda")
def t(n, **kw):
    for _ in range(2): runtime.predict(m, pop, d_theta, pred=pred, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): runtime.predict(m, pop, d_theta, pred=pred, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(3):
    print(f"rep{rep}: no status {t(20, want_status=False):.4f} ms   with status {t(20, status=status):.4f} ms")

"""Whole-population parity of the PAIR kernels at the BASELINE sizes (bench.py checks a sample): C4 50k subjects and
C2 10k subjects against the CPU oracle, every prediction and every status byte."""
import numpy as np
import torch
import oracle
from pharmsol_amd import runtime, synth

for name, (model, flat, theta), batch in (("C4", synth.config_c4(50_000), True), ("C2", synth.config_c2(10_000), False),
                                          ("C4 x 400k", synth.config_c4(400_000), True)):
    pop = runtime.DevicePopulation(flat, 0)
    pred, st = runtime.predict(model, pop, np.ascontiguousarray(theta), batch=batch)
    torch.cuda.synchronize()
    got, st = pred.cpu().numpy(), st.cpu().numpy()
    want, wst = (oracle.predict_batch if batch else oracle.predict)(model, flat, theta)
    assert got.shape == want.shape and (st == wst).all()
    ok = np.isfinite(want)
    assert (np.isfinite(got) == ok).all()
    scale = np.maximum(np.abs(want[ok]), 1e-12 * np.abs(want[ok]).max())
    print(f"{name}: {got.size} predictions, max rel err {(np.abs(got[ok] - want[ok]) / scale).max():.3e}, kernel {runtime.last_kernel_name()}")

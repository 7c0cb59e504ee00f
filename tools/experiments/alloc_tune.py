"""Kernel time of the C3 bench pass as a function of WHICH allocation holds the prediction matrix, for consecutive
and spread chunk membership.  (The store-pattern probe shows two speeds of the same scattered write pattern depending
on the allocation; this is the real kernel.)"""
import os, sys
import numpy as np, torch
from pharmsol_amd import _ffi, runtime, synth

S, P = 100_000, 1000
m, flat, theta = synth.config_c3(S, P)
d_theta = torch.as_tensor(theta, device="cuda")
pops = {}
for sp in (0, 1):
    os.environ["PMX_TUNE_SPREAD"] = str(sp)
    _ffi.lib().pmx_debug_reload_env()  # (the library reads its switches once per process)
    pops[sp] = runtime.DevicePopulation(flat, 0)
    tmp = torch.empty((pops[sp].n_observations, 8), dtype=torch.float64, device="cuda")
    runtime.predict(m, pops[sp], d_theta[:8].contiguous(), pred=tmp)  # builds the class plan under this setting
    torch.cuda.synchronize()
    del tmp

def time_it(pop, pred, n=10):
    for _ in range(2):
        runtime.predict(m, pop, d_theta, pred=pred, want_status=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        runtime.predict(m, pop, d_theta, pred=pred, want_status=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

bufs = []
for k in range(6):
    pred = torch.empty((pops[0].n_observations, P), dtype=torch.float64, device="cuda")
    bufs.append(pred)  # keep every candidate alive: each one sits on different memory
    t0, t1 = time_it(pops[0], pred), time_it(pops[1], pred)
    print(f"allocation {k} @ {pred.data_ptr():#x}: consecutive {t0:.4f} ms   spread {t1:.4f} ms")

#!/usr/bin/env python3
"""Where the fused log-likelihood kernel's wave time goes (diagnostic build, -DPMX_LL_STAMPS: s_memtime stamps around
the phases of a chunk, summed over waves).  usage on the GPU box:
   tools/ab_variant.sh ll_stamps "-DPMX_LL_STAMPS" ; PMX_LIB=$PWD/pharmsol_amd/lib/ab/ll_stamps.so python tools/ll_stamps.py"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pharmsol_amd import AssayErrorModel, AssayErrorModels, ErrorPoly, _abi, _ffi, runtime, synth  # noqa: E402

m, flat, theta = synth.config_c3(100_000, 1000)
pop0 = runtime.DevicePopulation(flat, 0)
p0, _ = runtime.predict(m, pop0, theta[:1])
torch.cuda.synchronize()
vals = np.abs(p0.cpu().numpy()[:, 0]) * np.exp(0.2 * (synth.SplitMix64(1).uniform(pop0.n_observations) - 0.5)) + 0.05
flat.ev_value = flat.ev_value.copy()
flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION] = vals
em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
pop = runtime.DevicePopulation(flat, 0)
th = torch.as_tensor(theta, device="cuda")
ll = torch.empty((pop.n_subjects, 1000), dtype=torch.float64, device="cuda")
emc = em.to_c(m)
for _ in range(30):
    runtime.loglik(m, pop, emc, th, ll=ll, want_status=False)
torch.cuda.synchronize()
L = _ffi.lib()
L.pmx_debug_ll_stamps.argtypes = [C.c_void_p, C.c_int32]
out = (C.c_uint64 * 5)()
L.pmx_debug_ll_stamps(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 10
e0.record()
for _ in range(K):
    runtime.loglik(m, pop, emc, th, ll=ll, want_status=False)
e1.record()
torch.cuda.synchronize()
L.pmx_debug_ll_stamps(out, 0)
t = [int(x) / K for x in out]
names = ["chunk header", "slow steps", "fast runs", "epilogue (sums, stores, status)", "whole wave"]
print("kernel", runtime.last_kernel_name(), "ms/pass", e0.elapsed_time(e1) / K)
for n, v in zip(names, t):
    print(f"{n:34s} {v:16.0f} cycles summed over waves  ({100 * v / t[4]:5.1f} % of wave time)")

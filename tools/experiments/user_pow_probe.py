"""Is the user-closure workload's plain-vs-traced gap tied to the device libm's pow()?  The same model with pow() and with
exp2(e * log2(x)) (no table-driven routine), timed with pmx_time_predict_device; run plain and under rocprofv3 --kernel-trace."""
import ctypes as C

import torch

from pharmsol_amd import _ffi, analytical, bolus, infusion, runtime, synth

SRC_POW = synth.USER_COVARIATE_SRC
SRC_EXP = SRC_POW.replace("pow(wt / 70.0, 0.75)", "exp2(0.75 * log2(wt / 70.0))").replace(
    "pow(renal / 90.0, 0.25)", "exp2(0.25 * log2(renal / 90.0))").replace(
    "pow(90.0 / cov[COV_renal], 0.1)", "exp2(0.1 * log2(90.0 / cov[COV_renal]))").replace(
    "pow(cov[COV_renal] / 90.0, 0.1)", "exp2(0.1 * log2(cov[COV_renal] / 90.0))")
SRC_NONE = SRC_POW.replace("pow(wt / 70.0, 0.75)", "(wt / 70.0)").replace("pow(renal / 90.0, 0.25)", "(renal / 90.0)").replace(
    "pow(90.0 / cov[COV_renal], 0.1)", "(90.0 / cov[COV_renal])").replace("pow(cov[COV_renal] / 90.0, 0.1)", "(cov[COV_renal] / 90.0)")
flat = synth.population_user(50_000)
theta = torch.as_tensor(synth.theta_user(256), device="cuda")
L = _ffi.lib()
for name, src in (("pow", SRC_POW), ("exp2*log2", SRC_EXP), ("no transcendental", SRC_NONE), ("pow again", SRC_POW)):
    m = analytical(name="one_cmt_abs_covariates", params=["ka", "ke0", "v", "tlag", "f_oral", "base_gut", "base_central"],
                   derived=["ke", "adjusted_v"], covariates=["wt", "renal"], states=["gut", "central"], outputs=["cp"],
                   routes=[bolus("oral", "gut"), infusion("iv", "central")], structure="one_compartment_with_absorption", source=src)
    pop = runtime.DevicePopulation(m.flatten_flat(flat) if hasattr(m, "flatten_flat") else flat, 0)
    pred = torch.empty((pop.n_observations, 256), dtype=torch.float64, device="cuda")
    dm = runtime._as_model(m)
    ms = C.c_double()
    for _ in range(2):
        _ffi.check(L.pmx_time_predict_device(dm.handle, pop.handle, theta.data_ptr(), 256, pred.data_ptr(), 256, 20,
                                             torch.cuda.current_stream().cuda_stream, C.byref(ms)))
    print("%-20s %.3f ms" % (name, ms.value), flush=True)

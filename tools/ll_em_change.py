"""What does changing the error model between log-likelihood calls cost?  C3 population, same theta, 20 calls with
20 different lambdas vs 20 calls with one."""
import numpy as np, torch, time
from pharmsol_amd import AssayErrorModel, AssayErrorModels, ErrorPoly, _abi, runtime, synth
m, flat, theta = synth.config_c3(100_000, 1000)
rng = np.random.default_rng(0)
flat.ev_value = flat.ev_value.copy()
flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION] = rng.uniform(1, 9, flat.n_observations)
pop = runtime.DevicePopulation(flat, 0)
d_theta = torch.as_tensor(theta, device="cuda")
ll = torch.empty((pop.n_subjects, 1000), dtype=torch.float64, device="cuda")
def em(l): return AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), l)).to_c(m)
same = [em(0.1)] * 20
diff = [em(0.1 + 0.01 * i) for i in range(20)]
for name, ems in (("same error model", same), ("new error model every call", diff), ("same again", same)):
    runtime.loglik(m, pop, ems[0], d_theta, ll=ll); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in ems:
        runtime.loglik(m, pop, e, d_theta, ll=ll)
    torch.cuda.synchronize()
    print(f"{name:30s} {(time.perf_counter() - t0) / len(ems) * 1e3:8.3f} ms per call")

"""Debug helper: re-run the log-likelihood branch of one analytical fuzz case on the GPU and show where it differs from the oracle.
usage: python tests/dbg_fuzz_ll.py <seed of tests/test_gpu_fuzz.py::test_random_analytical_configuration>"""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

import oracle
from pharmsol_amd import AssayErrorModel, AssayErrorModels, Data, ErrorPoly, _abi, runtime
from tests.test_gpu_fuzz import build_case
from tests.test_gpu_likelihood import _censor_some

seed = int(sys.argv[1])
m, subs, theta, batch, recipe = build_case(1000 + seed)
flat = m.flatten(Data(subs))
want, _ = oracle.predict(m, flat, theta)
rng = np.random.default_rng(seed)
vals = np.abs(np.where(np.isfinite(want[:, 0]), want[:, 0], 1.0)) * np.exp(rng.normal(0, 0.2, want.shape[0])) + 0.05
vals[rng.random(vals.shape) < 0.2] = np.nan
flat.ev_value = flat.ev_value.copy()
flat.ev_value[flat.ev_kind == _abi.PMX_EV_OBSERVATION] = vals
if seed % 6 == 0:
    flat = _censor_some(flat, rng)
em = AssayErrorModels.empty().add(0, AssayErrorModel.additive(ErrorPoly(0.05, 0.1, 0.0, 0.0), 0.1))
pop = runtime.DevicePopulation(flat, 0)
ll, lst = runtime.loglik(m, pop, em, np.ascontiguousarray(theta))
torch.cuda.synchronize()
print("kernel", runtime.last_kernel_name(), recipe)
wll, wlst = oracle.loglik(m, flat, em, theta)
gl = ll.cpu().numpy()
ok = np.isfinite(wll)
err = np.abs(gl - wll) / np.maximum(np.abs(wll), 1.0)
err[~ok] = 0
i = np.unravel_index(np.argmax(err), err.shape)
print("max err", err.max(), "at (subject, support)", i, "gpu", gl[i], "oracle", wll[i])
bad = np.argwhere(err > 1e-9)
print("n > 1e-9:", len(bad), "subjects", sorted(set(bad[:, 0]))[:10], "cols", sorted(set(bad[:, 1]))[:10])
s = int(i[0])
off = pop.observation_offsets()
is_obs = flat.ev_kind == _abi.PMX_EV_OBSERVATION
ev0 = flat.occ_ev_off[flat.subj_occ_off[s]]; ev1 = flat.occ_ev_off[flat.subj_occ_off[s + 1]]
print("subject events:")
for e in range(ev0, ev1):
    print("   t", flat.ev_time[e], "kind", flat.ev_kind[e], "val", flat.ev_value[e], "cens", flat.ev_censor[e] if flat.ev_censor is not None else None,
          "poly", flat.ev_errorpoly[e] if flat.ev_errorpoly is not None else None)
pred, _ = runtime.predict(m, pop, np.ascontiguousarray(theta)); torch.cuda.synchronize()
p = pred.cpu().numpy()
print("theta", theta[i[1]])
print("gpu pred ", p[off[s]:off[s + 1], i[1]])
print("orac pred", want[off[s]:off[s + 1], i[1]])

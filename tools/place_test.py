import numpy as np, torch, time
from pharmsol_amd import runtime, synth
import oracle
m, flat, theta = synth.config_c3(100_000, 1000)
pop = runtime.DevicePopulation(flat, 0)
d_theta = torch.as_tensor(theta, device="cuda")
t0 = time.perf_counter()
pred = runtime.place_predictions(m, pop, d_theta, search_gib=48)
torch.cuda.synchronize()
print("placed in %.2f s, window pass %.4f ms, ptr %#x" % (time.perf_counter() - t0, pred._pmx_owner.ms_per_pass, pred.data_ptr()))
def t(n=20, **kw):
    for _ in range(40): runtime.predict(m, pop, d_theta, pred=pred, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): runtime.predict(m, pop, d_theta, pred=pred, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
st = torch.zeros((pop.n_subjects, 1000), dtype=torch.uint8, device="cuda")
print("no status %.4f ms, with status %.4f ms" % (t(want_status=False), t(status=st)))
small = synth.config_c3(50, 1000)
got = pred[:350].cpu().numpy()
want, _ = oracle.predict(small[0], small[1], theta)
print("max rel err on the first 50 subjects:", float(np.abs(got / want - 1).max()))
free0 = torch.cuda.mem_get_info()[0] / 2**30
del pred
import gc; gc.collect()
print("free GiB before/after release: %.1f / %.1f" % (free0, torch.cuda.mem_get_info()[0] / 2**30))

import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
import oracle
from pharmsol_amd import runtime, synth
import __graft_entry__ as g
g.smoke()
def check(name, model, flat, theta, batch=False, tol=1e-6):
    pop = runtime.DevicePopulation(flat, 0)
    pred, st = runtime.predict(model, pop, theta, batch=batch)
    torch.cuda.synchronize()
    got = pred.cpu().numpy()
    if batch: want, ws = oracle.predict_batch(model, flat, theta)
    else: want, ws = oracle.predict(model, flat, theta)
    rel = np.abs(got-want)/np.maximum(np.abs(want),1e-12)
    print(name, runtime.last_kernel_name(), got.shape, "max rel", rel.max(), "status eq", (st.cpu().numpy()==ws).all(), flush=True)
m,f,t = synth.config_c2(1000); check("c2", m,f,t)
m,f,t = synth.config_c3(300, 1000); check("c3", m,f,t)
m,f,t = synth.config_c4(2000); check("c4", m,f,t, batch=True)
m,f,t = synth.config_c5(200, 512); check("c5", m,f,t)
m,f,t = synth.config_c5(20, 8); check("c5pair", m,f,t)
# timing C3 full
m,f,t = synth.config_c3(100000, 1000)
pop = runtime.DevicePopulation(f,0)
th = torch.as_tensor(t, device='cuda')
pred = torch.empty((pop.n_observations, 1000), dtype=torch.float64, device='cuda')
st = torch.zeros((pop.n_subjects,1000), dtype=torch.uint8, device='cuda')
for i in range(3): runtime.predict(m, pop, th, pred=pred, status=st)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10): runtime.predict(m, pop, th, pred=pred, status=st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/10
steps = pop.n_events*1000
print("C3 ms/pass", ms, "steps/s", steps/ms*1e3, "GB/s alg", (8*pop.n_observations*1000 + 8*1000*4 + 26*pop.n_events)/ms/1e6)

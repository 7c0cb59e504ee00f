"""Debug helper: re-run one analytical fuzz case (tests/test_gpu_fuzz.py build_case(seed)) on the GPU and show the rows /
support points where it differs from the oracle.  usage: python tests/dbg_fuzz_case.py <case seed = 1000 + test seed>"""
import sys; sys.path.insert(0,'.')
import numpy as np, torch
import oracle
from pharmsol_amd import Data, runtime
from tests.test_gpu_fuzz import build_case
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2235
m, subs, theta, batch, recipe = build_case(seed)
flat = m.flatten(Data(subs))
pop = runtime.DevicePopulation(flat, 0)
pred, st = runtime.predict(m, pop, np.ascontiguousarray(theta))
torch.cuda.synchronize()
got = pred.cpu().numpy(); want, wst = oracle.predict(m, flat, theta)
scale = np.maximum(np.abs(want), 1e-9*np.nanmax(np.abs(want)))
err = np.abs(got-want)/scale
bad = np.argwhere(err > 1e-6)
print("kernel", runtime.last_kernel_name(), "n bad", len(bad), "of", err.size)
off = pop.observation_offsets()
rows = sorted(set(bad[:,0])); cols = sorted(set(bad[:,1]))
print("bad rows", rows[:20]); print("bad cols", cols[:20], len(cols))
d = m.desc()
for c in cols[:6]:
    print("col", c, "theta", theta[c])
subj = np.searchsorted(off, rows, side='right')-1
print("subjects", sorted(set(subj)))
for s in sorted(set(subj))[:3]:
    print("subject", s, "obs rows", off[s], off[s+1])
    for o in subs[s].occasions:
        print("  occ", o.index, [(type(e).__name__, round(e.time, 6), getattr(e, "amount", None), getattr(e, "duration", None)) for e in o.events])
    r0=off[s]
    for c in cols[:2]:
        print("  got ", got[off[s]:off[s+1], c]); print("  want", want[off[s]:off[s+1], c])

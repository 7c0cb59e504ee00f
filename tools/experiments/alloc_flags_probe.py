"""Does the KIND of allocation decide the write rate of the row-strided prediction stream?  C3 pass time into buffers
from hipMalloc, hipExtMallocWithFlags(Contiguous | Uncached | Finegrained) and the library's placed buffer."""
import ctypes as C
import numpy as np
import torch
from pharmsol_amd import _ffi, runtime, synth

hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]

S, P = 100_000, 1000
model = synth.model_two_cpt_iv()
theta = synth.theta_c3(P)
flat = synth.population_c23(S)
pop = runtime.DevicePopulation(flat, 0)
d_theta = torch.as_tensor(np.ascontiguousarray(theta), device="cuda")
L = _ffi.lib()
dm = runtime._as_model(model)
nbytes = pop.n_observations * P * 8
ms = C.c_double()


def time_into(ptr, reps=20):
    _ffi.check(L.pmx_time_predict_device(dm.handle, pop.handle, d_theta.data_ptr(), P, ptr, P, 30, None, C.byref(ms)))
    _ffi.check(L.pmx_time_predict_device(dm.handle, pop.handle, d_theta.data_ptr(), P, ptr, P, reps, None, C.byref(ms)))
    return ms.value


FLAGS = {"hipMalloc": None, "Contiguous": 0x4, "Uncached": 0x3, "Finegrained": 0x1}
for round_ in range(3):
    held = []
    for name, fl in FLAGS.items():
        p = C.c_void_p()
        rc = hip.hipMalloc(C.byref(p), nbytes) if fl is None else hip.hipExtMallocWithFlags(C.byref(p), nbytes, fl)
        if rc != 0:
            print(f"round {round_} {name:12s} allocation failed rc={rc}")
            continue
        t = time_into(p.value)
        print(f"round {round_} {name:12s} ptr=0x{p.value:x}  {t:.4f} ms  {nbytes / t / 1e9:.2f} TB/s", flush=True)
        held.append(p)  # keep it: the next allocation lands elsewhere
    for p in held:
        hip.hipFree(p)
pred = runtime.place_predictions(model, pop, d_theta, search_gib=48)
print(f"placed buffer            {time_into(pred.data_ptr()):.4f} ms")

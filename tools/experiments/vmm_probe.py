"""Placement experiments (DESIGN.md "Where the matrix lives"): C3 pass time of the real kernel into buffers built
with a chosen virtual alignment / physical chunk size / mapping order (tools/experiments/vmm_probe.hip), next to plain hipMalloc.
Question: is the two-speed behaviour of the row-strided write stream a matter of page-table fragments (TLB reach:
amdgpu marks 2^k physically contiguous, equally aligned pages as one translation)?"""
import ctypes as C, os, sys
import numpy as np, torch
from pharmsol_amd import _ffi, runtime, synth

here = os.path.dirname(os.path.abspath(__file__))
V = C.CDLL(os.path.join(here, "bin", "libvmmprobe.so"))
V.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
V.vmm_free.argtypes = [C.c_void_p]
V.vmm_granularity.restype = C.c_size_t
MiB, GiB = 1 << 20, 1 << 30
S, P = 100_000, 1000
m, flat, theta = synth.config_c3(S, P)
pop = runtime.DevicePopulation(flat, 0)
d_theta = torch.as_tensor(theta, device="cuda")
dm = runtime._as_model(m)
need = pop.n_observations * P * 8
stream = torch.cuda.current_stream().cuda_stream
print("granularity min/recommended:", V.vmm_granularity(0), V.vmm_granularity(1), " matrix bytes:", need, flush=True)

def time_ptr(ptr, reps=10):
    ms = C.c_double()
    _ffi.check(_ffi.lib().pmx_time_predict_device(dm.handle, pop.handle, d_theta.data_ptr(), P, ptr, P, reps, stream, C.byref(ms)))
    return ms.value

# clocks up
warm = torch.empty((pop.n_observations, P), dtype=torch.float64, device="cuda")
for _ in range(3):
    time_ptr(warm.data_ptr(), 20)
print("torch.empty (first)      %.4f ms  ptr=%x" % (time_ptr(warm.data_ptr()), warm.data_ptr()), flush=True)

def vmm(chunk, align, shift=0, order=0, seed=1, label=""):
    out = C.c_void_p()
    rc = V.vmm_alloc(need, chunk, align, shift, order, seed, C.byref(out))
    if rc != 0:
        print(f"{label:42s} alloc failed rc={rc}", flush=True)
        return
    t = [time_ptr(out.value) for _ in range(2)]
    print(f"{label:42s} {min(t):.4f} ms  ptr={out.value:x}", flush=True)
    V.vmm_free(out)

for rnd in range(3):
    print(f"--- round {rnd}", flush=True)
    vmm(2 * MiB, 2 * MiB, label="chunk 2M   align 2M   in order")
    vmm(2 * MiB, 1 * GiB, label="chunk 2M   align 1G   in order")
    vmm(2 * MiB, 1 * GiB, order=2, seed=rnd + 1, label="chunk 2M   align 1G   scrambled")
    vmm(64 * MiB, 1 * GiB, label="chunk 64M  align 1G   in order")
    vmm(64 * MiB, 1 * GiB, order=2, seed=rnd + 1, label="chunk 64M  align 1G   scrambled")
    vmm(1 * GiB, 1 * GiB, label="chunk 1G   align 1G   in order")
    vmm(1 * GiB, 1 * GiB, order=1, label="chunk 1G   align 1G   reversed")
    vmm(1 * GiB, 1 * GiB, shift=2 * MiB, label="chunk 1G   align 1G   VA shifted 2M")
    vmm(1 * GiB, 2 * MiB, label="chunk 1G   align 2M   in order")
    vmm(6 * GiB, 1 * GiB, label="chunk 6G (one piece) align 1G")
    vmm(6 * GiB, 8 * GiB, label="chunk 6G (one piece) align 8G")
    x = torch.empty((pop.n_observations, P), dtype=torch.float64, device="cuda")
    print("torch.empty (fresh)      %.4f ms  ptr=%x" % (time_ptr(x.data_ptr()), x.data_ptr()), flush=True)
    del x
    torch.cuda.empty_cache()
    # dirty the physical memory: odd-sized allocations freed in an interleaved order
    junk = [torch.empty(int(np.random.default_rng(rnd * 100 + i).integers(50, 900)) * MiB, dtype=torch.uint8, device="cuda") for i in range(40)]
    del junk[::2]
    torch.cuda.empty_cache()
    vmm(1 * GiB, 1 * GiB, label="(fragmented) chunk 1G align 1G")
    vmm(2 * MiB, 1 * GiB, label="(fragmented) chunk 2M align 1G")
    del junk
    torch.cuda.empty_cache()

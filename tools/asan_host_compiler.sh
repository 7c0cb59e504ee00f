#!/bin/bash
# The host population compiler (pharmsol_amd/csrc/pmx_compile.cpp: validation, Occasion::sort, sub-segment splitting,
# lag lists, class detection) under AddressSanitizer + UBSan on the CPU (no GPU needed; GPU sanitizers are not available
# on the pool): a variant libpmx_hip.so whose pmx_compile.o is instrumented, driven through pmx_debug_compile /
# pmx_debug_class_plan by the CPU host-logic tests and by the fuzz suite's random populations.
# usage: tools/asan_host_compiler.sh [n fuzz populations, default 1500]     (run `make` first)
set -e
N=${1:-1500}
T=$(mktemp -d)
g++ -O1 -g -std=c++17 -fPIC -Wall -ffp-contract=off -Iinclude -fsanitize=address,undefined -fno-omit-frame-pointer \
    -c pharmsol_amd/csrc/pmx_compile.cpp -o $T/pmx_compile.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $T/libpmx_hip_asan.so $T/pmx_compile.o \
    pharmsol_amd/csrc/build/pmx_api.o pharmsol_amd/csrc/build/pmx_kernels.o pharmsol_amd/csrc/build/pmx_jit.o \
    pharmsol_amd/csrc/build/pmx_alloc.o -L/opt/rocm/lib -lhiprtc
export LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 PMX_LIB=$T/libpmx_hip_asan.so
python -m pytest tests/test_host_logic.py tests/test_data_model.py tests/test_pmetrics.py tests/test_api_surface.py -x -q -m "not gpu" -k "not isa"
python - "$N" <<'PY'
import sys
sys.path.insert(0, ".")
from pharmsol_amd import Data, _ffi, runtime, synth
import tests.test_gpu_fuzz as F
print("library under test:", _ffi.LIB_PATH)
for seed in range(int(sys.argv[1])):
    m, subs, theta, batch, recipe = F.build_case(1000 + seed)
    flat = m.flatten(Data(subs))
    runtime.compile_ops(m, flat)
    runtime.class_plan(m, flat)
for m, flat, _ in (synth.config_c3(5000, 8), synth.config_c5(3000, 8)):
    runtime.compile_ops(m, flat)
    runtime.class_plan(m, flat)
runtime.compile_ops(synth.model_user_covariates(), synth.population_user(2000))
m, fl = synth.model_two_cpt_iv(), synth.population_c23(4000, ragged=True)
runtime.compile_ops(m, fl)
runtime.class_plan(m, fl)
print("clean:", sys.argv[1], "fuzz populations + the synthetic configurations")
PY
rm -rf $T

#!/bin/bash
# LL kernel experiments on ONE box: chunks per block (persistent-style grids), store ablation (timing only)
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --loglik --no-cpu-baseline > gpurun_out/r03_ll2_$name.json 2> gpurun_out/r03_ll2_$name.err; }
L3=$PWD/pharmsol_amd/lib/ab/llw3.so
run w3 PMX_LIB=$L3
run w3_cpb66 PMX_LIB=$L3 PMX_TUNE_CPB=66
run w3_cpb33 PMX_LIB=$L3 PMX_TUNE_CPB=33
run w3_cpb16 PMX_LIB=$L3 PMX_TUNE_CPB=16
run w4_cpb49 PMX_TUNE_CPB=49
run w3_nostore PMX_LIB=$PWD/pharmsol_amd/lib/ab/llw3_nostore.so
python - <<'PY'
import json
for f in ("w3", "w3_cpb66", "w3_cpb33", "w3_cpb16", "w4_cpb49", "w3_nostore"):
    try:
        d = json.load(open("gpurun_out/r03_ll2_%s.json" % f))
        print(f, d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"], 4), d["max_rel_err_vs_cpu_ref"], d["parity_ok"])
    except Exception as e:
        print(f, "FAILED", e)
PY

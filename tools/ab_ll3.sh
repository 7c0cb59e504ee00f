#!/bin/bash
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --loglik --no-cpu-baseline > gpurun_out/r03_ll3_$name.json 2> gpurun_out/r03_ll3_$name.err; }
run base PMX_X=1
run stag1 PMX_LIB=$PWD/pharmsol_amd/lib/ab/ll_stag1.so
run stag2 PMX_LIB=$PWD/pharmsol_amd/lib/ab/ll_stag2.so
run stag2_cpb12 PMX_LIB=$PWD/pharmsol_amd/lib/ab/ll_stag2.so PMX_TUNE_CPB=12
python - <<'PY'
import json
for f in ("base", "stag1", "stag2", "stag2_cpb12"):
    try:
        d = json.load(open("gpurun_out/r03_ll3_%s.json" % f))
        print(f, d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"], 4), d["max_rel_err_vs_cpu_ref"], d["parity_ok"])
    except Exception as e:
        print(f, "FAILED", e)
PY

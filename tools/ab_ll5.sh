#!/bin/bash
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --loglik --no-cpu-baseline > gpurun_out/r03_ll5_$name.json 2> gpurun_out/r03_ll5_$name.err; }
run base PMX_X=1
run same PMX_LIB=$PWD/pharmsol_amd/lib/ab/ll_same.so
run base_b PMX_X=1
run same_b PMX_LIB=$PWD/pharmsol_amd/lib/ab/ll_same.so
python - <<'PY'
import json
for f in ("base", "same", "base_b", "same_b"):
    try:
        d = json.load(open("gpurun_out/r03_ll5_%s.json" % f))
        print(f, d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"], 4), d["max_rel_err_vs_cpu_ref"], d["parity_ok"])
    except Exception as e:
        print(f, "FAILED", e)
PY

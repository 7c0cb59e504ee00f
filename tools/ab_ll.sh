#!/bin/bash
# A/B on ONE box: fused log-likelihood on C3 - the pipelined exact-class kernel vs the round-2 kernel (PMX_TUNE_LL_OLD).
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --loglik --no-cpu-baseline > gpurun_out/r03_ll_$name.json 2> gpurun_out/r03_ll_$name.err; }
run new PMX_X=1
run old PMX_TUNE_LL_OLD=1
run new_b PMX_X=1
run new_cpb12 PMX_TUNE_CPB=12
python - <<'PY'
import json
for f in ("new", "old", "new_b", "new_cpb12"):
    try:
        d = json.load(open("gpurun_out/r03_ll_%s.json" % f))
        print(f, d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"], 4), d["max_rel_err_vs_cpu_ref"], d["parity_ok"])
    except Exception as e:
        print(f, "FAILED", e)
PY

#!/bin/bash
# C5 A/B on one box: the round-2 generic covariate walker (PMX_DISABLE_DYN3=1) against the matrix-free walker with
# 1 / 2 (default) / 3 kept segments per lane, plus any variant libraries named on the command line.
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --workload c5 --no-cpu-baseline > gpurun_out/r03_c5ab_$name.json 2> gpurun_out/r03_c5ab_$name.err; }
run old PMX_DISABLE_DYN3=1
run new PMX_X=1
run slots1 PMX_TUNE_PROP_SLOTS=1
run slots3 PMX_TUNE_PROP_SLOTS=3
for v in "$@"; do run $v PMX_LIB=$PWD/pharmsol_amd/lib/ab/$v.so; done
python - "$@" <<'PY'
import json, sys
for f in ["old", "new", "slots1", "slots3"] + sys.argv[1:]:
    try:
        d = json.load(open("gpurun_out/r03_c5ab_%s.json" % f))
        print(f, d["config"]["kernel"], round(d["roofline"]["kernel_ms"], 3), d["max_rel_err_vs_cpu_ref"], d["parity_ok"])
    except Exception as e:
        print(f, "FAILED", e)
PY

#!/bin/bash
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --loglik --no-cpu-baseline > gpurun_out/r03_ll4_$name.json 2> gpurun_out/r03_ll4_$name.err; }
run base PMX_X=1
run nt PMX_LIB=$PWD/pharmsol_amd/lib/ab/ll_nt.so
run sc1nt PMX_LIB=$PWD/pharmsol_amd/lib/ab/ll_sc1nt.so
run base_nostatus PMX_X=1
python - <<'PY'
import json
for f in ("base", "nt", "sc1nt", "base_nostatus"):
    try:
        d = json.load(open("gpurun_out/r03_ll4_%s.json" % f))
        print(f, d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"], 4), d["max_rel_err_vs_cpu_ref"], d["parity_ok"])
    except Exception as e:
        print(f, "FAILED", e)
PY

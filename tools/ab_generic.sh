#!/bin/bash
# A/B on ONE box: the lean step walker vs the round-2 generic walker on C3 with classing disabled (+ its log-likelihood form),
# then the GPU parity suites that exercise the generic path.  usage (on the GPU box): bash tools/ab_generic.sh
mkdir -p gpurun_out
python bench.py --no-class --no-cpu-baseline > gpurun_out/r03_noclass_steps.json 2> gpurun_out/e1.err
PMX_DISABLE_STEPS=1 python bench.py --no-class --no-cpu-baseline > gpurun_out/r03_noclass_old.json 2> gpurun_out/e2.err
python bench.py --no-class --loglik --no-cpu-baseline > gpurun_out/r03_noclass_steps_ll.json 2> gpurun_out/e3.err
PMX_DISABLE_STEPS=1 python bench.py --no-class --loglik --no-cpu-baseline > gpurun_out/r03_noclass_old_ll.json 2> gpurun_out/e4.err
python - <<'PY'
import json
for f in ("r03_noclass_steps", "r03_noclass_old", "r03_noclass_steps_ll", "r03_noclass_old_ll"):
    try:
        d = json.load(open("gpurun_out/" + f + ".json"))
        print(f, d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"], 4), d["max_rel_err_vs_cpu_ref"], d["parity_ok"])
    except Exception as e:
        print(f, "FAILED", e)
PY

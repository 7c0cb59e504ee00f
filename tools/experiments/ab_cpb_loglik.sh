#!/bin/bash
# chunks per block on the fused log-likelihood launch of C3 (exact classes): PMX_TUNE_CPB sweep, one box
mkdir -p gpurun_out/r03b
for rep in 1 2; do for cpb in 0 1 2 3 4 6 8; do
  PMX_TUNE_CPB=$cpb timeout -k 10 200 python bench.py --loglik --no-cpu-baseline --steps 20 > gpurun_out/r03b/c.json 2>/dev/null || echo FAIL
  python3 - "$cpb" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03b/c.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("cpb", sys.argv[1], "loglik kernel_ms", round(r["kernel_ms"],4), d["parity_ok"])
PY
done; done

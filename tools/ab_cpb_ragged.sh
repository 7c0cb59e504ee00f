#!/bin/bash
# chunks per block on the compute-bound (jittered-times) classed launch: PMX_TUNE_CPB sweep, one box
mkdir -p gpurun_out/r02u
for rep in 1 2; do for cpb in 0 1 2 3 4 8; do
  for mode in "" "--loglik"; do
  PMX_TUNE_CPB=$cpb timeout -k 10 200 python bench.py --ragged $mode --no-cpu-baseline --steps 20 > gpurun_out/r02u/c.json 2>/dev/null || echo FAIL
  python3 - "$cpb" "$mode" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r02u/c.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("cpb", sys.argv[1], sys.argv[2] or "pred", "kernel_ms", round(r["kernel_ms"],4), d["parity_ok"])
PY
  done
done; done

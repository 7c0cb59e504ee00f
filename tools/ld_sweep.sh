#!/bin/bash
# Same-device sweep of the prediction leading dimension (row pitch) for the C3 bench.
for rep in 1 2; do
  for ld in 0 1008 1024 1040 1056 1088; do
    python bench.py --no-cpu-baseline --steps 20 --ld $ld "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep ld=$ld', round(d['ms_per_step'],4), 'ms', d['roofline']['frac'])"
  done
done

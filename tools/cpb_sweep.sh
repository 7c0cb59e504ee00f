#!/bin/bash
# Same-device sweep of chunks-per-block of the classed kernel (PMX_TUNE_CPB), C3 bench.
for rep in 1 2; do
  for cpb in 1 2 3 4 5 6 7 8 10 12 14 16 24; do
    PMX_TUNE_CPB=$cpb python bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep cpb=$cpb', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
  done
done

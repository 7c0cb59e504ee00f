#!/bin/bash
# Where should the GRID/PAIR switch sit?  Same build, PMX_TUNE_GRID_MIN_P = 1 (always GRID) vs 1000000 (always PAIR).
for P in 2 4 8 16 24 32 48; do
  for mode in 1 1000000; do
    PMX_TUNE_GRID_MIN_P=$mode python bench.py --no-cpu-baseline --steps 10 --support $P "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=$P min_p=$mode', round(d['ms_per_step'],4), 'ms', '%.3e'%d['value'], d['config']['kernel'])"
  done
done

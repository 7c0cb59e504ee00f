#!/bin/bash
# C3 with per-subject jittered sampling times: generic GRID kernel vs singleton classes through the classed kernel.
for rep in 1 2; do
  python bench.py --ragged --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep generic      ', round(d['ms_per_step'],4), 'ms', d['config']['kernel'], d['max_rel_err_vs_cpu_ref'])"
  PMX_TUNE_MIN_CLASS=1 python bench.py --ragged --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep min_class=1  ', round(d['ms_per_step'],4), 'ms', d['config']['kernel'], d['max_rel_err_vs_cpu_ref'])"
done

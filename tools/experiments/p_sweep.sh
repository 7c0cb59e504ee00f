#!/bin/bash
# Small support grids (one partial tile of lanes): current build vs pharmsol_amd/lib/ab/base.so.
for P in 32 64 100 130 200; do
  python bench.py --no-cpu-baseline --steps 10 --support $P "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=$P current', round(d['ms_per_step'],4), 'ms', '%.3e'%d['value'], d['config']['kernel'], d['max_rel_err_vs_cpu_ref'])"
  PMX_LIB=$PWD/pharmsol_amd/lib/ab/base.so python bench.py --no-cpu-baseline --steps 10 --support $P "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=$P base   ', round(d['ms_per_step'],4), 'ms', '%.3e'%d['value'])"
done

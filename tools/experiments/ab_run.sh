#!/bin/bash
# Same-device A/B: current build vs every pharmsol_amd/lib/ab/*.so, interleaved rounds.  Extra bench args: $@
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep current ', round(d['ms_per_step'],4), 'ms', d['config']['kernel'])"
  for lib in pharmsol_amd/lib/ab/*.so; do
    PMX_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep $(basename $lib) ', round(d['ms_per_step'],4), 'ms', d['config']['kernel'])"
  done
done

#!/bin/bash
# Chunk membership: consecutive subjects vs spread (member j = subject c + j*n_chunks of the class), same build.
tools/bin/store_probe 2>/dev/null | grep -E "A2 one store per lane  |B classed map nt cpb=6|H members spread nt cpb=6" | head -3
for rep in 1 2 3; do
  for sp in 0 1; do
    PMX_TUNE_SPREAD=$sp python bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep spread=$sp', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4), d['max_rel_err_vs_cpu_ref'])"
  done
done

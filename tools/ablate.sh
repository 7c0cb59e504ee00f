#!/bin/bash
# Ablation of the classed kernel on C3, all variants in ONE session on ONE device (devices differ by ~10%).
# PMX_DEBUG_FLAGS bits: 1 = no prediction stores, 2 = no propagator math, 4 = 8-byte stores (no pairing),
#                       8 = plain round-robin block map (no XCD grouping)
for rep in 1 2; do
for f in ${FLAGS:-0 4 8 12 1 2 3}; do
  PMX_DEBUG_FLAGS=$f python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep flags=$f', round(d['ms_per_step'],4), 'ms', d['config']['kernel'])"
done
done

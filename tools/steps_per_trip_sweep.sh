#!/bin/bash
# PAIR-mode ODE kernel: RK4 steps per trip of the lane state machine (PMX_STEPS_PER_TRIP), libraries built as
# pharmsol_amd/lib/ab/k<K>.so; C4 at 50k (latency-bound) and 400k subjects (throughput-bound).
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for k in 1 4 8 16 32; do
  if [ $k = 8 ]; then unset PMX_LIB; else export PMX_LIB=pharmsol_amd/lib/ab/k$k.so; fi
  for n in 50000 400000; do
    python bench.py --workload c4 --subjects $n --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('K=$k subjects %7d  %.3f ms  %.3e steps/s  err %.2e' % (d['config']['subjects_per_gpu'], d['ms_per_step'], d['value'], d['max_rel_err_vs_cpu_ref']))" | tee -a gpurun_out/steps_per_trip.txt
  done
done

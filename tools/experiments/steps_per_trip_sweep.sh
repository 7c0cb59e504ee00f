#!/bin/bash
# PAIR-mode ODE kernel: RK4 steps per trip of the lane state machine (PMX_TUNE_STEPS_PER_TRIP) by batch size.
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for n in 12500 50000 100000 200000 400000; do
  for k in 1 8 16 32 64 128; do
    PMX_TUNE_STEPS_PER_TRIP=$k python bench.py --workload c4 --subjects $n --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('subjects %7d K=%3d  %.3f ms  %.3e steps/s  err %.2e' % (d['config']['subjects_per_gpu'], $k, d['ms_per_step'], d['value'], d['max_rel_err_vs_cpu_ref']))" | tee -a gpurun_out/steps_per_trip.txt
  done
done

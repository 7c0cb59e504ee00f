#!/bin/bash
# C4 (ODE, one theta per subject, PAIR mode) at growing subject counts: latency- or throughput-bound?
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for n in 6250 12500 25000 50000 100000 200000 400000; do
  python bench.py --workload c4 --subjects $n --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('subjects %7d  %.3f ms  %.3e steps/s' % (d['config']['subjects_per_gpu'], d['ms_per_step'], d['value']))" | tee -a gpurun_out/c4_scaling.txt
done

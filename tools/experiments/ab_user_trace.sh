#!/bin/bash
export TMPDIR=/tmp
for t in r2 r3; do
  d=$PWD; [ $t = r2 ] && d=$PWD/ab_r2tree
  out=$PWD/gpurun_out/ab_user_trace_$t; mkdir -p $out
  (cd $d && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 bench.py --workload user --steps 20 --warmup 3 --no-cpu-baseline --place-gib 0 > $out/bench.json 2> $out/err.txt)
  echo "== $t"; python3 -c "import json; d=json.load(open('$out/bench.json')); print('kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'])"
  st=$(find $out -name "*kernel_stats.csv" | head -1); head -6 "$st" | cut -c1-200
done

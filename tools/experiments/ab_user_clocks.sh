#!/bin/bash
# user-closure workload, round-2 tree vs working tree, plain runs with the clocks and the power polled beside them
for t in r2 r3; do
  d=$PWD; [ $t = r2 ] && d=$PWD/ab_r2tree
  out=$PWD/gpurun_out/ab_user_clk_$t.txt; : > $out
  (cd $d && python bench.py --workload user --no-cpu-baseline --place-gib 0 --steps 1500 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t kernel_ms', round(d['roofline']['kernel_ms'],3))") &
  pid=$!
  while kill -0 $pid 2>/dev/null; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' ' >> $out; echo >> $out; sleep 0.5; done
  wait $pid
  echo "== $t (last 8 polls)"; tail -8 $out | cut -c1-300
done

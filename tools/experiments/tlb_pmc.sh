#!/bin/bash
# Address-translation counters of the C3 kernel into the FIRST plain allocation (usually a slow one) and into the best of
# eight (the time-boxed placement hypothesis: scattered row walks vs translation reach).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03_tlb_pmc
mkdir -p $OUT
for mode in first best; do
  extra="--alloc-tries 1"; [ $mode = best ] && extra="--place-gib 1 --alloc-tries 8"
  timeout -k 10 150 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum \
    --output-format csv -d $OUT/$mode -o pmc -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --place-gib 0 $extra > $OUT/$mode.json 2> $OUT/$mode.err
  echo "$mode rc=$?"
  python3 - $OUT/$mode <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "classed" in k:
        print(k)
        for c, xs in sorted(v.items()):
            print("   %-50s %16.1f  (n=%d)" % (c, sum(xs) / len(xs), len(xs)))
PY
  python3 -c "import json,sys; d=json.load(open('$OUT/$mode.json')); print('$mode kernel_ms', d['roofline']['kernel_ms'], d['config']['prediction_buffer'])"
done

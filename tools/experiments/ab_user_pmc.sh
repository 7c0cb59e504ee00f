#!/bin/bash
# user-closure workload, round-2 tree (ab_r2tree/, built by hand from a3acae1) against the working tree: instruction mix
export TMPDIR=/tmp
for t in r2 r3; do
  d=$PWD; [ $t = r2 ] && d=$PWD/ab_r2tree
  out=$PWD/gpurun_out/ab_user_$t; mkdir -p $out
  (cd $d && timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out -o pmc -- python3 bench.py --workload user --steps 3 --warmup 1 --no-cpu-baseline --place-gib 0 > $out/bench.json 2> $out/err.txt)
  python3 - $out <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "agrid" in k:
        print(sys.argv[1].split("_")[-1], k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
PY
done

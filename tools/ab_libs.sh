#!/bin/bash
# A/B of library builds on one box: tools/ab_libs.sh <outdir> "<bench args>" <lib name|cur> ...   (libs from tools/ab_variant.sh)
OUT=$1; ARGS=$2; shift 2
mkdir -p $OUT
for rep in 1 2; do for v in "$@"; do
  lib=""; [ $v != cur ] && lib=$PWD/pharmsol_amd/lib/ab/$v.so
  PMX_LIB=$lib timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline > $OUT/tmp.json 2>/dev/null || echo "FAIL $v"
  python3 - "$v" "$ARGS" $OUT/tmp.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], "|", sys.argv[2], "| kernel_ms", round(r["kernel_ms"],4), "frac", r.get("frac") and round(r["frac"],4), "err", d.get("max_rel_err_vs_cpu_ref"), d["parity_ok"])
PY
done; done

#!/bin/bash
# The two SQ counter passes of tools/pmc_run.sh only (instruction mix and wait cycles), for A/B variants.
# usage: tools/pmc_quick.sh <outdir> [bench args...]      (variant selection through the environment: PMX_LIB, PMX_TUNE_*)
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --place-gib 0 $*"
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pass$i" -o pmc -- python3 bench.py $ARGS > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$?"
done
python3 tools/pmc_summary.py "$OUT" "$OUT.json" > "$OUT.txt"; cat "$OUT.txt"

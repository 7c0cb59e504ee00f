#!/bin/bash
# PMC counters of the headline kernel writing into a FAST and into a SLOW window of one placement arena (same box, same
# process layout: the landscape of an arena is reproducible from process to process on one box).
# usage: tools/experiments/window_pmc.sh <outdir> <arena GiB>
set -u
export TMPDIR=/tmp
OUT=$1; GIB=$2
mkdir -p $OUT
PMX_TUNE_PLACE_FULL=1 PMX_DEBUG_PLACEMENT=1 python3 bench.py --place-gib $GIB --no-cpu-baseline --steps 5 2> $OUT/landscape.txt > /dev/null
grep "window at" $OUT/landscape.txt | awk '{print NR-1, $8}' > $OUT/windows.txt
FAST=$(sort -k2 -n $OUT/windows.txt | head -1 | cut -d" " -f1)
SLOW=$(sort -k2 -n $OUT/windows.txt | tail -1 | cut -d" " -f1)
echo "fast window $FAST slow window $SLOW" | tee $OUT/choice.txt
for kind in fast slow; do
  W=$FAST; [ $kind = slow ] && W=$SLOW
  i=0; mkdir -p $OUT/$kind
  for set in \
    "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" \
    "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
    "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum" \
    "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_WRITEBACK_sum TCC_WRITE_sum" \
    "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" ; do
    i=$((i+1))
    PMX_TUNE_PLACE_WINDOW=$W rocprofv3 --pmc $set --output-format csv -d "$OUT/$kind/pass$i" -o pmc -- python3 bench.py --place-gib $GIB --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/$kind/pass$i.log" 2>&1
    echo "$kind pass $i rc=$?"
  done
  python3 tools/pmc_summary.py $OUT/$kind $OUT/$kind.json > $OUT/$kind.txt
done
paste $OUT/fast.txt $OUT/slow.txt

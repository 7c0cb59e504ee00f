#!/bin/bash
# Round profile (run on the GPU box): the default bench line; rocprofv3 kernel traces of the placed and the
# first-allocation run with the timed passes isolated (last K dispatches); PMC passes (separate runs, counters only) for
# the headline and the compute-bound workloads -> profiles/kernel_counters.json.
# usage: tools/profile_round.sh <tag>      (outputs under gpurun_out/prof_<tag>/)
set -u
export TMPDIR=/tmp
TAG=$1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
K=20
PMX_DEBUG_PLACEMENT=1 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
# the traced "placed" run sits in the window the untraced run chose (an arena's landscape repeats from process to
# process on one box) without searching, so the trace and its --stats average hold passes into that window only
WIN=$(grep "chosen window" $OUT/bench.err | tail -1 | awk '{print $4}'); echo "window ${WIN:-none}"
# (a box whose arena held no window faster than the plain first allocation: the bench kept that one, so does the trace)
PLAIN=0; grep -q '"prediction_buffer": "best of [0-9]* plain' $OUT/bench.json && PLAIN=1 && WIN="" && echo "bench kept the first allocation"
for mode in placed first; do
  extra=""; [ $mode = first ] && extra="--place-gib 0"; [ $mode = placed ] && [ $PLAIN = 1 ] && extra="--place-gib 0"
  export -n PMX_TUNE_PLACE_WINDOW; [ $mode = placed ] && [ -n "${WIN:-}" ] && export PMX_TUNE_PLACE_WINDOW=$WIN
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$mode -o kt -- python3 bench.py --no-cpu-baseline --steps $K $extra > $OUT/bench_under_rocprof_$mode.json 2> $OUT/trace_$mode.err; echo "trace $mode rc=$?"
  kt=$(find $OUT/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_last_k.py "$kt" $K $OUT/bench_under_rocprof_$mode.json > $OUT/trace_${mode}_last_k.json; cat $OUT/trace_${mode}_last_k.json
  st=$(find $OUT/trace_$mode -name "*kernel_stats.csv" | head -1); cp "$st" $OUT/kernel_stats_$mode.csv
done
unset PMX_TUNE_PLACE_WINDOW
for w in "c3:" "c5:--workload c5" "c3_ragged:--ragged" "c3_generic:--no-class" "c3_loglik:--loglik" "c3_ragged_loglik:--ragged --loglik" "c4:--workload c4" "user:--workload user"; do
  key=${w%%:*}; args=${w#*:}
  tools/pmc_run.sh $OUT/pmc_$key $args > $OUT/pmc_$key.log 2>&1; echo "pmc $key rc=$?"
  python3 tools/pmc_summary.py $OUT/pmc_$key $OUT/pmc_$key.json > $OUT/pmc_$key.txt
  python3 tools/kernel_counters.py $key $OUT/pmc_$key.json "round $TAG build: rocprofv3 --pmc, separate passes (tools/pmc_run.sh $args), mean per dispatch" > /dev/null
done
cp profiles/kernel_counters.json $OUT/kernel_counters.json
for w in c2 c4 c5 user; do python bench.py --workload $w --no-cpu-baseline --steps 10 > $OUT/bench_$w.json 2>> $OUT/bench.err; done
python bench.py --ragged --no-cpu-baseline --steps 10 > $OUT/bench_c3_ragged.json 2>> $OUT/bench.err
python bench.py --loglik --no-cpu-baseline --steps 10 > $OUT/bench_c3_loglik.json 2>> $OUT/bench.err
python bench.py --no-class --no-cpu-baseline --steps 10 > $OUT/bench_c3_generic.json 2>> $OUT/bench.err
PMX_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --subjects 50000 --scaling strong --no-cpu-baseline > $OUT/bench_2rank_gloo_rehearsal.json 2> $OUT/rehearsal.err; echo "rehearsal rc=$?"
cat $OUT/bench.json

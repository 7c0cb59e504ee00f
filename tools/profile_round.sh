#!/bin/bash
# Round profile: rocprofv3 kernel-trace stats of the default bench command + PMC traffic passes.
# usage: tools/profile_round.sh <tag>      (outputs under gpurun_out/prof_<tag>/)
set -u
export TMPDIR=/tmp
TAG=$1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o kt -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err; echo "trace rc=$?"
tools/pmc_run.sh $OUT/pmc > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
for w in c2 c4 c5; do python bench.py --workload $w --no-cpu-baseline --steps 10 > $OUT/bench_$w.json 2>> $OUT/bench.err; done
python bench.py --no-class --no-cpu-baseline --steps 10 > $OUT/bench_c3_generic.json 2>> $OUT/bench.err
PMX_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --subjects 20000 > $OUT/bench_2rank_gloo_rehearsal.json 2> $OUT/rehearsal.err; echo "rehearsal rc=$?"
cat $OUT/bench.json

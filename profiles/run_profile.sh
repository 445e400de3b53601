#!/bin/bash
# Profiles bench.py under rocprofv3 on the GPU box (invoke through gpurun from the repo root):
#   gpurun -- 'bash profiles/run_profile.sh r1'
# Writes raw output under gpurun_out/prof_<tag>/; copy the summaries worth keeping into profiles/.
TAG=${1:-r1}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
PMCARGS="--steps 4 --warmup 1 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --no-kernel-timing"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $PMCARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $PMCARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.log
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $PMCARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.log
find $OUT -name "*.csv" | head -50
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done

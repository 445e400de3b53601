#!/bin/bash
# Profiles bench.py under rocprofv3 on the GPU box (invoke through gpurun from the repo root):
#   gpurun -- 'bash profiles/run_profile.sh r6'            (or: ... r6 latency)
# Writes raw output under gpurun_out/prof_<tag>/; profiles/summarize.py condenses it into profiles/<tag>_*.
TAG=${1:-r6}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--cpu-seconds 0 --legs none"
LEGS=$REPO/bench_support/run_legs.py
ARGS="--steps 200 --warmup 5 --repeats 1 $COMMON"
# (second argument "latency": only the one-frame legs are re-run, the rest of gpurun_out/prof_<tag>/ is kept)
ONLY=${2:-all}
if [ "$ONLY" = all ]; then
# the driver's own command, un-traced: the line of record and its detail file
(cd $REPO && python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_default_detail.json > $OUT/bench_default.json 2> $OUT/bench_default.log)
# the bench default: two contexts alternating (a projection beside the other context's feature kernels)
# (only launches of the timed schedule in this trace: no plane-estimated leg, no kernels-alone pass)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS --no-exclusive --detail $OUT/bench_trace.json > $OUT/bench_trace_line.json 2> $OUT/trace.log
# one context, one kernel at a time: what each kernel takes with the GPU to itself (1024 frames per launch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_x -- python3 $REPO/bench.py --contexts 1 $ARGS --detail $OUT/bench_trace_x.json > $OUT/bench_trace_x_line.json 2> $OUT/trace_x.log
# the plane-estimated leg on its own (k_rs_batch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_e -- python3 $LEGS --legs estimated --est-steps 2 > $OUT/bench_trace_e.json 2> $OUT/trace_e.log
fi
# the one-frame-per-call legs (supplied plane, RANSAC and semantic plane estimated inside the call): kernels of a frame
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_l -- python3 $LEGS --legs latency --latency-frames 100 > $OUT/bench_trace_l.json 2> $OUT/trace_l.log
# the same legs WITHOUT the tracer (the numbers of record for the one-frame calls: rocprofv3 costs them 15-30 us)
python3 $LEGS --legs latency,streaming --latency-frames 200 --streaming-batches 24 > $OUT/bench_latency.json 2> $OUT/latency.log
if [ "$ONLY" = all ]; then
# counters: rocprofv3 serialises the kernels in these passes, so they are collected on the one-context schedule
PMCARGS="--contexts 1 --steps 4 --warmup 1 --repeats 1 $COMMON --no-kernel-timing"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $PMCARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $PMCARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.log
# L1 / L2 request counts of the gather-bound feature kernel (gather roof)
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc_tcp -- python3 $REPO/bench.py $PMCARGS > $OUT/bench_pmc_tcp.json 2> $OUT/pmc_tcp.log
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $REPO/bench.py $PMCARGS > $OUT/bench_pmc_tcc.json 2> $OUT/pmc_tcc.log
# the random-gather line rates of this box (ceilings of that roof), same session
if [ -x $REPO/profiles/tools/libs/randgather ]; then $REPO/profiles/tools/libs/randgather > $OUT/randgather.txt 2>&1; fi
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $PMCARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.log
# the other single-GPU BASELINE configs, one leg per run
for c in 3 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c$c -- python3 $LEGS --legs c$c > $OUT/bench_c$c.json 2> $OUT/trace_c$c.log
done
# counter passes of the other legs (one leg per command, one context, so that a kernel's launches are all of one size):
#   2k     config 2 at k = 7 (features on returns whose window holds >= 6 returns), 1024 frames per launch
#   3n     config 3, features around the returns (c0_dispose), 256 frames per launch
#   5b256  config 5, 256 sequences per step (the DENSE instantiation of the feature kernel)
leg() {  # key, bench arguments
  local key=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/leg_${key}_trace -- python3 $LEGS "$@" > $OUT/leg_${key}_trace.json 2> $OUT/leg_${key}_trace.log
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/leg_${key}_fetch -- python3 $LEGS "$@" > $OUT/leg_${key}_fetch.json 2> $OUT/leg_${key}_fetch.log
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/leg_${key}_write -- python3 $LEGS "$@" > $OUT/leg_${key}_write.json 2> $OUT/leg_${key}_write.log
}
leg 2k --legs c2k1
leg 3n --legs c3n
leg 5b256 --legs c5b256
fi
cd $REPO
python3 profiles/summarize.py $TAG > $OUT/summary.log 2>&1
python3 profiles/summarize_config_pmc.py $TAG 2k 1024 "python3 bench_support/run_legs.py --legs c2k1" >> $OUT/summary.log 2>&1
python3 profiles/summarize_config_pmc.py $TAG 3n 256 "python3 bench_support/run_legs.py --legs c3n" >> $OUT/summary.log 2>&1
python3 profiles/summarize_config_pmc.py $TAG 5b256 256 "python3 bench_support/run_legs.py --legs c5b256" >> $OUT/summary.log 2>&1
python3 profiles/summarize_config.py $TAG 3 >> $OUT/summary.log 2>&1
python3 profiles/summarize_config.py $TAG 5 >> $OUT/summary.log 2>&1
cp $OUT/bench_default.json profiles/${TAG}_bench_default.json 2>/dev/null
mkdir -p $OUT/keep && cp profiles/${TAG}_*.md profiles/${TAG}_*.csv profiles/${TAG}_*.txt profiles/traffic.json $OUT/keep/ 2>/dev/null
tail -40 $OUT/summary.log

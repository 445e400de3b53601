#!/bin/bash
# usage: ab_cusplit.sh ROUNDS  - the bench default schedule against CU-split schedules on ONE box: the pair's projection
# stream confined to N CUs, classification + feature kernels to the other 256 - N (test build, MLD_CU_SPLIT; LAB.md).
ROUNDS=${1:-2}
COMMON="--steps 60 --warmup 5 --repeats 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated --verify-slots 16"
AB="MLD_HIP_LIBRARY=$PWD/mono_lidar_depth_amd/lib/libmld_hip_ab.so"
run() {  # label, env, extra bench args
  echo -n "$1: "
  env $AB $2 timeout 300 python bench.py $COMMON $3 2>gpurun_out/ab_last.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
except Exception as e:
    print('no json', e); sys.exit(0)
r=d['roofline']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), [round(v,4) for v in d['timed_loops']['ms_per_step']], {k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items()}, 'verified', d['verified'])"
}
for r in $(seq 1 $ROUNDS); do
  run "default r$r" "MLD_DUMMY=1" ""
  run "pair-unsplit r$r" "MLD_DUMMY=1" "--pair"
  for n in 64 96 128 160 192; do
    run "split-$n r$r" "MLD_CU_SPLIT=$n" "--pair --shared-mode 0"
    run "split-${n}s r$r" "MLD_CU_SPLIT=${n},s" "--pair --shared-mode 0"
  done
done

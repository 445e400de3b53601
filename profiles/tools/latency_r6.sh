#!/bin/bash
# usage: latency_r6.sh ROUNDS lib ...  - the one-frame-per-call legs (PCIe-inclusive) per library variant on ONE box:
#   supplied / RANSAC / semantic / process medians (us) and the config-5 frame call (10 000 tracks, device pointers)
ROUNDS=$1; shift
for round in $(seq 1 $ROUNDS); do
for lib in "$@"; do
  case "$lib" in
    -) L="MLD_DUMMY=1";;
    *) L="MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so";;
  esac
  echo -n "$lib r$round: "
  env $L timeout 300 python bench_support/run_legs.py --legs latency --latency-frames 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['latency']
g=lambda x: round(x['ms_per_frame_median']*1e3,1)
print('supplied', g(d), 'ransac', g(d['estimated']['ransac']), 'semantic', g(d['estimated']['semantic']), 'process', g(d['process']['one_call']), 'kernels_us', round(d['breakdown_us_median']['kernels_us'],1), end=' ')"
  env $L timeout 300 python bench_support/run_legs.py --legs c5b16 --config-frames 100 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['configs']['5']['batched']['16']
print('| c5 S=16 step', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})"
done; done

#!/bin/bash
# usage: ab_road_crit.sh lib ...  - per library variant (NAME.so = product build, NAME_ab.so = test build) on ONE box: the
#   config-2 default step and the 5000-configuration parity sweep on the lane-per-feature route (LAB.md 5.33)
C2="--steps 60 --warmup 5 --repeats 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated --no-exclusive"
for lib in "$@"; do
  echo -n "$lib c2: "
  MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so python bench.py $C2 2>/dev/null | python profiles/tools/show_step.py
  echo -n "$lib 3n: "
  MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so python bench.py --only-config 3 --leg near 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['near_returns']['modes']['c0_dispose']
print(round(d['associations_per_s']/1e6,1),'M/s', {k:round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"
  MLD_HIP_AB_LIBRARY=$PWD/profiles/tools/libs/${lib}_ab.so python profiles/tools/random_sweep.py 3000 ${SWEEP:-2500} fused 2>&1 | grep -v amdgpu.ids | grep "random sweep\|MISMATCH"
done

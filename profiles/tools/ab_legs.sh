#!/bin/bash
# usage: ab_legs.sh ROUNDS lib ...  - per library variant on ONE box: the config-2 default schedule, config 2 at k = 7
#   (--only-config 2), config 3 with the features around the returns (--only-config 3 --leg near).
ROUNDS=$1; shift
C2="--steps 60 --warmup 5 --repeats 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated --verify-slots 32"
for round in $(seq 1 $ROUNDS); do
for lib in "$@"; do
  case "$lib" in
    -) L="MLD_DUMMY=1";;
    *) L="MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so";;
  esac
  echo -n "$lib r$round c2: "
  env $L timeout 300 python bench.py $C2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; x=(r.get('exclusive') or {}).get('kernels_ms',{})
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items()}, 'alone', {k:round(v*1e3,1) for k,v in x.items()}, d['verified'])"
  echo -n "$lib r$round 2k: "
  env $L timeout 300 python bench.py --only-config 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"
  echo -n "$lib r$round 2k one context: "
  env $L timeout 300 python bench.py --only-config 2 --contexts 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"
  echo -n "$lib r$round 3n: "
  env $L timeout 300 python bench.py --only-config 3 --leg near 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['near_returns']['modes']['c0_dispose']
print(round(d['associations_per_s']/1e6,1),'M/s', {k:round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"
done; done

#!/bin/bash
# k_classify with the bitmap staged in LDS (68 KB per block: cannot start on a CU whose LDS the other context's feature
# wavefronts hold) against reading it in place (4 KB): the config-2 default schedule and the k = 7 leg, one box.
AB="MLD_HIP_LIBRARY=$PWD/mono_lidar_depth_amd/lib/libmld_hip_ab.so"
C2="--steps 60 --warmup 5 --repeats 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated --verify-slots 32"
for round in 1 2; do
for st in 1 0; do
  echo -n "staged=$st r$round c2: "
  env $AB MLD_CLASSIFY_STAGED=$st timeout 300 python bench.py $C2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; x=(r.get('exclusive') or {}).get('kernels_ms',{})
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items()}, 'alone', {k:round(v*1e3,1) for k,v in x.items()}, d['verified'])"
  echo -n "staged=$st r$round 2k: "
  env $AB MLD_CLASSIFY_STAGED=$st timeout 300 python bench.py --only-config 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"
done; done

#!/bin/bash
# default bench.py (two contexts, steps alternating) per profiles/tools/libs/*.so, ROUNDS times
ROUNDS=${1:-2}
for round in $(seq 1 $ROUNDS); do
for lib in profiles/tools/libs/*.so; do
  echo -n "$(basename $lib .so) r$round: "
  MLD_HIP_LIBRARY=$PWD/$lib timeout 200 python bench.py --steps 60 --warmup 5 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated --no-exclusive 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items()}, 'verified', d['verified'])"
done; done

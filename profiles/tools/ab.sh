#!/bin/bash
# runs bench.py for every profiles/tools/libs/*.so, alternating; prints kernel times (us per 1024 frames)
for round in 1 2 3; do
for lib in profiles/tools/libs/*.so; do
  echo -n "$(basename $lib .so) r$round: "
  MLD_HIP_LIBRARY=$PWD/$lib timeout 120 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v['avg_ms']*1e3,1) for k,v in r['kernels'].items()}, 'wave', round(r['k_feature_wave_ms']*1e3,1))"
done; done

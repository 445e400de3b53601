#!/bin/bash
# bench.py (default alternating contexts, and --contexts 1) per profiles/tools/libs/*.so
for lib in profiles/tools/libs/*.so; do
  for nc in 1 2; do
  echo -n "$(basename $lib .so) contexts=$nc: "
  MLD_HIP_LIBRARY=$PWD/$lib timeout 120 python bench.py --contexts $nc --steps 20 --warmup 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items()}, 'excl', {k:round(v*1e3,1) for k,v in (r.get('exclusive') or {}).get('kernels_ms',{}).items()}, 'verified', d['verified'])"
  done
done

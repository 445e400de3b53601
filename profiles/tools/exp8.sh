#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--steps 20 --warmup 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0"
for cfg in "--contexts 1" "--contexts 2" "--contexts 4" "--contexts 2 --slots 256" "--contexts 1 --slots 512"; do
for r in 1 2; do
echo -n "$cfg r$r: "
timeout 120 python bench.py $B $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v['avg_ms']*1e3,1) for k,v in r['kernels'].items()}, 'classify', round(r['k_sort_features_ms']*1e3,1))"
done; done

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp4
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
B="--steps 20 --warmup 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0"
for r in 1 2; do
for leg in 0 1; do
echo -n "legacy=$leg r$r: "
MLD_LEGACY_SPLIT=$leg timeout 120 python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v['avg_ms']*1e3,1) for k,v in r['kernels'].items()}, 'sort/classify', round(r['k_sort_features_ms']*1e3,1), 'wave', round(r['k_feature_wave_ms']*1e3,1), d['result_types'])"
done; done

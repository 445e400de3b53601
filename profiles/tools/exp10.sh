#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --only-config 5 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config5', round(d['ms_per_frame'],4), 'ms/frame', round(d['associations_per_s']/1e6,1), 'M/s', d['kernels_ms_per_launch'], d['verified'])"
python bench.py --only-config 3 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config3', {k:(round(v['associations_per_s']/1e9,2), v['verified']) for k,v in d['modes'].items()})"
python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items()}, d['verified'], d['plane_estimated'])"

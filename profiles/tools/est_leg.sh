#!/bin/bash
# the plane-estimated leg of bench.py (k_rs_batch ahead of the projection): two contexts and one context
for extra in "" "--contexts 1"; do
python bench.py --steps 20 --warmup 3 --repeats 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 $extra 2>gpurun_out/est.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['plane_estimated']
print('$extra', 'step', round(d['ms_per_step'],4), 'estimated', round(e['ms_per_step'],4), {k:round(v*1e3,1) for k,v in e['kernels_ms_per_launch'].items()}, 'frames/launch', e['frame_slots_per_launch'], e['verified'])"
done

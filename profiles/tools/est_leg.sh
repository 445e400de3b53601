#!/bin/bash
# the plane-estimated leg of bench.py (k_rs_batch ahead of the projection): two contexts in turn, two contexts with
# half a step each, one context; MLD_RS_PRIORITY=1: k_rs_batch on a stream of the highest priority
run() {
python bench.py --steps 100 --warmup 3 --repeats 1 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 $1 2>gpurun_out/est.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['plane_estimated']
print('$1 prio=$MLD_RS_PRIORITY', 'step', round(d['ms_per_step'],4), 'estimated', round(e['ms_per_step'],4), {k:round(v*1e3,1) for k,v in e['kernels_ms_per_launch'].items()}, 'frames/launch', e['frame_slots_per_launch'], e['verified'])"
}
for r in 1 2; do
for extra in "--est-schedule turn" "--est-schedule split" "--contexts 1"; do
  MLD_RS_PRIORITY=0 run "$extra"
  MLD_RS_PRIORITY=1 run "$extra"
done
done

#!/bin/bash
# the plane-estimated leg of bench.py (k_rs_batch ahead of the projection): two contexts with half a step each and one
# context; arguments: libraries to compare ("-" = the in-tree one)
run() {
python bench.py --steps 100 --warmup 3 --repeats 1 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 $1 2>gpurun_out/est.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['plane_estimated']
print('$2 $1', 'step', round(d['ms_per_step'],4), 'estimated', round(e['ms_per_step'],4), {k:round(v*1e3,1) for k,v in e['kernels_ms_per_launch'].items()}, 'frames/launch', e['frame_slots_per_launch'], e['verified'])"
}
LIBS=${@:--}
for r in 1 2; do
for extra in "" "--est-schedule alternate" "--contexts 1"; do
  for lib in $LIBS; do
    if [ "$lib" = "-" ]; then unset MLD_HIP_LIBRARY; else export MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so; fi
    run "$extra" "$lib"
  done
done
done

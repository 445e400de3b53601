#!/bin/bash
# The 1024-frame step cut into launch sets of S frame slots dealt to the contexts in turn (bench.py --slots S): does the
# feature kernel get faster when it reads what its projection wrote a few tens of microseconds earlier?
for spec in "0 2" "512 2" "256 2" "128 2" "64 2" "256 4" "128 4" "128 8"; do
set -- $spec
python bench.py --steps 40 --warmup 5 --repeats 3 --no-estimated --config-frames 0 --streaming-batches 0 --cpu-seconds 0 --latency-frames 0 --slots $1 --contexts $2 --verify-slots 2 --no-exclusive > gpurun_out/slots_$1_$2.json 2>gpurun_out/slots.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/slots_$1_$2.json').read().strip().splitlines()[-1])
    k=d['roofline']['kernels']; S=d['config']['frame_slots_per_launch']
    print('slots', $1, 'contexts', $2, 'G/s', round(d['value']/1e9,3), 'ms/step', round(d['ms_per_step'],4), 'per launch us', {n:round(v['avg_ms']*1e3,1) for n,v in k.items() if 'avg_ms' in v}, 'per frame ns', {n:round(v['avg_ms']*1e6/S,1) for n,v in k.items() if 'avg_ms' in v}, d['verified'])
except Exception as e:
    print('slots', $1, 'contexts', $2, 'failed', e, open('gpurun_out/slots.err').read()[-300:])
PY
done

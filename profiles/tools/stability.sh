#!/bin/bash
# per-loop step times of the default schedule over long loops (the hand-over schedule has a fast and a slow regime)
for extra in "" "--shared-mode 2305" "--shared-mode 1793"; do
  echo -n "[$extra] "
  python bench.py --steps 200 --warmup 5 --repeats 7 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated --no-exclusive $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],4), [round(x,4) for x in d['timed_loops']['ms_per_step']], {k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items() if v})"
done

#!/bin/bash
# where does the two-context overlap of the secondary legs get lost?  config 2 at k = 7 (two contexts):
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
d=d['configs']['2']['near_returns']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v*1e3) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"; }
echo -n "A standalone c2k:                         "; python bench_support/run_legs.py --legs c2k 2>/dev/null | show
echo -n "B standalone streaming,c2k:               "; python bench_support/run_legs.py --legs streaming,c2k 2>/dev/null | show
echo -n "C standalone estimated,c2k:               "; python bench_support/run_legs.py --legs estimated,c2k 2>/dev/null | show
echo -n "D standalone latency,c2k:                 "; python bench_support/run_legs.py --legs latency,c2k 2>/dev/null | show
echo -n "E bench.py headline, then child c2k:      "; python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --legs c2k --detail gpurun_out/r6a/probe2_detail.json >/dev/null 2>&1; python -c "
import json
d=json.load(open('gpurun_out/r6a/probe2_detail.json'))['configs']['2']['near_returns']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v*1e3) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"
echo -n "F bench.py with cpu baseline, child c2k:  "; python bench.py --steps 20 --warmup 5 --cpu-seconds 6 --legs c2k --detail gpurun_out/r6a/probe2_detail.json >/dev/null 2>&1; python -c "
import json
d=json.load(open('gpurun_out/r6a/probe2_detail.json'))['configs']['2']['near_returns']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v*1e3) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"

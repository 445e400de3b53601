#!/bin/bash
# usage: ab_r3.sh ROUNDS "lib[:bench args]" ...   - default bench.py schedule per variant on ONE box, ROUNDS times
# (lib = name of profiles/tools/libs/<name>.so, or "-" for the in-tree library)
ROUNDS=$1; shift
COMMON="--steps 60 --warmup 5 --repeats 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated"
for round in $(seq 1 $ROUNDS); do
for v in "$@"; do
  lib=${v%%:*}; extra=""; [[ "$v" == *:* ]] && extra=${v#*:}
  echo -n "$v r$round: "
  if [ "$lib" = "-" ]; then L="MLD_DUMMY=1"; else L="MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so"; fi
  env $L timeout 300 python bench.py $COMMON $extra 2>gpurun_out/ab_last.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
except Exception as e:
    print('no json', e); sys.exit(0)
r=d['roofline']; x=(r.get('exclusive') or {}).get('kernels_ms',{})
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), [round(v,4) for v in d['timed_loops']['ms_per_step']], {k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items()}, 'alone', {k:round(v*1e3,1) for k,v in x.items()}, 'verified', d['verified'], 'dom', r['kernel'])"
done; done

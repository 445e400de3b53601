#!/bin/bash
# usage: ab_r3.sh ROUNDS "lib[:bench args]" ...   - default bench.py schedule per variant on ONE box, ROUNDS times
# (lib = name of profiles/tools/libs/<name>.so, or "-" for the in-tree library)
ROUNDS=$1; shift
COMMON="--steps 60 --warmup 5 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated"
for round in $(seq 1 $ROUNDS); do
for v in "$@"; do
  lib=${v%%:*}; extra=""; [[ "$v" == *:* ]] && extra=${v#*:}
  echo -n "$v r$round: "
  if [ "$lib" = "-" ]; then L="MLD_DUMMY=1"; else L="MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so"; fi
  env $L timeout 300 python bench.py $COMMON $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; x=(r.get('exclusive') or {}).get('kernels_ms',{})
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items()}, 'alone', {k:round(v*1e3,1) for k,v in x.items()}, 'verified', d['verified'])"
done; done

#!/bin/bash
# usage: ab_r4.sh ROUNDS "lib[@ENV=V,ENV2=V][:bench args]" ...  - the bench default schedule per variant on ONE box.
#   lib = "-" (in-tree product library), "ab" (in-tree test build), or NAME of profiles/tools/libs/NAME.so
ROUNDS=$1; shift
COMMON="--steps 60 --warmup 5 --repeats 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated --verify-slots 32"
for round in $(seq 1 $ROUNDS); do
for v in "$@"; do
  spec=${v%%:*}; extra=""; [[ "$v" == *:* ]] && extra=${v#*:}
  lib=${spec%%@*}; envs=""; [[ "$spec" == *@* ]] && envs=$(echo "${spec#*@}" | tr ',' ' ')
  echo -n "$v r$round: "
  case "$lib" in
    -) L="MLD_DUMMY=1";;
    ab) L="MLD_HIP_LIBRARY=$PWD/mono_lidar_depth_amd/lib/libmld_hip_ab.so";;
    *) L="MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so";;
  esac
  env $L $envs timeout 300 python bench.py $COMMON $extra 2>gpurun_out/ab_last.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
except Exception as e:
    print('no json', e); sys.exit(0)
r=d['roofline']; x=(r.get('exclusive') or {}).get('kernels_ms',{})
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), [round(v,4) for v in d['timed_loops']['ms_per_step']], {k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items()}, 'alone', {k:round(v*1e3,1) for k,v in x.items()}, 'verified', d['verified'])"
done; done

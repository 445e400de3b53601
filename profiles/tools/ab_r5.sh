#!/bin/bash
# usage: ab_r5.sh ROUNDS lib ...   - per library variant on ONE box: BASELINE config 5 (bench.py --only-config 5: one frame
#   per call and the batched legs) and the config-2 default schedule.  lib = "-" (in-tree product library) or NAME of
#   profiles/tools/libs/NAME.so (mkvariant.sh).  MLD_AB_C2=0 skips the config-2 line.
ROUNDS=$1; shift
C2="--steps 60 --warmup 5 --repeats 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated --verify-slots 32"
for round in $(seq 1 $ROUNDS); do
for lib in "$@"; do
  case "$lib" in
    -) L="MLD_DUMMY=1";;
    *) L="MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so";;
  esac
  echo -n "$lib r$round c5: "
  env $L timeout 600 python bench.py --only-config 5 2>gpurun_out/ab_last.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
except Exception as e:
    print('no json', e); sys.exit(0)
print('frame', round(d['ms_per_frame'],4), end=' ')
for k,v in d['batched'].items():
    print('| S',k, round(v['ms_per_step'],4), 'ms', round(v['associations_per_s']/1e9,3), 'G', {a:round(b*1e3,1) for a,b in v['kernels_ms_per_launch'].items()}, end=' ')
print('verified', d['verified'])"
  if [ "${MLD_AB_C2:-1}" != "0" ]; then
  echo -n "$lib r$round c2: "
  env $L timeout 300 python bench.py $C2 2>gpurun_out/ab_last.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
except Exception as e:
    print('no json', e); sys.exit(0)
r=d['roofline']; x=(r.get('exclusive') or {}).get('kernels_ms',{})
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), [round(v,4) for v in d['timed_loops']['ms_per_step']], {k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items()}, 'alone', {k:round(v*1e3,1) for k,v in x.items()}, 'verified', d['verified'])"
  fi
done; done

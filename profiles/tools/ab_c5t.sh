#!/bin/bash
# usage: ab_c5t.sh ROUNDS lib ...  - per library variant on ONE box: config 5 with 256 sequences (one context / two in turn) and
#   config 2 at k = 7 (two contexts), kernel times of the two-context phase beside the step
ROUNDS=$1; shift
for round in $(seq 1 $ROUNDS); do
for lib in "$@"; do
  case "$lib" in
    -) L="MLD_DUMMY=1";;
    *) L="MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so";;
  esac
  echo -n "$lib r$round 5b256: "
  env $L timeout 300 python bench.py --only-config 5 --leg 256t 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['batched']['256']; t=d['two_contexts']['classify']
print('one', round(d['ms_per_step'],4), 'two', round(t['ms_per_step'],4), {k:round(v*1e3) for k,v in t['kernels_ms_per_launch'].items()}, d['verified'])"
  echo -n "$lib r$round 2k: "
  env $L timeout 300 python bench.py --only-config 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"
done; done

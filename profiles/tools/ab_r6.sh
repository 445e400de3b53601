#!/bin/bash
# usage: ab_r6.sh ROUNDS lib ...   - per library variant on ONE box, the legs round 6 judges the feature kernel on:
#   c2     the bench default (two contexts alternating), compact line
#   2k1    config 2 at k = 7, one context          (k_feature_fused THE dominant kernel)
#   3n     config 3, features near the returns     (same)
#   5b256  config 5, 256 sequences, one context    (DENSE instantiation)   [MLD_AB_T=1: with the two-context schedule too]
# lib = "-" (in-tree product library) or NAME of profiles/tools/libs/NAME.so (mkvariant.sh).  MLD_AB_LEGS="c2 2k1 3n 5b256" selects.
ROUNDS=$1; shift
LEGS=${MLD_AB_LEGS:-"c2 2k1 3n 5b256"}
C2="--steps 60 --warmup 5 --repeats 3 --cpu-seconds 0 --legs none --verify-slots 32"
for round in $(seq 1 $ROUNDS); do
for lib in "$@"; do
  case "$lib" in
    -) L="MLD_DUMMY=1";;
    *) L="MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so";;
  esac
  for leg in $LEGS; do
  echo -n "$lib r$round $leg: "
  case "$leg" in
  c2) env $L timeout 300 python bench.py $C2 --detail gpurun_out/ab_detail.json 2>gpurun_out/ab_last.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
except Exception as e:
    print('no json', e); sys.exit(0)
r=d['roofline']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in r['kernels_ms'].items()}, 'alone', {k:round(v*1e3,1) for k,v in (r.get('kernels_alone_ms') or {}).items()}, 'verified', d['verified'])";;
  2k1) env $L timeout 300 python bench_support/run_legs.py --legs c2k1 2>gpurun_out/ab_last.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['configs']['2']['near_returns']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])";;
  3n) env $L timeout 300 python bench_support/run_legs.py --legs c3n 2>gpurun_out/ab_last.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['configs']['3']['near_returns']['modes']['c0_dispose']
print(round(d['associations_per_s']/1e6,1),'M/s', {k:round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])";;
  5b256) S=256; if [ "${MLD_AB_T:-0}" = 1 ]; then S=256t; fi
    env $L timeout 400 python bench_support/run_legs.py --legs c5b$S 2>gpurun_out/ab_last.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['configs']['5']['batched']['256']; t=(d.get('two_contexts') or {}).get('classify')
print('one', round(d['ms_per_step'],4), {k:round(v*1e3) for k,v in d['kernels_ms_per_launch'].items()}, ('two %.4f %s' % (t['ms_per_step'], {k:round(v*1e3) for k,v in t['kernels_ms_per_launch'].items()})) if t else '', d['verified'])";;
  esac
  done
done; done

#!/bin/bash
# round 6: would the short-list half of a dense frame's features (narrow list <= 8: class 0 of the live queue) run faster on
# the lean instantiation at four wavefronts per SIMD than on the DENSE one at two?  Both kernels on THAT subset only:
# narrow capacity 8 sends every other feature to the wave kernel (whose time is not of interest here).
AB=$PWD/mono_lidar_depth_amd/lib/libmld_hip_ab.so
for round in 1 2; do
for v in "2 48 8 56" "0 48 8 56" "0 40 8 48" "0 32 8 40" "1 48 8 56" "2 48 12 60" "1 48 12 60" "0 48 12 60"; do
  set -- $v
  echo -n "r$round DENSE $1 capacities $2/$3 budget $4: "
  MLD_HIP_LIBRARY=$AB MLD_FORCE_DENSE=$1 MLD_CAP_WIDE=$2 MLD_CAP_NARROW=$3 MLD_KTOTAL=$4 timeout 600 python bench_support/run_legs.py --legs c5b64 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['configs']['5']['batched']['64']
print('step', round(d['ms_per_step'],4), {k:round(v*1e3) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"
done; done

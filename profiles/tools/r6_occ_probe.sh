#!/bin/bash
# round 6, item 4 (a third / fourth wavefront per SIMD for k_feature_fused): what does occupancy buy on the legs where the
# kernel runs alone?  The test build (env switches) with list capacities 32 / 24 (14 KB of LDS per wavefront: 11 per CU),
# 24 / 16 (10 KB: 16 per CU = 4 per SIMD at 105 registers) and 20 / 12 (8 KB) - no feature of these legs overflows either.
AB=$PWD/mono_lidar_depth_amd/lib/libmld_hip_ab.so
for round in 1 2; do
for cap in "32 24" "24 16" "20 12"; do
  set -- $cap
  for leg in c2k1 c3n; do
    echo -n "r$round capacities $1/$2 $leg: "
    MLD_HIP_LIBRARY=$AB MLD_K1MAX=$1 MLD_KMAIN=$2 timeout 300 python bench_support/run_legs.py --legs $leg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['configs']
d = d['2']['near_returns'] if '2' in d else d['3']['near_returns']['modes']['c0_dispose']
print(round(d.get('value', d.get('associations_per_s'))/1e6,1),'M/s', {k:round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"
  done
done; done

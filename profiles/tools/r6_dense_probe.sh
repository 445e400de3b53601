#!/bin/bash
# round 6: what would a third wavefront per SIMD buy the DENSE lane-per-feature kernel (config 5, 256 sequences, one context)?
# The test build with MLD_FORCE_DENSE (2: 235 registers, corner-search tiers to 24; 1: 168 registers, tiers to 16) and
# MLD_KTOTAL (LDS entries per lane for wide + narrow list: 72 = 18 KB = 8 wavefronts per CU ... 52 = 13 KB = 12 per CU;
# features whose lists do not fit go to the wave kernel: its time is part of the step).
AB=$PWD/mono_lidar_depth_amd/lib/libmld_hip_ab.so
for round in 1 2; do
for v in "2 72" "1 72" "1 64" "1 60" "1 56" "1 52" "2 64"; do
  set -- $v
  echo -n "r$round DENSE $1 budget $2: "
  MLD_HIP_LIBRARY=$AB MLD_FORCE_DENSE=$1 MLD_KTOTAL=$2 timeout 400 python bench_support/run_legs.py --legs c5b256 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['configs']['5']['batched']['256']
print('step', round(d['ms_per_step'],4), {k:round(v*1e3) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"
done; done

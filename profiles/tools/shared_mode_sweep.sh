#!/bin/bash
# Wavefronts of k_feature_fused per CU in the shared-GPU mode of the two-context schedule (mld_set_shared_gpu bits 8..15;
# 1 = the default, 8): usage: shared_mode_sweep.sh [n ...]   (default: 8 10 12 14 16)
C2="--steps 60 --warmup 5 --repeats 3 --cpu-seconds 0 --legs none --verify-slots 32 --no-exclusive"
for n in ${@:-8 10 12 14 16}; do
  sm=$((1 + 256 * n))
  echo -n "feature wavefronts per CU $n (shared-mode $sm): "
  timeout 300 python bench.py $C2 --shared-mode $sm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in r['kernels_ms'].items()}, 'verified', d['verified'])"
done

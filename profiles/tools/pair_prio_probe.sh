#!/bin/bash
# (the "LOW priority" rows need profiles/tools/libs/pairprio.so: profiles/patches/r6_pair_low_prio.patch + mkvariant.sh pairprio -DMLD_PAIR_LOW_PRIO)
C2="--steps 60 --warmup 5 --repeats 3 --cpu-seconds 0 --legs none --verify-slots 32 --no-exclusive"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in r['kernels_ms'].items()}, 'verified', d['verified'])"; }
for r in 1 2; do
echo -n "default (hand-over, n=10): "; timeout 300 python bench.py $C2 2>/dev/null | show
for n in 10 12 14 16; do
  sm=$((1 + 256 * n))
  echo -n "pair, normal priority, n=$n: "; timeout 300 python bench.py $C2 --pair --shared-mode $sm 2>/dev/null | show
  echo -n "pair, projection stream LOW priority, n=$n: "; MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/pairprio.so timeout 300 python bench.py $C2 --pair --shared-mode $sm 2>/dev/null | show
done
done

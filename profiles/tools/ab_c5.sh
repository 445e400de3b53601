#!/bin/bash
# bench.py --only-config 5 per profiles/tools/libs/*.so
for lib in profiles/tools/libs/*.so; do
  echo -n "$(basename $lib .so): "
  MLD_HIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --only-config 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_frame'],4), {k:round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()}, d['verified'])"
done

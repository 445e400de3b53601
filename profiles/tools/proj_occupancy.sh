#!/bin/bash
# k_project_scatter ALONE (one context) at capped occupancy: unused dynamic LDS per 256-thread block through the test
# build (libmld_hip_ab.so, MLD_PROJ_LDS).  160 KB per CU: 20 KB -> 8 blocks = 8 waves/SIMD, 26 KB -> 6, 40 KB -> 4,
# 53 KB -> 3, 80 KB -> 2, 160 KB -> 1.
COMMON="--contexts 1 --steps 30 --warmup 3 --repeats 1 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated --verify-slots 2"
for lds in 0 20000 26000 40000 53000 64000 80000; do
  echo -n "MLD_PROJ_LDS=$lds: "
  MLD_PROJ_LDS=$lds MLD_HIP_LIBRARY=$PWD/mono_lidar_depth_amd/lib/libmld_hip_ab.so python bench.py $COMMON 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print({k:round(v.get('avg_ms',0)*1e3,1) for k,v in r['kernels'].items()})"
done

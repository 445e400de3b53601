#!/bin/bash
# usage: latency_ab.sh ROUNDS  - the one-frame latency legs of bench.py (supplied / estimated.ransac / estimated.semantic),
# the shipped library against the test build without the helper thread (MLD_FRAME_HELPER=0) and with MLD_FRAME_COPY=1
# (results through device memory + a D2H copy), same box
ROUNDS=${1:-2}
COMMON="--steps 4 --warmup 1 --repeats 1 --min-timed-seconds 0 --no-estimated --config-frames 0 --streaming-batches 0 --cpu-seconds 0 --frames-per-step 64 --verify-slots 4 --no-exclusive"
for r in $(seq 1 $ROUNDS); do
for v in product nohelper copy; do
  case $v in
    copy) E="MLD_HIP_LIBRARY=$PWD/mono_lidar_depth_amd/lib/libmld_hip_ab.so MLD_FRAME_COPY=1";;
    nohelper) E="MLD_HIP_LIBRARY=$PWD/mono_lidar_depth_amd/lib/libmld_hip_ab.so MLD_FRAME_HELPER=0";;
    *) E="MLD_DUMMY=1";;
  esac
  echo -n "$v r$r: "
  env $E python bench.py $COMMON 2>gpurun_out/lat_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())['latency']
f=lambda x:(round(x['ms_per_frame_median']*1e3,1), round(x['ms_per_frame_p99']*1e3,1), {k:round(v,1) for k,v in (x['host_us_median'] or {}).items()}, 'gpu', {k:round(v,1) for k,v in (x['breakdown_us_median'] or {}).items() if k in ('h2d_us','plane_us','kernels_us','d2h_us','gpu_us')})
print('supplied',f(d)); print('   ransac',f(d['estimated']['ransac']),d['estimated']['ransac']['verified']); print('   semantic',f(d['estimated']['semantic']),d['estimated']['semantic']['verified'])"
done; done

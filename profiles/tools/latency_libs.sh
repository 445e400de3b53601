#!/bin/bash
# usage: latency_libs.sh ROUNDS lib ...  - the one-frame latency legs of bench.py for library variants (profiles/tools/libs/NAME.so;
# "-" = the in-tree product library) on one box: median / p99 of the call and the GPU phases of the instrumented pass
ROUNDS=$1; shift
COMMON="--steps 4 --warmup 1 --repeats 1 --min-timed-seconds 0 --no-estimated --config-frames 0 --streaming-batches 0 --cpu-seconds 0 --frames-per-step 64 --verify-slots 4 --no-exclusive --latency-frames 300"
for r in $(seq 1 $ROUNDS); do
for lib in "$@"; do
  case "$lib" in
    -) E="MLD_DUMMY=1";;
    *) E="MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so";;
  esac
  echo -n "$lib r$r: "
  env $E python bench.py $COMMON 2>gpurun_out/lat_libs.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())['latency']
f=lambda x:(round(x['ms_per_frame_median']*1e3,1), round(x['ms_per_frame_p99']*1e3,1), {k:round(v,1) for k,v in (x['breakdown_us_median'] or {}).items() if k in ('h2d_us','plane_us','kernels_us','gpu_us')})
print('supplied',f(d),'ransac',f(d['estimated']['ransac']),d['estimated']['ransac']['verified'],'semantic',f(d['estimated']['semantic']),d['estimated']['semantic']['verified'], 'process', round(d['process']['one_call']['ms_per_frame_median']*1e3,1))"
done; done

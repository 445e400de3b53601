#!/bin/bash
# round 6: where the projection's time goes beside the 7 TB/s a pure stream of its records reaches (streamread.hip):
# diagnostic builds (WRONG results) without the map atomicMax / the bitmap atomicOr / both, with a plain store instead of the
# atomicMax, and with the point loads alone.  One context: every kernel has the GPU to itself.
mkdir -p gpurun_out
for r in 1 2; do
for lib in ${@:-pd_base pd_nomax pd_noor pd_noatom pd_store pd_loads}; do
  MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so python bench.py --contexts 1 --steps 40 --warmup 3 --repeats 1 --cpu-seconds 0 --legs none --verify-slots 2 2>gpurun_out/pd.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', 'step', round(d['ms_per_step'],4), d['roofline']['kernels_ms'], 'verified', d['verified'])"
done
done

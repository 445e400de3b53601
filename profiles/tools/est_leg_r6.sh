#!/bin/bash
# round 6: the plane-estimated leg (k_rs_batch ahead of the projection) through bench_support/run_legs.py: two contexts with
# half a step each ("halves", the default), whole steps in turn ("alternate"), one context.  Arguments: libraries ("-" = in-tree)
mkdir -p gpurun_out
run() {
python bench_support/run_legs.py --legs estimated --est-steps 40 $1 2>gpurun_out/est.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['plane_estimated']
print('$2 [$1]', 'estimated step', round(e['ms_per_step'],4), {k:round(v*1e3,1) for k,v in e['kernels_ms_per_launch'].items()}, 'frames/launch', e['frame_slots_per_launch'], e['verified'])"
}
LIBS=${@:--}
for r in 1 2; do
for extra in "" "--est-schedule alternate" "--contexts 1" ${MLD_EST_EXTRA:+"$MLD_EST_EXTRA"}; do
  for lib in $LIBS; do
    if [ "$lib" = "-" ]; then unset MLD_HIP_LIBRARY; else export MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/$lib.so; fi
    run "$extra" "$lib"
  done
done
done

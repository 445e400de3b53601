#!/bin/bash
# round 6: the batch's projection as c launches (test build, MLD_PROJ_CHUNKS): does a workgroup too large to be placed among
# projection blocks (k_rs_batch: 16 wavefronts, 150 KB of LDS, all of a SIMD's registers; k_classify: 16 wavefronts, 63 KB) get
# in at the boundaries?  The plane-estimated leg in both schedules, and the headline step.
mkdir -p gpurun_out
export MLD_HIP_LIBRARY=${MLD_PROBE_LIB:-$PWD/mono_lidar_depth_amd/lib/libmld_hip_ab.so}
est() {
python bench_support/run_legs.py --legs estimated --est-steps 40 $1 2>gpurun_out/est.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['plane_estimated']
print('chunks ${MLD_PROJ_CHUNKS:-1} [$1]', 'estimated step', round(e['ms_per_step'],4), {k:round(v*1e3,1) for k,v in e['kernels_ms_per_launch'].items()}, 'frames/launch', e['frame_slots_per_launch'], e['verified'])"
}
head() {
python bench.py --steps 100 --warmup 5 --repeats 2 --cpu-seconds 0 --legs none $1 2>gpurun_out/est.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('chunks ${MLD_PROJ_CHUNKS:-1} headline [$1]', round(d['ms_per_step'],4), d['roofline']['kernels_ms'], d['verified'])"
}
for c in ${@:-1 4 8 16}; do
  export MLD_PROJ_CHUNKS=$c
  est ""
  est "--est-schedule alternate"
  [ -n "$MLD_PROBE_EST_ONLY" ] || head ""
  [ -n "$MLD_PROBE_EST_ONLY" ] || head "--handover projection"
done

#!/bin/bash
# round 6: parity sweeps beyond the slice the suite runs (tests/test_sweep_gpu.py), on the final kernels; seeds no earlier
# round has used.  Output: gpurun_out/r6_random_sweep.txt (-> profiles/r6_random_sweep.txt)
OUT=gpurun_out/r6_random_sweep.txt
mkdir -p gpurun_out
: > $OUT
T=profiles/tools
run() { echo "# $*" >> $OUT; timeout 1500 python "$@" 2>/dev/null | grep -v "^/opt/amdgpu" >> $OUT; echo >> $OUT; }
run $T/random_sweep.py 30000 3000 fused
run $T/random_sweep.py 30000 1500 default
run $T/random_sweep.py 30000 1500 wave-only
run $T/random_sweep.py 30000 1500 dense
run $T/random_sweep.py 40000 300 dense dense128
run $T/random_sweep.py 40000 300 default dense128
run $T/random_sweep.py 40000 300 fused dense128
run $T/random_sweep_batch.py 50000 600 5
run $T/random_sweep_estimate.py 60000 2000
run $T/random_sweep_tracklets.py 70000 250
cat $OUT

#!/bin/bash
# round 6: the parity sweeps once more on the FINAL kernels (after LAB.md 6.13-6.16), fresh seeds
OUT=gpurun_out/r6_random_sweep_final.txt
mkdir -p gpurun_out
: > $OUT
T=profiles/tools
run() { echo "# $*" >> $OUT; timeout 1500 python "$@" 2>/dev/null | grep -v "^/opt/amdgpu" >> $OUT; echo >> $OUT; }
run $T/random_sweep.py 80000 2500 fused
run $T/random_sweep.py 80000 1200 default
run $T/random_sweep.py 80000 1200 wave-only
run $T/random_sweep.py 80000 1200 dense
run $T/random_sweep.py 85000 200 dense dense128
run $T/random_sweep.py 85000 200 default dense128
run $T/random_sweep.py 85000 200 fused dense128
run $T/random_sweep_batch.py 86000 400 5
run $T/random_sweep_estimate.py 87000 1500
run $T/random_sweep_tracklets.py 88000 150
cat $OUT

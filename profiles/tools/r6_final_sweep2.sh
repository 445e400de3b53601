#!/bin/bash
# round 6: once more after the short tiers of LAB.md 6.20 (the last kernel change of the round), fresh seeds
OUT=gpurun_out/r6_random_sweep_final2.txt
mkdir -p gpurun_out
: > $OUT
T=profiles/tools
run() { echo "# $*" >> $OUT; timeout 1500 python "$@" 2>/dev/null | grep -v "^/opt/amdgpu" >> $OUT; echo >> $OUT; }
run $T/random_sweep.py 95000 2000 fused
run $T/random_sweep.py 95000 1000 default
run $T/random_sweep.py 95000 1000 dense
run $T/random_sweep.py 97000 150 fused dense128
run $T/random_sweep_batch.py 98000 300 5
run $T/random_sweep_tracklets.py 99000 100
cat $OUT

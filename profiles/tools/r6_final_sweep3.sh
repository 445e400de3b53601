#!/bin/bash
# round 6, closing: more seeds on the final tree (kernel sources unchanged since LAB.md 6.20); part a / b = two GPU calls
OUT=gpurun_out/r6_random_sweep_final3$1.txt
mkdir -p gpurun_out
: > $OUT
T=profiles/tools
run() { echo "# $*" >> $OUT; timeout 1500 python "$@" 2>/dev/null | grep -v "^/opt/amdgpu" >> $OUT; echo >> $OUT; }
if [ "$1" = "a" ]; then
run $T/random_sweep.py 110000 9000 fused
run $T/random_sweep.py 110000 3000 default
run $T/random_sweep_estimate.py 119000 2500
else
run $T/random_sweep.py 120000 3000 dense
run $T/random_sweep.py 124000 400 fused dense128
run $T/random_sweep_batch.py 125000 900 5
run $T/random_sweep_tracklets.py 127000 300
run $T/random_sweep.py 128000 3000 wave-only
fi
cat $OUT

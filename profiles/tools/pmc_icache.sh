#!/bin/bash
# instruction-cache behaviour of the kernels (one context: rocprofv3 serialises the kernels in counter passes)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--contexts 1 --steps 4 --warmup 1 --repeats 1 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-kernel-timing --no-estimated"
rocprofv3 -L > $OUT/counters.txt 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py $A > $OUT/b1.json 2> $OUT/p1.log
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py $A > $OUT/b2.json 2> $OUT/p2.log
cd $REPO
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        if "mld" in k:
            print(p, k[:60], {c: round(sum(v)/len(v)/1e6,3) for c, v in acc[k].items()}, "(millions)")
PY
grep -i "icache\|ifetch" $OUT/counters.txt | head -20
tail -2 $OUT/p1.log

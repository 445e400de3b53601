#!/bin/bash
# PMC passes over the feature kernels: L1->L2 request count / latency, L2 hit rate, TA busy
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcf_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --no-kernel-timing"
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py $A > $OUT/b1.json 2> $OUT/p1.log
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py $A > $OUT/b2.json 2> $OUT/p2.log
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY --output-format csv -d $OUT/p3 -- python3 $REPO/bench.py $A > $OUT/b3.json 2> $OUT/p3.log
cd $REPO
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2","p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        if "feature" in k or "project_scatter" in k:
            print(p, k, {c: round(sum(v)/len(v),1) for c, v in acc[k].items()})
PY
tail -3 $OUT/p1.log $OUT/p2.log $OUT/p3.log

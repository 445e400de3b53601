#!/bin/bash
# SQ instruction-mix / issue counters of the feature kernels.  usage: pmc_sq.sh <tag> [env assignments...]
TAG=${1:-x}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcsq_$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --no-kernel-timing"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py $A > $OUT/b1.json 2> $OUT/p1.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAIT_ANY --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py $A > $OUT/b2.json 2> $OUT/p2.log
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/p3 -- python3 $REPO/bench.py $A > $OUT/b3.json 2> $OUT/p3.log
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
        if "feature" in k or "classify" in k or "sort" in k:
            print(p, k, {c: round(sum(v)/len(v)/1e6,2) for c, v in acc[k].items()}, "(millions)")
PY

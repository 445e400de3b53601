#!/bin/bash
# counters of the DENSE lane-per-feature kernel (config 5, 256 sequences per step, one context): issue, waits, instruction cache
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_dense
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
L="$REPO/bench_support/run_legs.py --legs c5b256"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $OUT/p1 -- python3 $L > $OUT/b1.json 2> $OUT/p1.log
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p2 -- python3 $L > $OUT/b2.json 2> $OUT/p2.log
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p3 -- python3 $L > $OUT/b3.json 2> $OUT/p3.log
cd $REPO
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2","p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_feature" in k or "k_classify" in k or "k_project" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(p, k, {c: round(sum(v) / len(v)) for c, v in sorted(acc[k].items())}, "launches", len(next(iter(acc[k].values()))))
PY

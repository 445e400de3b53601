#!/bin/bash
# Dynamic instruction counts per kernel launch (1024 frames, one context).  usage: pmc_insts.sh [lib.so ...]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS="$@"; [ -z "$LIBS" ] && LIBS=$REPO/mono_lidar_depth_amd/lib/libmld_hip.so
cd /tmp && export TMPDIR=/tmp
A="--contexts 1 --steps 3 --warmup 1 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-kernel-timing --no-estimated --verify-slots 0"
for lib in $LIBS; do
  OUT=$REPO/gpurun_out/pmci_$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
  export MLD_HIP_LIBRARY=$(cd $REPO && realpath $lib)
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py $A > $OUT/b1.json 2> $OUT/p1.log
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mld::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if k.startswith("k_project_scatter") or k.startswith("k_feature_fused") or k.startswith("k_classify"):
        print("$(basename $lib .so)", k, {c.replace("SQ_", ""): round(sum(v) / len(v) / 1e6, 1) for c, v in acc[k].items()}, "(millions)")
PY
done

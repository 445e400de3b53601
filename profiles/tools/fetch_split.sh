#!/bin/bash
# Where do the feature kernel's fetches from memory come from?  TCC_EA0_RDREQ (64-byte requests beyond L2) and L2 requests of
# k_feature_fused per 1024-frame launch, one context, for the product kernel and the diagnostic builds without the key gathers,
# without the point gathers, without both (wrong results; libs built with mkvariant.sh NAME -DMLD_AB_SWITCHES -DMLD_DIAG_NO_KEYS ...)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
A="--contexts 1 --steps 3 --warmup 1 --repeats 1 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-kernel-timing --no-estimated --verify-slots 0"
for lib in cur nokeys nopoints nogathers; do
  OUT=$REPO/gpurun_out/fsplit_$lib; rm -rf $OUT; mkdir -p $OUT
  export MLD_HIP_LIBRARY=$REPO/profiles/tools/libs/$lib.so
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py $A > $OUT/b1.json 2> $OUT/p1.log
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mld::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if k.startswith("k_feature_fused") or k.startswith("k_classify") or k.startswith("k_project_scatter"):
        print("$lib", k, {c.replace("_sum", ""): round(sum(v) / len(v) / 1e6, 2) for c, v in acc[k].items()}, "(millions per launch)")
PY
done

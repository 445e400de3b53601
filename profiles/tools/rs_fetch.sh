#!/bin/bash
# HBM bytes k_rs_batch fetches per sample point (FETCH_SIZE / TCC_EA0_RDREQ of a 1024-slot launch, every slot's cloud in
# its own memory); argument: library ("-" = in-tree)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_rs_fetch
mkdir -p $OUT
if [ "${1:--}" != "-" ]; then export MLD_HIP_LIBRARY=$REPO/profiles/tools/libs/$1.so; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/p1 -- python3 $REPO/profiles/tools/rs_batch_only.py > $OUT/b1.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    if "rs_batch" in k:
        v = {c: sum(x) / len(x) for c, x in acc[k].items()}
        print(k, {c: round(x / 1e6, 2) for c, x in v.items()}, "(millions per launch; FETCH_SIZE in KB)")
        print("  per sample point (1024 x 6000):", round(v.get("FETCH_SIZE", 0) * 1024 / 6.144e6, 1), "bytes fetched,", round(v.get("TCC_EA0_RDREQ_sum", 0) / 6.144e6, 2), "read requests")
PY

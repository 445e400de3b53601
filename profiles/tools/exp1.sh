#!/bin/bash
# round-2 experiment 1: GPU suite, baseline bench, phase stamps, counter list
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp1
O=gpurun_out/exp1
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $O/pytest.log
B="--steps 20 --warmup 3 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0"
timeout 300 python bench.py $B > $O/bench_slab.json 2> $O/bench_slab.err
MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/stamps.so timeout 300 python profiles/tools/stamps.py > $O/stamps.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -i -c "" $O/counters.txt
grep -i "utcl\|tlb\|TCC_HIT\|TCC_MISS\|TCP_TCC_READ\|TCP_PENDING\|TA_BUSY\|TCP_TOTAL_CACHE" $O/counters.txt | head -40
cat $O/pytest.log; cat $O/stamps.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/exp1/bench_slab.json').read()); r=d['roofline']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],4), {k:round(v['avg_ms']*1e3,1) for k,v in r['kernels'].items()}, 'sort', r['k_sort_features_ms'], 'wave', r['k_feature_wave_ms'])
PY

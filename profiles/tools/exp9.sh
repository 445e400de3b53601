#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp9
( time python bench.py > gpurun_out/exp9/bench_default.json 2> gpurun_out/exp9/bench_default.err ) 2>&1 | grep real
echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/exp9/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value', round(d['value']/1e6,1), 'ms/step', round(d['ms_per_step'],4), 'verified', d['verified'], d['verification'])
print('roofline', {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k not in ('kernels','traffic_profile')})
print('kernels', json.dumps(r['kernels'])[:1500])
print('estimated', d.get('plane_estimated')); print('cpu', d['cpu_baseline'] and round(d['cpu_baseline']['value']/1e6,2), 'latency', d['latency'] and d['latency']['ms_per_frame_median'], 'streaming', d['streaming'] and round(d['streaming']['frames_per_s']))
print('config3', json.dumps(d['configs'].get('3'))[:2500])
print('config5', json.dumps(d['configs'].get('5'))[:2500])
PY
tail -5 gpurun_out/exp9/bench_default.err
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -5

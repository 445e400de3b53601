#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python profiles/latency_mode.py 2>&1 | grep -v amdgpu | tail -10
python bench.py --only-config 5 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config5', round(d['ms_per_frame'],4), 'ms/frame', round(d['associations_per_s']/1e6,1), 'M/s', d['kernels_ms_per_launch'], d['verified'])"

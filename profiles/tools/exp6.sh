#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python - <<'PY'
import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from mono_lidar_depth_amd import capi, synth, GroundPlane
from helpers import make_estimator, run_oracle
P = capi.params_c0()
worst = 0
for seed in range(8):
    cloud = synth.make_cloud(synth.HDL64, seed=seed, frame=seed*3); uv = synth.make_features(4000, seed=seed); plane = synth.make_ground_plane(cloud)
    est = make_estimator(P); d,t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    _, (d0,t0) = run_oracle(P, cloud, uv, plane)
    assert np.array_equal(t,t0), (seed, (t!=t0).sum())
    r = t0==16
    worst = max(worst, np.abs(d[r]-d0[r]).max())
print("road-path max |depth - oracle| over 8 frames x 4000 features:", worst)
PY
MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/stamps.so timeout 300 python profiles/tools/stamps.py 2>&1 | grep -v amdgpu.ids
bash profiles/tools/ab2.sh 2

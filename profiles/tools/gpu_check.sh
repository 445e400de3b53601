python -m pytest tests -q -m gpu --tb=short -x 2>&1 | tail -3
python - <<'PY'
import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from mono_lidar_depth_amd import capi, synth, GroundPlane
from helpers import make_estimator, run_oracle
P = capi.params_c0()
worst = 0
for seed in range(6):
    cloud = synth.make_cloud(synth.HDL64, seed=seed, frame=seed*3); uv = synth.make_features(4000, seed=seed); plane = synth.make_ground_plane(cloud)
    est = make_estimator(P); d,t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    _, (d0,t0) = run_oracle(P, cloud, uv, plane)
    assert np.array_equal(t,t0)
    r = t0==16
    worst = max(worst, np.abs(d[r]-d0[r]).max())
print("road-path max |depth - oracle| over 6 frames:", worst)
PY
python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1),'M/s', {k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})"

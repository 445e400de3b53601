python - <<'PY'
import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from test_golden import load_case
from mono_lidar_depth_amd import CameraPinhole, DepthEstimator, GroundPlane
g = load_case("tight")
cam = CameraPinhole(g["cam"].width, g["cam"].height, g["cam"].focal_length, g["cam"].principal_point_x, g["cam"].principal_point_y)
est = DepthEstimator(device=0); est.InitConfig(g["P"]); est.Initialize(cam, g["T"])
d, t = est.CalculateDepth(g["cloud"], g["uv"], GroundPlane(*g["plane"]))
bad = np.nonzero((t != g["type"]) | ~np.isclose(d, g["depth"], rtol=0, atol=1e-4, equal_nan=True))[0]
for i in bad: print(i, g["uv"][i], "gpu", t[i], d[i], "gold", g["type"][i], g["depth"][i])
off = g["road_pos_off"]
for i in bad: print("road inliers", off[i+1]-off[i], "road nb", g["road_idx_off"][i+1]-g["road_idx_off"][i])
PY

import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mono_lidar_depth_amd import capi, synth, GroundPlane, CameraPinhole
from helpers import make_estimator, run_oracle
W, H = 1, 64
P0 = capi.params_c0().replace(pixelarea_search_witdh=4, pixelarea_search_height=4)
cam = CameraPinhole(W, H, 0.6 * max(W, H), W / 2.0, H / 2.0)
cloud = synth.make_cloud(synth.HDL64_KITTI, seed=52, frame=0)
plane = synth.make_ground_plane(cloud)
uv = synth.make_features(500, seed=52, width=W, height=H)
est = make_estimator(P0, camera=cam)
d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
ref, (d0, t0) = run_oracle(P0, cloud, uv, plane, camera=cam)
bad = np.nonzero(t != t0)[0]
print("differ:", len(bad), "of", len(t), bad[:10])
P = P0.replace(treshold_depth_enabled=0, treshold_depth_local_enabled=0, do_use_cut_behind_camera=0)
est2 = make_estimator(P, camera=cam)
d2, t2 = est2.CalculateDepth(cloud, uv, GroundPlane(*plane))
ref2, (d02, t02) = run_oracle(P, cloud, uv, plane, camera=cam)
for i in bad[:10]:
    tr = ref2.trace_feature(uv[i, 0], uv[i, 1])
    print(i, "uv", uv[i], "gpu", t2[i], d2[i], "oracle", t02[i], d02[i], "n_inl", len(tr["road_pos"]), "nwide", len(tr["road_idx"]), "plane", tr["plane_n"], tr["plane_offset"])
r = t02 == 16
print("all road: max diff", np.nanmax(np.abs(d2[r] - d02[r])), "types equal", np.array_equal(t2, t02))

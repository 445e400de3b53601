#!/usr/bin/env python3
"""One frame at a time through the host-pointer entry points (the reference's ROS usage): PCIe-inclusive latency.

Per frame: mld_set_cloud (H2D 2.1 MB + projection) + mld_set_ground_plane (inlier list H2D + mask build) +
mld_calculate_depth (uv H2D, kernels, depth/type D2H, synchronise).  Reports median / p99 ms per frame and
associations/s; this is NOT the throughput number of bench.py (inputs resident in HBM, many frames per launch).
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mono_lidar_depth_amd import CameraPinhole, DepthEstimator, GroundPlane, capi, synth  # noqa: E402

cam = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV)
P = capi.params_c0()
clouds = [synth.make_cloud(synth.HDL64, seed=0, frame=f) for f in range(8)]
planes = [synth.make_ground_plane(c) for c in clouds]
uvs = [synth.make_features(2000, seed=f) for f in range(8)]
est = DepthEstimator(device=0, max_points=clouds[0].shape[0], max_features=2000)
est.InitConfig(P)
est.Initialize(cam, synth.T_CAM_LIDAR)
for variant in ("plane given as index list", "plane = 6000-point sample (RansacPlane-sized)", "no plane"):
    ts = []
    for it in range(220):
        i = it % 8
        if variant.startswith("plane given"):
            gp = GroundPlane(*planes[i])
        elif variant.startswith("plane = 6000"):
            gp = GroundPlane(planes[i][0], planes[i][1][::max(1, planes[i][1].size // 6000)])
        else:
            from mono_lidar_depth_amd import NO_PLANE
            gp = NO_PLANE
        t0 = time.perf_counter()
        d, t = est.CalculateDepth(clouds[i], uvs[i], gp)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[20:]) * 1e3
    print(f"{variant:48s} median {np.median(ts):.3f} ms  p99 {np.percentile(ts, 99):.3f} ms  "
          f"-> {2000 / np.median(ts) * 1e3 / 1e6:.2f} M assoc/s, {(t == 1).sum()} + {(t == 16).sum()} depths")

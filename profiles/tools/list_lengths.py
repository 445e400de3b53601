#!/usr/bin/env python3
"""Joint distribution of the narrow (k1) and road-window (k2) list lengths of the LIVE features on the bench workloads (CPU
only: the oracle's pixel map + integral image) - what a list capacity (LDS per wavefront of k_feature_fused) would send to
the wave kernel.  usage: list_lengths.py [frames=4]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mono_lidar_depth_amd import CameraPinhole, capi, synth, traffic  # noqa: E402
from oracle import oracle  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
P = capi.params_c0()
cam = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV)
T = synth.T_CAM_LIDAR
w, h = P.pixelarea_search_witdh, P.pixelarea_search_height
rng = np.random.default_rng(5)


def feats(kind, cloud, it):
    if kind == "c2":
        return synth.make_features(2000, seed=it)
    if kind == "c2k":
        return synth.make_features_k_neighbours(cloud, 2000, seed=it, min_neighbours=6, window=(w, h))
    if kind == "c3n":
        return synth.make_features_near_points(cloud, 5000, seed=it)
    u0 = rng.integers(0, cam.width, 10000).astype(np.float64)   # config 5: integer pixels, lower three quarters
    v0 = rng.integers(100, cam.height, 10000).astype(np.float64)
    return np.stack([u0, v0], axis=1)


for kind, scanner in (("c2", synth.HDL64), ("c2k", synth.HDL64), ("c3n", synth.VLP16), ("c5", synth.DENSE128)):
    K1, K2 = [], []
    for it in range(frames):
        cloud = synth.make_cloud(scanner, seed=it, frame=it)
        ref = oracle.OracleDepthEstimator(P, cam.as_struct(), T)
        ref.set_cloud(cloud)
        pm = ref.pixel_map().reshape(cam.height, cam.width)
        uv = feats(kind, cloud, it)
        k1 = traffic.neighbour_counts(pm, uv, w * 0.5, h * 0.5)
        k2 = traffic.neighbour_counts(pm, uv, w * 0.5 * 2.0, h * 0.5 * 1.5)
        live = k1 >= max(P.radiusSearch_count_min, 1)
        K1.append(k1[live])
        K2.append(k2[live])
    k1, k2 = np.concatenate(K1), np.concatenate(K2)
    print(f"{kind}: {k1.size} live features; k1 mean {k1.mean():.1f} p99 {np.percentile(k1, 99):.0f} max {k1.max()}; "
          f"k2 mean {k2.mean():.1f} p99 {np.percentile(k2, 99):.0f} max {k2.max()}; k1+k2 p99 {np.percentile(k1 + k2, 99):.0f} max {(k1 + k2).max()}")
    for name, ovf in (("32 / 24 separate (round 5)", (k2 > 32) | (k1 > 24)), ("48 / 24 separate (round 5, dense)", (k2 > 48) | (k1 > 24)),
                      ("24 / 16 separate", (k2 > 24) | (k1 > 16)),
                      ("shared 40 (k2 <= 32)", (k2 > 32) | (k1 + k2 > 40)), ("shared 36", (k2 > 32) | (k1 + k2 > 36)),
                      ("shared 32", (k1 + k2 > 32)),
                      ("shared 52 (k2 <= 48)", (k2 > 48) | (k1 + k2 > 52)), ("shared 48", (k1 + k2 > 48)),
                      ("shared 64 (k2 <= 48)", (k2 > 48) | (k1 + k2 > 64))):
        print(f"    {name:36s} overflow {100.0 * ovf.mean():6.3f} %")

#!/usr/bin/env python3
"""A sweep seed whose road depth left the tolerance on the lane-per-feature route: the features with the largest
|depth - oracle|, and for each the conditioning of its M-estimator fit (eigenvalues of the weighted scatter of the oracle's
inliers, ray / normal cosine) and the error estimate finish_road_fast would form from them.
usage: road_mismatch_probe.py seed [route]"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
seed = int(sys.argv[1])
route = sys.argv[2] if len(sys.argv) > 2 else "fused"
if route != "default":
    os.environ["MLD_FORCE_WAVE_PATH" if route == "wave-only" else "MLD_FORCE_THREAD_PATH"] = "1"
    from mono_lidar_depth_amd import capi
    capi._lib = capi.load_ab()
from mono_lidar_depth_amd import GroundPlane, synth  # noqa: E402
from helpers import make_estimator, run_oracle  # noqa: E402
from test_randomized_gpu import _random_setup  # noqa: E402

P, cam, T, scanner, kw = _random_setup(seed)
print("seed", seed, "scanner", scanner, "camera", cam.width, cam.height, round(cam.focal_length, 1))
print({k: kw[k] for k in ("pixelarea_search_witdh", "pixelarea_search_height", "treshold_depth_enabled", "treshold_depth_local_enabled",
                          "plane_estimator_use_mestimator", "ransac_plane_point_distance_treshold", "do_use_histogram_segmentation")})
cloud = synth.make_cloud(scanner, seed=200 + seed, frame=seed % 5)
uv = synth.make_features(900, seed=300 + seed, width=cam.width, height=cam.height)
plane = synth.make_ground_plane(cloud)
est = make_estimator(P, camera=cam, T=T)
d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
lane, handed = est.pathCounts() if hasattr(est, "pathCounts") else (None, None)
ref, (d0, t0) = run_oracle(P, cloud, uv, plane, camera=cam, T=T)
print("types equal", np.array_equal(t, t0), "path counts", lane, handed)
diff = np.where(np.isfinite(d) & np.isfinite(d0), np.abs(d - d0), 0.0)
vis = ref.point_index()
cam_pts = ref.cloud_camera_cs()
cam_pts = cam_pts if cam_pts.shape[1] == 3 else cam_pts.T
c = np.asarray(plane[0], dtype=np.float32)
n32 = c[:3].astype(np.float64)
pn, po = n32 / np.sqrt(n32 @ n32), float(c[3])
for i in np.argsort(-diff)[:6]:
    tr = ref.trace_feature(*uv[i])
    raw = [int(vis[tr["road_idx"][p]]) for p in tr["road_pos"]]
    line = f"feature {i}: type {t[i]} / {t0[i]} depth {d[i]:.9f} oracle {d0[i]:.9f} diff {diff[i]:.3e}; {len(raw)} inliers"
    if len(raw) >= 3:
        X = cam_pts[raw]
        w = 1.0 / np.abs(X @ pn + po)
        ctr = (X * w[:, None]).sum(0) / w.sum()
        M = (X - ctr) * np.sqrt(w)[:, None]
        sv = np.linalg.svd(M, compute_uv=False)
        ev = np.sort(sv ** 2)
        U = np.linalg.svd(M.T, full_matrices=False)[0]
        n = U[:, -1]
        ray = ref.viewing_ray(*uv[i])
        nd = abs(float(n @ ray))
        gap = (ev[1] - ev[0]) / ev[2]
        est_err = abs(d0[i]) * 1e-14 / (gap * nd) if gap * nd > 0 else float("inf")
        line += (f"; weights {w.min():.3g} .. {w.max():.3g}; eigenvalues {ev[0]:.3e} {ev[1]:.3e} {ev[2]:.3e}; gap {gap:.3e}; "
                 f"|n.ray| {nd:.3e}; estimate {est_err:.3e}; |centre| {np.abs(ctr).max():.1f}")
    print(line)
est.close()

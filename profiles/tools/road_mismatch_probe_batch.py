#!/usr/bin/env python3
"""road_mismatch_probe.py for a seed of the BATCHED sweep (tests/sweeps.py:check_batch: a launch set of 5 ragged frames through
the shipped library): the worst road depths per frame with the conditioning of their fits.
usage: road_mismatch_probe_batch.py seed"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import torch  # noqa: E402

from mono_lidar_depth_amd import synth  # noqa: E402
from helpers import make_estimator, run_oracle  # noqa: E402
from sweeps import mask_of  # noqa: E402
from test_randomized_gpu import _random_setup  # noqa: E402

seed, B = int(sys.argv[1]), 5
dev = torch.device("cuda:0")
P, cam, T, scanner, kw = _random_setup(seed)
print("seed", seed, "scanner", scanner, "camera", cam.width, cam.height, round(cam.focal_length, 1))
print({k: kw[k] for k in ("pixelarea_search_witdh", "pixelarea_search_height", "treshold_depth_enabled", "treshold_depth_local_enabled",
                          "plane_estimator_use_mestimator", "ransac_plane_point_distance_treshold", "do_use_histogram_segmentation")})
clouds = [synth.make_cloud(scanner, seed=200 + seed, frame=(seed + b) % 7) for b in range(B)]
uvs = [synth.make_features(600 + 97 * b, seed=300 + seed + 1000 * b, width=cam.width, height=cam.height) for b in range(B)]
planes = [synth.make_ground_plane(c) for c in clouds]
est = make_estimator(P, camera=cam, T=T, max_frames=B, max_features=max(u.shape[0] for u in uvs))
t_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
t_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
t_masks = [mask_of(p[1], c.shape[0], dev) for p, c in zip(planes, clouds)]
t_depth = [torch.full((u.shape[0],), 7.0, dtype=torch.float64, device=dev) for u in uvs]
t_type = [torch.full((u.shape[0],), -7, dtype=torch.int32, device=dev) for u in uvs]
torch.cuda.synchronize()
batch = est.prepareBatch(t_clouds, t_uvs, t_depth, t_type, np.stack([p[0] for p in planes]), t_masks, stride_bytes=16)
est.runBatch(batch)
est.synchronize()
for b in range(B):
    d, t = t_depth[b].cpu().numpy(), t_type[b].cpu().numpy()
    ref, (d0, t0) = run_oracle(P, clouds[b], uvs[b], planes[b], camera=cam, T=T)
    diff = np.where(np.isfinite(d) & np.isfinite(d0), np.abs(d - d0), 0.0)
    lane, handed = est.pathCounts(b) if hasattr(est, "pathCounts") else (None, None)
    print(f"frame {b}: types equal {np.array_equal(t, t0)}, max diff {diff.max():.3e}, path counts {lane} {handed}")
    if diff.max() < 1e-6:
        continue
    vis = ref.point_index()
    cam_pts = ref.cloud_camera_cs()
    cam_pts = cam_pts if cam_pts.shape[1] == 3 else cam_pts.T
    c = np.asarray(planes[b][0], dtype=np.float32)
    n32 = c[:3].astype(np.float64)
    pn, po = n32 / np.sqrt(n32 @ n32), float(c[3])
    for i in np.argsort(-diff)[:4]:
        tr = ref.trace_feature(*uvs[b][i])
        raw = [int(vis[tr["road_idx"][p]]) for p in tr["road_pos"]]
        line = f"  feature {i}: type {t[i]} / {t0[i]} depth {d[i]:.9f} oracle {d0[i]:.9f} diff {diff[i]:.3e}; {len(raw)} inliers of {len(tr['road_idx'])}"
        if len(raw) >= 3:
            X = cam_pts[raw]
            w = 1.0 / np.abs(X @ pn + po)
            ctr = (X * w[:, None]).sum(0) / w.sum()
            M = (X - ctr) * np.sqrt(w)[:, None]
            ev = np.sort(np.linalg.svd(M, compute_uv=False) ** 2)
            n = np.linalg.svd(M.T, full_matrices=False)[0][:, -1]
            ray = ref.viewing_ray(*uvs[b][i])
            nd = abs(float(n @ ray))
            gap = (ev[1] - ev[0]) / ev[2]
            first = cam_pts[raw[0]]
            cdev = np.abs(ctr - first).max()
            errc = max(1e-14, 8.88e-16 * cdev * np.sqrt(w.sum() / ev.sum()))
            line += (f"; weights {w.min():.3g} .. {w.max():.3g}; eigenvalues {ev[0]:.3e} {ev[1]:.3e} {ev[2]:.3e}; gap {gap:.3e}; |n.ray| {nd:.3e}; "
                     f"estimate {abs(d0[i]) * errc / (gap * nd):.3e}; |centre| {np.abs(ctr).max():.1f}; first inlier {raw[0]} idx order {raw[:6]}")
        print(line)
est.close()

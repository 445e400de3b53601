#!/usr/bin/env python3
"""BASELINE config 5 through the tracklet API: 128x4096 cloud (524 288 points), 10 000 tracks, 10 % new per frame.
Host-pointer entry point (mld_tracklets_depth: cloud + tracks copied in, depths copied out, previous frame served
from its resident slot).  Prints ms per frame and associations/s.  Run on the GPU box."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mono_lidar_depth_amd import CameraPinhole, GroundPlane, TrackletDepthModule, capi, synth  # noqa: E402

P = capi.params_c0()
cam = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV)
mod = TrackletDepthModule(P, cam, synth.T_CAM_LIDAR)
rng = np.random.default_rng(1)
n_tracks = 10000
clouds = [synth.make_cloud(synth.DENSE128, seed=2, frame=f) for f in range(6)]
planes = [synth.make_ground_plane(c) for c in clouds]
ids = np.arange(n_tracks, dtype=np.int64)
next_id = n_tracks
ts, ta, n_assoc = [], [], 0
for it in range(60):
    i = it % len(clouds)
    # 10 % of the tracks are replaced by new ones every frame
    repl = rng.choice(n_tracks, n_tracks // 10, replace=False)
    ids[repl] = np.arange(next_id, next_id + repl.size)
    next_id += repl.size
    u0 = rng.integers(0, cam.width, n_tracks).astype(np.float32)
    v0 = rng.integers(100, cam.height, n_tracks).astype(np.float32)
    u1 = (u0 + rng.integers(-3, 4, n_tracks)).astype(np.float32)
    v1 = (v0 + rng.integers(-2, 3, n_tracks)).astype(np.float32)
    t0 = time.perf_counter()
    d_cur, d_last, is_new = mod.process(clouds[i], ids, u0, v0, u1, v1, GroundPlane(*planes[i]))
    ts.append(time.perf_counter() - t0)
    ta.append(mod.last_abi_seconds)
    n_assoc = n_tracks + int(is_new.sum())
ts = np.array(ts[10:]) * 1e3
ta = np.array(ta[10:]) * 1e3
print(f"C-ABI calls only (setInputCloud + ground plane + mld_tracklets_depth): median {np.median(ta):.3f} ms/frame, "
      f"p99 {np.percentile(ta, 99):.3f} ms -> {n_assoc / np.median(ta) * 1e3 / 1e6:.1f} M associations/s")
print(f"config 5 (tracklet API, host pointers): median {np.median(ts):.3f} ms/frame, p99 {np.percentile(ts, 99):.3f} ms, "
      f"{n_assoc} associations/frame -> {n_assoc / np.median(ts) * 1e3 / 1e6:.1f} M associations/s "
      f"(includes the Python tracklet-map bookkeeping of {n_tracks} tracks)")

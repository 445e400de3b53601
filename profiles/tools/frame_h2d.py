#!/usr/bin/env python3
"""Where the one-frame call's 75-80 us "h2d" go: the same 2.1 MB cloud copied (a) by torch alone, (b) by mld_set_cloud,
(c) inside mld_calculate_depth_frame with / without features and plane, pinned and pageable sources, page-aligned and
numpy-allocated; GPU-phase breakdown from mld_frame_timing."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mono_lidar_depth_amd import CameraPinhole, DepthEstimator, GroundPlane, NO_PLANE, capi, synth

dev = torch.device("cuda:0")
P = capi.params_c0()
cam = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV)
cloud = synth.make_cloud(synth.HDL64, seed=0, frame=0)
plane = synth.make_ground_plane(cloud)
uv = synth.make_features(2000, seed=1)
pinned = torch.from_numpy(cloud).pin_memory().numpy()
aligned_t = torch.empty(cloud.shape, dtype=torch.float32)   # torch's CPU allocator: 64-byte aligned, fresh pages
aligned_t.copy_(torch.from_numpy(cloud))
aligned = aligned_t.numpy()


def med(fn, n=200):
    ts = []
    for _ in range(n + 20):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts[20:]))


dst = torch.empty(cloud.shape, dtype=torch.float32, device=dev)
for name, src in (("numpy", cloud), ("torch-cpu", aligned), ("pinned", pinned)):
    t = torch.from_numpy(src)

    def f():
        dst.copy_(t, non_blocking=True)
        torch.cuda.synchronize()
    print(f"torch copy, {name} source: {med(f):.1f} us")

est = DepthEstimator(device=0, max_points=cloud.shape[0], max_features=4096)
est.InitConfig(P)
est.Initialize(cam, synth.T_CAM_LIDAR)
lib, ctx = est._lib, est._ctx
for name, src in (("numpy", cloud), ("torch-cpu", aligned), ("pinned", pinned)):
    print(f"mld_set_cloud (copy + projection, synchronous), {name} source: "
          f"{med(lambda: est._check(lib.mld_set_cloud(ctx, 0, src.ctypes.data, src.shape[0], 16))):.1f} us")
uv0 = np.zeros((0, 2))
for name, src in (("numpy", cloud), ("torch-cpu", aligned), ("pinned", pinned)):
    for what, u, gp in (("no features, no plane", uv0, NO_PLANE), ("2000 features, no plane", uv, NO_PLANE),
                        ("2000 features + supplied plane", uv, GroundPlane(*plane))):
        t = med(lambda: est.CalculateDepth(src, u, gp))
        est.timingEnable(True)
        ph = []
        for _ in range(60):
            est.CalculateDepth(src, u, gp)
            ph.append(est.frameTiming())
        est.timingEnable(False)
        b = {k: float(np.median([p[k] for p in ph[5:]])) for k in ("h2d_us", "kernels_us", "gpu_us", "copycall_us", "wait_us", "total_us")}
        print(f"frame call, {name} source, {what}: {t:.1f} us; instrumented: " + ", ".join(f"{k[:-3]} {v:.1f}" for k, v in b.items()))

#!/usr/bin/env python3
"""Times the feature kernels under workload variants to see where the launch time goes (run on the GPU box)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mono_lidar_depth_amd import CameraPinhole, DepthEstimator, capi, synth  # noqa: E402

B, F, U = 256, 2000, 8
dev = torch.device("cuda:0")
cam = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV)
clouds_h = [synth.make_cloud(synth.HDL64, seed=0, frame=f) for f in range(U)]
planes_h = [synth.make_ground_plane(c) for c in clouds_h]
N = clouds_h[0].shape[0]


def mask_of(inl):
    m = np.zeros((N + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
    return m.view(np.int32)


t_clouds = [torch.from_numpy(clouds_h[b % U]).to(dev).clone() for b in range(B)]
t_masks = [torch.from_numpy(mask_of(planes_h[b % U][1])).to(dev).clone() for b in range(B)]
coeffs = np.stack([planes_h[b % U][0] for b in range(B)])


def run(name, P, uv_fn, with_plane=True, steps=10):
    uvs = [torch.from_numpy(uv_fn(b)).to(dev) for b in range(B)]
    depth = [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(B)]
    types = [torch.empty(F, dtype=torch.int32, device=dev) for _ in range(B)]
    torch.cuda.synchronize()
    est = DepthEstimator(device=0, max_frames=B)
    est.InitConfig(P)
    est.Initialize(cam, synth.T_CAM_LIDAR)
    batch = est.prepareBatch(t_clouds, uvs, depth, types, coeffs, t_masks)
    if not with_plane:
        batch["coeffs"] = None
    for _ in range(3):
        _step(est, batch, with_plane)
    est.synchronize()
    est.timingEnable(True)
    est.timingReset()
    for _ in range(steps):
        _step(est, batch, with_plane)
    est.synchronize()
    kp, _ = est.kernelTimeMs(0)
    kf, _ = est.kernelTimeMs(1)
    hist = np.zeros(21, dtype=np.int64)
    for b in range(0, B, 32):
        hist += est.resultHistogram(types[b])
    print(f"{name:34s} project {kp * 1e3:7.1f} us   feature {kf * 1e3:7.1f} us   types {dict((i, int(c)) for i, c in enumerate(hist) if c)}")
    est.close()


def _step(est, b, with_plane):
    import ctypes as C
    lib, ctx, n = est._lib, est._ctx, b["n"]
    est._check(lib.mld_set_clouds_device(ctx, n, b["cloud_ptrs"], b["cloud_n"], b["stride"]))
    if with_plane:
        est._check(lib.mld_set_ground_planes_mask_device(ctx, n, b["coeffs"].ctypes.data_as(C.POINTER(C.c_float)), b["mask_ptrs"]))
    else:
        for i in range(n):
            est._check(lib.mld_set_ground_plane(ctx, i, None, None, 0))
    est._check(lib.mld_calculate_depths_device(ctx, n, b["uv_ptrs"], b["F"], b["depth_ptrs"], b["type_ptrs"]))


P0 = capi.params_c0()
full = lambda b: synth.make_features(F, seed=b)
sky = lambda b: synth.make_features(F, seed=b) * np.array([1.0, 0.3])          # rows 0..112: no lidar points
low = lambda b: synth.make_features(F, seed=b) * np.array([1.0, 0.55]) + np.array([0, 160.0])  # rows 160..366
run("full (config 2)", P0, full)
run("no plane (road off)", P0, full, with_plane=False)
run("sky features only (all type 2)", P0, sky)
run("lower image only", P0, low)
run("lower image, no plane", P0, low, with_plane=False)
run("lower, no hist", P0.replace(do_use_histogram_segmentation=0), low, with_plane=False)
os.environ["MLD_FORCE_WAVE_PATH"] = "1"
run("full, wave path only", P0, full)

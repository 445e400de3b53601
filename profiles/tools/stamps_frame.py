#!/usr/bin/env python3
"""Phase shares of k_feature_wave for ONE frame per call (diagnostic build: mkvariant.sh stamps -DMLD_STAMPS, run with
MLD_HIP_LIBRARY pointing at it).  usage: stamps_frame.py [features] [scanner: HDL64|DENSE128]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mono_lidar_depth_amd import CameraPinhole, DepthEstimator, GroundPlane, capi, synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
scanner = getattr(synth, sys.argv[2] if len(sys.argv) > 2 else "HDL64")
NAMES = {1: "narrow window gather", 2: "histogram", 3: "triangle", 4: "min/max + records", 5: "phase 2: main tail",
         6: "road window gather", 7: "road filter", 8: "min/max + moments", 9: "phase 4: road tail", 12: "stores"}
P = capi.params_c0()
cam = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV)
dev = torch.device("cuda:0")
clouds = [synth.make_cloud(scanner, seed=0, frame=f) for f in range(4)]
planes = [synth.make_ground_plane(c) for c in clouds]
t_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
t_inl = [torch.from_numpy(p[1]).to(dev) for p in planes]
t_uv = [torch.from_numpy(synth.make_features(F, seed=b, integer=scanner is synth.DENSE128)).to(dev) for b in range(4)]
est = DepthEstimator(device=0, max_points=clouds[0].shape[0], max_features=F)
est.InitConfig(P)
est.Initialize(cam, synth.T_CAM_LIDAR)
out = np.zeros((2, 32768, 16), dtype=np.uint32)


def frame(i):
    est.setInputCloud(t_clouds[i % 4], GroundPlane(planes[i % 4][0], t_inl[i % 4]))
    return est.CalculateDepth(t_uv[i % 4])


for i in range(3):
    frame(i)
est.synchronize()
est._lib.mld_debug_read_stamps(C.c_void_p(out.ctypes.data))  # clears
n = 8
est.timingEnable(True)
est.timingReset()
for i in range(n):
    frame(i)
est.synchronize()
est._lib.mld_debug_read_stamps(C.c_void_p(out.ctypes.data))
v = out[1].astype(np.float64)
waves = v[:, 15].sum()
ph = v[:, :15].sum(0)
tot = ph.sum()
print(f"k_feature_wave: {waves / n:.0f} stamped features per frame, {tot / max(1, waves):.0f} cycles per feature "
      f"(kernel {est.kernelTimeMs(3)[0] * 1e3:.1f} us with stamps)")
for i in range(15):
    if ph[i]:
        print(f"   {i:2d} {NAMES.get(i, '?'):24s} {100.0 * ph[i] / tot:5.1f} %   {ph[i] / waves:8.0f} cyc/feature")

"""Where k_rs_batch spends a block's time (diagnostic build -DMLD_DIAG_RS_PHASES, profiles/tools/libs/rsphases.so):
   MLD_HIP_LIBRARY=profiles/tools/libs/rsphases.so python profiles/tools/rs_phases.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mono_lidar_depth_amd import CameraPinhole, DepthEstimator, capi, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = capi.params_c0()
cam, T = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV), synth.T_CAM_LIDAR
dev = torch.device("cuda:0")
U = 16
clouds = [torch.from_numpy(synth.make_cloud(synth.HDL64, seed=1, frame=b)).to(dev) for b in range(U)]
est = DepthEstimator(device=0, max_frames=B, max_features=64)
est.InitConfig(P)
est.Initialize(cam, T)
lib = est._lib
lib.mld_debug_rs_phases.argtypes = [C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 16)()
names = ["sample", "rounds", "inlier list", "partial sums", "combination", "eigenvector", "mask", "plane"]
for it in range(3):
    est.setInputCloudsEstimatePlanes([clouds[b % U] for b in range(B)], list(range(1, B + 1)))
    est.synchronize()
    assert lib.mld_debug_rs_phases(out) == 0
    v = np.array(list(out), dtype=np.float64)
    print("us per block:", {n: round(v[i] / 100.0 / B, 2) for i, n in enumerate(names)},
          "| rounds in detail:", {n: round(v[i] / 100.0 / B, 2) for i, n in ((9, "model"), (10, "distances"), (11, "barrier"), (1, "replay"), (14, "empty marker"))},
          "total", round((v[:8].sum() + v[9:12].sum()) / 100.0 / B, 2), "iterations/block", round(v[8] / B, 1),
          "valid draws of wavefront 0 per block", round(v[13] / B, 2))

"""Three 1024-slot launches of the batched plane estimation, every slot's cloud in its own memory (for counter passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mono_lidar_depth_amd import CameraPinhole, DepthEstimator, capi, synth  # noqa: E402

B, U = 1024, 16
dev = torch.device("cuda:0")
est = DepthEstimator(device=0, max_frames=B, max_features=64)
est.InitConfig(capi.params_c0())
est.Initialize(CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV), synth.T_CAM_LIDAR)
unique = [torch.from_numpy(synth.make_cloud(synth.HDL64, seed=0, frame=b)).to(dev) for b in range(U)]
batch = [unique[b % U].clone() for b in range(B)]
for it in range(3):
    est.setInputCloudsEstimatePlanes(batch, list(range(1, B + 1)))
    est.synchronize()
est.close()

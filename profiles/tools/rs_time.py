# production library: event-timed k_rs_batch for batches of 256 / 512 / 1024 slots, repeated calls (experiment)
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from mono_lidar_depth_amd import CameraPinhole, DepthEstimator, capi, synth
P = capi.params_c0()
cam, T = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV), synth.T_CAM_LIDAR
dev = torch.device("cuda:0")
U = 16
clouds = [torch.from_numpy(synth.make_cloud(synth.HDL64, seed=1, frame=b)).to(dev) for b in range(U)]
for B in (256, 512, 1024):
    est = DepthEstimator(device=0, max_frames=B, max_features=64)
    est.InitConfig(P); est.Initialize(cam, T)
    cl = [clouds[b % U].clone() for b in range(B)]
    seeds = list(range(1, B + 1))
    for it in range(3):
        est.setInputCloudsEstimatePlanes(cl, seeds); est.synchronize()
    est.timingEnable(True); est.timingReset()
    for it in range(10):
        est.setInputCloudsEstimatePlanes(cl, seeds); est.synchronize()
    print(B, "k_rs_batch ms, launches:", est.kernelTimeMs(4), "project:", est.kernelTimeMs(0), flush=True)
    est.close()

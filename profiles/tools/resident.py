"""Builds the device-resident config-2 batch bench.py times (B frame slots, U distinct clouds) for the diagnostic tools."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mono_lidar_depth_amd import CameraPinhole, DepthEstimator, capi, synth  # noqa: E402


def build(B=1024, F=2000, U=16, scanner=None, P=None, seed=0, integer_uv=False, uv_fn=None, list_capacity=None):
    dev = torch.device("cuda:0")
    scanner = scanner or synth.HDL64
    P = P or capi.params_c0()
    cam = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV)
    clouds_h = [synth.make_cloud(scanner, seed=seed, frame=f) for f in range(U)]
    planes_h = [synth.make_ground_plane(c) for c in clouds_h]
    N = clouds_h[0].shape[0]
    words = (N + 31) // 32

    def mask_of(inl):
        m = np.zeros(words, dtype=np.uint32)
        np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
        return m.view(np.int32)

    all_clouds = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    all_masks = torch.empty((B, words), dtype=torch.int32, device=dev)
    all_uvs = torch.empty((B, F, 2), dtype=torch.float64, device=dev)
    all_depth = torch.empty((B, F), dtype=torch.float64, device=dev)
    all_type = torch.empty((B, F), dtype=torch.int32, device=dev)
    du = [torch.from_numpy(c).to(dev) for c in clouds_h]
    mu = [torch.from_numpy(mask_of(p[1])).to(dev) for p in planes_h]
    uvs_h = []
    for b in range(B):
        all_clouds[b].copy_(du[b % U])
        all_masks[b].copy_(mu[b % U])
        uv = uv_fn(F, seed * 100000 + b) if uv_fn else synth.make_features(F, seed=seed * 100000 + b)
        if integer_uv:
            uv = np.floor(uv)
        uvs_h.append(uv)
        all_uvs[b].copy_(torch.from_numpy(uv))
    coeffs = np.stack([planes_h[b % U][0] for b in range(B)])
    torch.cuda.synchronize()
    est = DepthEstimator(device=0, max_frames=B, max_features=F)
    est.InitConfig(P)
    est.Initialize(cam, synth.T_CAM_LIDAR)
    if list_capacity:
        est.setListCapacity(*list_capacity)
    batch = est.prepareBatch([all_clouds[b] for b in range(B)], [all_uvs[b] for b in range(B)],
                             [all_depth[b] for b in range(B)], [all_type[b] for b in range(B)], coeffs,
                             [all_masks[b] for b in range(B)], stride_bytes=16)
    return {"est": est, "batch": batch, "clouds_h": clouds_h, "planes_h": planes_h, "uvs_h": uvs_h, "P": P, "cam": cam,
            "depth": all_depth, "type": all_type, "N": N, "F": F, "B": B}

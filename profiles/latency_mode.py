#!/usr/bin/env python3
"""Where one frame per call (the reference's ROS usage) spends its time: copies vs kernels vs call overhead.
Run on the GPU box: python profiles/latency_mode.py"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mono_lidar_depth_amd import CameraPinhole, DepthEstimator, GroundPlane, capi, synth  # noqa: E402

P = capi.params_c0()
cam = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV)
clouds = [synth.make_cloud(synth.HDL64, seed=0, frame=f) for f in range(8)]
planes = [synth.make_ground_plane(c) for c in clouds]
uvs = [synth.make_features(2000, seed=b) for b in range(8)]
N, F = clouds[0].shape[0], 2000
dev = torch.device("cuda:0")


def med(fn, n=200, warm=10):
    ts = []
    for it in range(n + warm):
        t0 = time.perf_counter()
        fn(it)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[warm:]) * 1e3
    return float(np.median(ts)), float(np.percentile(ts, 99))


est = DepthEstimator(device=0, max_points=N, max_features=F)
est.InitConfig(P)
est.Initialize(cam, synth.T_CAM_LIDAR)
print("full host-pointer frame (setInputCloud + plane + CalculateDepth): median %.3f ms  p99 %.3f" %
      med(lambda it: est.CalculateDepth(clouds[it % 8], uvs[it % 8], GroundPlane(*planes[it % 8]))))

# copies alone
d_cloud = torch.empty((N, 4), dtype=torch.float32, device=dev)
h_pinned = torch.from_numpy(clouds[0]).pin_memory()
h_page = torch.from_numpy(clouds[1])


def cp(src):
    d_cloud.copy_(src, non_blocking=True)
    torch.cuda.synchronize()


print("H2D 2.1 MB pageable: median %.3f ms  p99 %.3f" % med(lambda it: cp(h_page)))
print("H2D 2.1 MB pinned:   median %.3f ms  p99 %.3f" % med(lambda it: cp(h_pinned)))
stage = torch.empty((N, 4), dtype=torch.float32).pin_memory()
t0 = time.perf_counter()
for _ in range(200):
    stage.copy_(h_page)
print("host memcpy 2.1 MB pageable -> pinned: %.3f ms" % ((time.perf_counter() - t0) / 200 * 1e3))

# kernels alone: everything device resident, single slot
words = (N + 31) // 32
m = np.zeros(words, dtype=np.uint32)
np.bitwise_or.at(m, planes[0][1] >> 5, (np.uint32(1) << (planes[0][1] & 31).astype(np.uint32)))
d_mask = torch.from_numpy(m.view(np.int32)).to(dev)
d_cl = torch.from_numpy(clouds[0]).to(dev)
d_uv = torch.from_numpy(uvs[0]).to(dev)
d_depth = torch.empty(F, dtype=torch.float64, device=dev)
d_type = torch.empty(F, dtype=torch.int32, device=dev)
lib, ctx = est._lib, est._ctx
co = (C.c_float * 4)(*[float(x) for x in planes[0][0]])
torch.cuda.synchronize()


def dev_frame(it):
    lib.mld_set_cloud_device(ctx, 0, d_cl.data_ptr(), N, 16)
    lib.mld_set_ground_plane_mask_device(ctx, 0, co, d_mask.data_ptr())
    lib.mld_calculate_depth_device(ctx, 0, d_uv.data_ptr(), F, d_depth.data_ptr(), d_type.data_ptr())
    lib.mld_synchronize(ctx)


print("device-resident single frame (5 launches + sync): median %.3f ms  p99 %.3f" % med(dev_frame))
est.timingEnable(True)
est.timingReset()
for it in range(50):
    dev_frame(it)
for k, name in ((0, "project"), (5, "classify"), (1, "fused"), (3, "wave")):
    print("   %-9s %.1f us" % (name, est.kernelTimeMs(k)[0] * 1e3))

# the single-call frame entry point, arguments prepared once (no numpy / ctypes marshalling in the loop)
inl = np.ascontiguousarray(planes[0][1], dtype=np.int32)
uvh = np.ascontiguousarray(uvs[0])
depth = np.empty(F)
types = np.empty(F, dtype=np.int32)
cl = np.ascontiguousarray(clouds[0])
est.timingEnable(False)


def frame_call(it, with_plane=True, nf=F):
    lib.mld_calculate_depth_frame(ctx, 0, cl.ctypes.data, N, 16, co if with_plane else None, inl.ctypes.data if with_plane else None,
                                  inl.size if with_plane else 0, uvh.ctypes.data, nf, depth.ctypes.data, types.ctypes.data)


print("mld_calculate_depth_frame, prepared args:        median %.3f ms  p99 %.3f" % med(frame_call))
print("mld_calculate_depth_frame, no plane:             median %.3f ms  p99 %.3f" % med(lambda it: frame_call(it, False)))
print("mld_calculate_depth_frame, 64 features:          median %.3f ms  p99 %.3f" % med(lambda it: frame_call(it, True, 64)))
pin = torch.from_numpy(clouds[0]).pin_memory()


def frame_pinned(it):
    lib.mld_calculate_depth_frame(ctx, 0, pin.data_ptr(), N, 16, co, inl.ctypes.data, inl.size, uvh.ctypes.data, F,
                                  depth.ctypes.data, types.ctypes.data)


print("mld_calculate_depth_frame, cloud in pinned memory: median %.3f ms  p99 %.3f" % med(frame_pinned))

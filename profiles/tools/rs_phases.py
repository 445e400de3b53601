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
lib.mld_debug_rs_slots.argtypes = [C.POINTER(C.c_uint)]
out = (C.c_ulonglong * 16)()
names = ["sample", "rounds", "inlier list", "partial sums", "combination", "eigenvector", "mask", "plane"]
# every slot its own copy of its cloud, as in bench.py (16 buffers shared by 1024 slots would be served from the caches)
batch = [clouds[b % U].clone() for b in range(B)]
for it in range(3):
    est.setInputCloudsEstimatePlanes(batch, list(range(1, B + 1)))
    est.synchronize()
    assert lib.mld_debug_rs_phases(out) == 0
    v = np.array(list(out), dtype=np.float64)
    print("us per block:", {n: round(v[i] / 100.0 / B, 2) for i, n in enumerate(names)},
          "| rounds in detail:", {n: round(v[i] / 100.0 / B, 2) for i, n in ((9, "valid list"), (10, "distances"), (11, "barrier"), (1, "replay"), (14, "empty marker"), (12, "constants"), (13, "model_of"), (15, "kernel arguments"))},
          "total", round((v[:8].sum() + v[9:16].sum()) / 100.0 / B, 2), "iterations/block", round(v[8] / B, 1),
          "")
    st = (C.c_uint * (4096 * 4))()
    assert lib.mld_debug_rs_slots(st) == 0
    t = np.array(list(st), dtype=np.float64).reshape(4096, 4)[:B]
    dur, end, hw = t[:, 0], t[:, 1], t[:, 3].astype(np.int64)
    start = (end - dur) % 2**32
    t0 = start.min()
    cu = (hw >> 8) & 0xF | ((hw >> 12) & 0x3) << 4 | ((hw >> 13) & 0x7) << 6 | (hw >> 20) << 10  # cu, sh, se, xcc
    print("us per slot: mean", round(dur.mean() / 100, 1), "percentiles 50/90/99/max",
          [round(float(x) / 100, 1) for x in np.percentile(dur, [50, 90, 99, 100])],
          "| kernel span (first start to last end)", round((end.max() - t0) / 100, 1), "us; distinct CUs", len(set(cu.tolist())))
    gaps, per_cu = [], []
    for c in set(cu.tolist()):
        i = np.where(cu == c)[0]
        i = i[np.argsort(start[i])]
        per_cu.append(len(i))
        gaps += [(start[i[j + 1]] - end[i[j]]) / 100 for j in range(len(i) - 1)]
    print("slots per CU min/mean/max", min(per_cu), round(float(np.mean(per_cu)), 2), max(per_cu), "| gap between a block's end and the next block's start on its CU: mean",
          round(float(np.mean(gaps)), 1), "percentiles 50/90/max", [round(float(x), 1) for x in np.percentile(gaps, [50, 90, 100])],
          "| block starts (us after the first): the 64th / 128th / 192nd / 256th / 512th", [round(float(x) / 100, 1) for x in np.sort(start - t0)[[63, 127, 191, 255, 511]]])
    stamps = (C.c_ulonglong * 4)()
    lib.mld_debug_rs_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
    assert lib.mld_debug_rs_stamps(stamps) == 0
    lo = lambda v: float(v % 2**32)  # noqa: E731
    print("stream order: stamp kernel before the launch -> first block start", round(((t0 - lo(stamps[0])) % 2**32) / 100, 1),
          "us; last block end -> stamp kernel after the launch", round(((lo(stamps[1]) - end.max()) % 2**32) / 100, 1), "us")
    xcc = hw >> 20
    print("per XCD: blocks", [int((xcc == x).sum()) for x in range(8)], "| last end (us after the first start)",
          [round(float(end[xcc == x].max() - t0) / 100, 1) for x in range(8)], "| sum of block durations / 32 CUs",
          [round(float(dur[xcc == x].sum()) / 3200, 1) for x in range(8)])
    last_cus = sorted(set(cu.tolist()), key=lambda c: -end[cu == c].max())[:3]
    for c in last_cus:
        i = np.where(cu == c)[0]
        i = i[np.argsort(start[i])]
        print("  CU", c, "blocks (id, start, end):", [(int(j), round(float(start[j] - t0) / 100, 1), round(float(end[j] - t0) / 100, 1)) for j in i])
    first = np.argsort(start)[:256]
    print("the first 256 blocks: mean start per XCD (us)", [round(float((start[first][(hw[first] >> 20) == x] - t0).mean()) / 100, 1) for x in range(8)],
          "| block ids among them: min/max", int(first.min()), int(first.max()))

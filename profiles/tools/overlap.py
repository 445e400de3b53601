#!/usr/bin/env python3
"""Projection (HBM streaming) beside the feature kernels (gather / f64-issue bound): one context; two contexts that
merely share the GPU; two contexts alternating (mld_order_after), with and without mld_set_shared_gpu."""
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from mono_lidar_depth_amd import CameraPinhole, capi, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P = capi.params_c0()
cam = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV)
T = synth.T_CAM_LIDAR


def timeit(fn, res, n):
    for _ in range(3):
        fn()
    res.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    res.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for NC in [int(x) for x in os.environ.get("NCS", "1,2").split(",")]:
    res = bench.Resident(P, cam, T, synth.HDL64, B, 16, 2000, 0, 0, contexts=NC)

    def independent():
        for e, b in res.batches:
            e.runBatch(b)

    for shared in ((False,) if NC == 1 else (False, True)):
        for e in res.ests:
            e.setSharedGpu(shared)
        for name, fn in (("independent", independent),) + ((("alternating", res.run_step),) if NC > 1 else ()):
            ms = timeit(fn, res, steps)
            print(f"contexts={NC} shared_gpu={int(shared)} {name:12s}: {ms:.4f} ms per {B} frames  "
                  f"({B * 2000 / ms / 1e6:.2f} G assoc/s)", flush=True)
    ok, rep = res.verify(3)
    print("   verified", ok, rep["max_abs_depth_diff_m"])
    res.close()
    del res
    torch.cuda.empty_cache()

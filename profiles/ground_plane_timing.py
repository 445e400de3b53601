#!/usr/bin/env python3
"""Per-frame ground-plane estimation: HIP (mld_estimate_ground_plane / mld_estimate_semantic_plane,
cloud already on the device) against the CPU restatement on this host.  Run on the GPU box."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import torch  # noqa: E402
from mono_lidar_depth_amd import capi, synth  # noqa: E402
from helpers import make_estimator, make_oracle  # noqa: E402

P = capi.params_c0()
cloud = synth.make_cloud(synth.HDL64, seed=1, frame=0)
img = synth.make_label_image(cloud)
est = make_estimator(P)
t_cloud = torch.from_numpy(cloud).cuda()
t_img = torch.from_numpy(img).cuda()
est.setInputCloud(t_cloud, None, plane_given=False)
ref = make_oracle(P)
ref.set_cloud(cloud)


def timed(fn, n):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


for iters in (100, 1000, 10000):
    Pi = P.replace(ransac_plane_max_iterations=iters)
    e2 = make_estimator(Pi)
    e2.setInputCloud(t_cloud, None, plane_given=False)
    r2 = make_oracle(Pi)
    r2.set_cloud(cloud)
    g = timed(lambda: e2.estimateGroundPlane(0, 7), 30)
    c = timed(lambda: r2.estimate_ground_plane(7), 5)
    print(f"RANSAC plane, max_iterations={iters:5d}: HIP {g:7.3f} ms   CPU restatement (1 thread) {c:8.2f} ms")
g = timed(lambda: est.estimateSemanticPlane(t_img, (6, 7, 8, 9), 0.1), 30)
c = timed(lambda: ref.estimate_semantic_plane(img, (6, 7, 8, 9), 0.1), 5)
print(f"semantic plane (label image on the device):  HIP {g:7.3f} ms   CPU restatement (1 thread) {c:8.2f} ms")

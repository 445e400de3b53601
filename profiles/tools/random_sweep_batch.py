#!/usr/bin/env python3
"""The parity sweep of random_sweep.py through the BATCHED entry points of the shipped library (frame slots: k_classify +
k_feature_fused + k_feature_wave, plane known at projection time): per random configuration a launch set of B frames
(different frames of the scanner, different feature sets, ragged feature counts) against the oracle (TEST TOOL: the oracle
is the checker).  usage: random_sweep_batch.py first_seed n_seeds [frames_per_set=5]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from mono_lidar_depth_amd import synth  # noqa: E402
from helpers import assert_depth_parity, make_estimator, run_oracle  # noqa: E402
from test_randomized_gpu import _random_setup  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")


def mask_of(inl, n):
    m = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
    return torch.from_numpy(m.view(np.int32)).to(dev)


t0 = time.perf_counter()
bad, worst, worst_seed, frames = [], 0.0, -1, 0
for seed in range(first, first + count):
    P, cam, T, scanner, kw = _random_setup(seed)
    clouds = [synth.make_cloud(scanner, seed=200 + seed, frame=(seed + b) % 7) for b in range(B)]
    uvs = [synth.make_features(600 + 97 * b, seed=300 + seed + 1000 * b, width=cam.width, height=cam.height) for b in range(B)]
    planes = [synth.make_ground_plane(c) for c in clouds]
    est = make_estimator(P, camera=cam, T=T, max_frames=B, max_features=max(u.shape[0] for u in uvs))
    try:
        t_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
        t_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
        t_masks = [mask_of(p[1], c.shape[0]) for p, c in zip(planes, clouds)]
        t_depth = [torch.full((u.shape[0],), 7.0, dtype=torch.float64, device=dev) for u in uvs]
        t_type = [torch.full((u.shape[0],), -7, dtype=torch.int32, device=dev) for u in uvs]
        torch.cuda.synchronize()
        batch = est.prepareBatch(t_clouds, t_uvs, t_depth, t_type, np.stack([p[0] for p in planes]), t_masks, stride_bytes=16)
        est.runBatch(batch)  # (setInputClouds with the planes known at projection time + CalculateDepths)
        est.synchronize()
        for b in range(B):
            _, (d0, ty0) = run_oracle(P, clouds[b], uvs[b], planes[b], camera=cam, T=T)
            diff = assert_depth_parity(t_depth[b].cpu().numpy(), t_type[b].cpu().numpy(), d0, ty0, exact_main=not P.do_use_PCA)
            frames += 1
            if float(diff.max(initial=0.0)) > worst:
                worst, worst_seed = float(diff.max(initial=0.0)), seed
    except AssertionError as e:  # noqa: PERF203
        bad.append((seed, str(e)[:200]))
    finally:
        est.close()
print(f"batched random sweep (shipped library, {B} frames per launch set): seeds {first} .. {first + count - 1}: "
      f"{count - len(bad)} of {count} configurations ({frames} frames) equal to the oracle in {time.perf_counter() - t0:.0f} s; "
      f"max |depth - oracle| = {worst:.3e} m (seed {worst_seed})")
for s, why in bad[:20]:
    print("MISMATCH seed", s, why)
sys.exit(1 if bad else 0)

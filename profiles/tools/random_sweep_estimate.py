#!/usr/bin/env python3
"""Parity sweep of the production call - the plane ESTIMATED inside CalculateDepth (RANSAC with a seed, or the semantic
label image) - over random configurations (the generator of tests/test_randomized_gpu.py + random RANSAC thresholds,
iteration caps and pass-through limits): coefficients and inlier sets bit-equal to the oracle's estimate for the same request,
then the depths as in random_sweep.py (TEST TOOL: the oracle is the checker).
usage: random_sweep_estimate.py first_seed n_seeds"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from mono_lidar_depth_amd import ExceptionPclInvalid, RansacPlane, SemanticPlane, synth  # noqa: E402
from helpers import assert_depth_parity, make_estimator, make_oracle  # noqa: E402
from test_randomized_gpu import _random_setup  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
LABELS = (6, 7, 8, 9)
t0 = time.perf_counter()
bad, worst, worst_seed, n_r, n_s, n_fail = [], 0.0, -1, 0, 0, 0
for seed in range(first, first + count):
    P, cam, T, scanner, kw = _random_setup(seed)
    rng = np.random.default_rng(77000 + seed)
    P = P.replace(do_use_ransac_plane=1,
                  ransac_plane_distance_treshold=float(rng.choice([0.05, 0.1, 0.3])),
                  ransac_plane_refinement_treshold=float(rng.choice([0.05, 0.15, 0.4])),
                  ransac_plane_max_iterations=int(rng.choice([20, 200, 1000])),
                  ransac_plane_use_refinement=int(rng.random() < 0.8),
                  ransac_plane_min_z=float(rng.choice([-1001.0, -3.0])), ransac_plane_max_z=float(rng.choice([1000.0, -0.5])))
    cloud = synth.make_cloud(scanner, seed=200 + seed, frame=seed % 5)
    uv = synth.make_features(700, seed=300 + seed, width=cam.width, height=cam.height)
    semantic = bool(seed & 1)
    est = make_estimator(P, camera=cam, T=T)
    ref = make_oracle(P, camera=cam, T=T)
    ref.set_cloud(cloud)
    try:
        if semantic:
            # (rendered with the KITTI-like camera of the synthetic scenes at this camera's size: for a random camera the
            #  labels do not line up with the projection - any image is a valid request, both sides read the same one)
            img = synth.make_label_image(cloud, width=cam.width, height=cam.height)
            gp = SemanticPlane(img, LABELS, float(P.ransac_plane_refinement_treshold))
            n_s += 1
        else:
            gp = RansacPlane(seed=seed + 1)
            n_r += 1
        try:
            d, t = est.CalculateDepth(cloud, uv, gp)
            failed = False
        except ExceptionPclInvalid:
            failed = True
        try:
            c0, inl0 = (ref.estimate_semantic_plane(img, LABELS, float(P.ransac_plane_refinement_treshold)) if semantic
                        else ref.estimate_ground_plane(seed + 1))
            ref_failed = False
        except Exception:  # noqa: BLE001
            ref_failed = True
        assert failed == ref_failed, f"estimation outcome differs: hip failed={failed}, oracle failed={ref_failed}"
        if failed:
            n_fail += 1
            continue
        assert np.array_equal(gp.getModelCoeffs(), c0), "plane coefficients differ"
        assert np.array_equal(gp.getInlinersIndex(), inl0), "inlier sets differ"
        d0, ty0 = ref.calculate_depth(uv)
        diff = assert_depth_parity(d, t, d0, ty0, exact_main=not P.do_use_PCA)
        if float(diff.max(initial=0.0)) > worst:
            worst, worst_seed = float(diff.max(initial=0.0)), seed
    except AssertionError as e:  # noqa: PERF203
        bad.append((seed, str(e)[:200]))
    finally:
        est.close()
print(f"estimated-plane sweep: seeds {first} .. {first + count - 1}: {count - len(bad)} of {count} configurations equal to the oracle "
      f"({n_r} RANSAC, {n_s} semantic, {n_fail} where both report an invalid cloud) in {time.perf_counter() - t0:.0f} s; "
      f"max |depth - oracle| = {worst:.3e} m (seed {worst_seed})")
for s, why in bad[:20]:
    print("MISMATCH seed", s, why)
sys.exit(1 if bad else 0)

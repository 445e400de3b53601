#!/usr/bin/env python3
"""Soak of the one-frame entry points (helper thread, pinned staging block, lazy inlier lists): thousands of calls with
changing cloud sizes, feature counts and plane kinds on two contexts driven from two host threads; every K-th call is checked
against the oracle (TEST TOOL: the oracle is the checker).  usage: soak_frames.py [calls_per_thread] [check_every]"""
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from mono_lidar_depth_amd import GroundPlane, RansacPlane, SemanticPlane, NO_PLANE, capi, synth  # noqa: E402
from helpers import assert_depth_parity, make_estimator, make_oracle  # noqa: E402

CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 25
LABELS = (6, 7, 8, 9)
P = capi.params_c0()
clouds = [synth.make_cloud(sc, seed=40 + i, frame=i) for i, sc in enumerate((synth.HDL64, synth.HDL64_KITTI, synth.VLP16, synth.HDL64))]
planes = [synth.make_ground_plane(c) for c in clouds]
images = [synth.make_label_image(c) for c in clouds]
errors = []


def worker(tid):
    try:
        rng = np.random.default_rng(1000 + tid)
        est = make_estimator(P)
        checked = 0
        for k in range(CALLS):
            ci = int(rng.integers(len(clouds)))
            cloud = clouds[ci]
            n = int(rng.choice([cloud.shape[0], cloud.shape[0], int(rng.integers(3, cloud.shape[0])), 5000]))
            sub = np.ascontiguousarray(cloud[:n])
            F = int(rng.choice([0, 1, 63, 64, 65, 500, 2000, 3000]))
            uv = synth.make_features(max(F, 1), seed=int(rng.integers(1 << 30)))[:F]
            kind = int(rng.integers(5))
            check = (k % EVERY) == 0
            ref = None
            if check:
                ref = make_oracle(P)
                ref.set_cloud(sub)
            if kind == 0:      # supplied plane (indices beyond the truncated cloud dropped)
                co, inl = planes[ci]
                inl = inl[inl < n]
                gp = GroundPlane(co, inl)
                if check:
                    ref.set_ground_plane(co, inl)
            elif kind == 1:    # RANSAC inside the call
                seed = int(rng.integers(1 << 20))
                gp = RansacPlane(seed=seed)
                inl_ref = None
                if check:
                    try:
                        inl_ref = ref.estimate_ground_plane(seed)[1]
                    except RuntimeError:
                        ref = None
            elif kind == 2:    # semantic plane inside the call
                gp = SemanticPlane(images[ci], LABELS, P.ransac_plane_refinement_treshold)
                if check:
                    try:
                        ref.estimate_semantic_plane(images[ci], LABELS, P.ransac_plane_refinement_treshold)
                    except Exception:  # noqa: BLE001  (too few candidates in a truncated cloud)
                        ref = None
            elif kind == 3:    # no plane
                gp = NO_PLANE
                if check:
                    ref.set_ground_plane(None, None)
            else:              # null pointer: a RansacPlane is created and estimated (seed 0)
                gp = None
                if check:
                    try:
                        ref.estimate_ground_plane(0)
                    except RuntimeError:
                        ref = None
            try:
                d, t = est.CalculateDepth(sub, uv.reshape(-1, 2) if F else np.zeros((0, 2)), gp, uv_layout="Fx2")
            except Exception as e:  # noqa: BLE001
                if kind in (1, 2, 4) and "invalid" in str(e).lower():
                    continue  # GroundPlane::ExceptionPclInvalid on a truncated cloud without enough ground points
                raise
            if check and ref is not None and F:
                d0, t0 = ref.calculate_depth(uv)
                assert_depth_parity(d, t, d0, t0)
                if kind == 1 and inl_ref is not None:
                    assert np.array_equal(gp.getInlinersIndex(), inl_ref)  # (the lazily fetched list)
                checked += 1
        est.close()
        print(f"thread {tid}: {CALLS} calls, {checked} checked against the oracle", flush=True)
    except Exception as e:  # noqa: BLE001
        import traceback
        errors.append((tid, traceback.format_exc()))


t0 = time.time()
ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
for t in ths:
    t.start()
for t in ths:
    t.join()
print(f"soak: {2 * CALLS} calls in {time.time() - t0:.1f} s, errors: {len(errors)}")
for tid, tb in errors:
    print(f"--- thread {tid}\n{tb}")
sys.exit(1 if errors else 0)

"""Contexts are independent: two host threads, each with its own context on the same GPU, running different frames
concurrently (ctypes releases the GIL during the C calls) must both reproduce the oracle."""
import threading

import numpy as np
import pytest

from mono_lidar_depth_amd import GroundPlane, capi, synth

from helpers import assert_depth_parity, make_estimator, run_oracle

pytestmark = pytest.mark.gpu


def test_two_contexts_from_two_threads():
    P = capi.params_c0()
    jobs = []
    for t in range(2):
        frames = []
        for k in range(6):
            cloud = synth.make_cloud(synth.HDL64_KITTI, seed=600 + t, frame=k)
            uv = synth.make_features(800, seed=700 + 10 * t + k)
            frames.append((cloud, uv, synth.make_ground_plane(cloud)))
        jobs.append(frames)
    results = [[None] * 6 for _ in range(2)]
    errors = []

    def worker(t):
        try:
            est = make_estimator(P)
            for rep in range(5):  # repeat to overlap the two threads' calls
                for k, (cloud, uv, plane) in enumerate(jobs[t]):
                    results[t][k] = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(2):
        for k, (cloud, uv, plane) in enumerate(jobs[t]):
            _, (d0, t0) = run_oracle(P, cloud, uv, plane)
            assert_depth_parity(results[t][k][0], results[t][k][1], d0, t0)

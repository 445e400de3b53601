#!/usr/bin/env python3
"""Parity sweep beyond the committed seeds of tests/test_randomized_gpu.py: random parameter sets, cameras, mountings and
scanners (the test's own generator), HIP against the oracle (TEST TOOL: the oracle is the checker) - result types equal,
depths within 1e-4 m (bit-exact off the road / PCA paths), `_pointIndex` and the pixel map bit-exact.
usage: random_sweep.py first_seed n_seeds [route] [scanner]     route: default | fused | wave-only | dense (tests/conftest.py);
       scanner "dense128": every configuration on the 128 x 4096 cloud of BASELINE config 5 with 4000 features (lists of up to
       48 neighbours: the DENSE instantiations' register tiers)"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from mono_lidar_depth_amd import GroundPlane, synth  # noqa: E402
from helpers import assert_depth_parity, make_estimator, run_oracle  # noqa: E402
from test_randomized_gpu import _random_setup  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
route = sys.argv[3] if len(sys.argv) > 3 else "default"
if route != "default":  # the kernel routes of the parity suite: the test build of the same sources, switches by environment
    import os
    os.environ["MLD_FORCE_WAVE_PATH" if route == "wave-only" else "MLD_FORCE_THREAD_PATH"] = "1"
    if route == "dense":
        os.environ["MLD_K1MAX"] = "48"
    from mono_lidar_depth_amd import capi
    capi._lib = capi.load_ab()
t0 = time.perf_counter()
bad, types_seen, worst, worst_seed = [], {}, 0.0, -1
for seed in range(first, first + count):
    P, cam, T, scanner, kw = _random_setup(seed)
    nfeat = 900
    if len(sys.argv) > 4 and sys.argv[4] == "dense128":
        scanner, nfeat = synth.DENSE128, 4000
    cloud = synth.make_cloud(scanner, seed=200 + seed, frame=seed % 5)
    uv = synth.make_features(nfeat, seed=300 + seed, width=cam.width, height=cam.height)
    plane = synth.make_ground_plane(cloud)
    est = make_estimator(P, camera=cam, T=T)
    try:
        d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
        ref, (d0, t0_) = run_oracle(P, cloud, uv, plane, camera=cam, T=T)
        diff = assert_depth_parity(d, t, d0, t0_, exact_main=not P.do_use_PCA)
        if float(diff.max(initial=0.0)) > worst:
            worst, worst_seed = float(diff.max(initial=0.0)), seed
        assert np.array_equal(est.getPointIndex(), ref.point_index())
        assert np.array_equal(est.getPixelMap(), ref.pixel_map())
        for k, n in zip(*np.unique(t0_, return_counts=True)):
            types_seen[int(k)] = types_seen.get(int(k), 0) + int(n)
    except AssertionError as e:  # noqa: PERF203
        bad.append((seed, str(e)[:200]))
    finally:
        est.close()
tag = route + " route" + (", " + sys.argv[4] if len(sys.argv) > 4 else "")
print(f"random sweep ({tag}): seeds {first} .. {first + count - 1}: {count - len(bad)} of {count} configurations equal to the oracle "
      f"in {time.perf_counter() - t0:.0f} s; max |depth - oracle| = {worst:.3e} m (seed {worst_seed})")
print("result types met (type: features):", dict(sorted(types_seen.items())))
for s, why in bad[:20]:
    print("MISMATCH seed", s, why)
sys.exit(1 if bad else 0)

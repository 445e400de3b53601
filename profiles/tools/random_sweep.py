#!/usr/bin/env python3
"""Parity sweep beyond the slice tests/test_sweep_gpu.py runs: random parameter sets, cameras, mountings and scanners, HIP
against the oracle (TEST TOOL: the oracle is the checker; the per-configuration check is tests/sweeps.py:check_single) -
result types equal, depths within 1e-4 m (bit-exact off the road / PCA paths), `_pointIndex` and the pixel map bit-exact.
usage: random_sweep.py first_seed n_seeds [route] [scanner]     route: default | fused | wave-only | dense (tests/conftest.py);
       scanner "dense128": every configuration on the 128 x 4096 cloud of BASELINE config 5 with 4000 features (lists of up to
       48 neighbours: the DENSE instantiations' register tiers)"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import sweeps  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
route = sys.argv[3] if len(sys.argv) > 3 else "default"
dense128 = len(sys.argv) > 4 and sys.argv[4] == "dense128"
if route != "default":  # the kernel routes of the parity suite: the test build of the same sources, switches by environment
    os.environ["MLD_FORCE_WAVE_PATH" if route == "wave-only" else "MLD_FORCE_THREAD_PATH"] = "1"
    if route == "dense":
        os.environ["MLD_K1MAX"] = "48"
    from mono_lidar_depth_amd import capi
    capi._lib = capi.load_ab()
t0 = time.perf_counter()
bad, types_seen, worst, worst_seed = [], {}, 0.0, -1
for seed in range(first, first + count):
    try:
        dm, t_ref = sweeps.check_single(seed, dense128)
        if dm > worst:
            worst, worst_seed = dm, seed
        for k, n in zip(*np.unique(t_ref, return_counts=True)):
            types_seen[int(k)] = types_seen.get(int(k), 0) + int(n)
    except AssertionError as e:  # noqa: PERF203
        bad.append((seed, str(e)[:200]))
tag = route + " route" + (", dense128" if dense128 else "")
print(f"random sweep ({tag}): seeds {first} .. {first + count - 1}: {count - len(bad)} of {count} configurations equal to the oracle "
      f"in {time.perf_counter() - t0:.0f} s; max |depth - oracle| = {worst:.3e} m (seed {worst_seed})")
print("result types met (type: features):", dict(sorted(types_seen.items())))
for s, why in bad[:20]:
    print("MISMATCH seed", s, why)
sys.exit(1 if bad else 0)

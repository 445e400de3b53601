#!/usr/bin/env python3
"""Parity sweep of the production call - the plane ESTIMATED inside CalculateDepth (RANSAC with a seed, or the semantic
label image) - over random configurations (+ random RANSAC thresholds, iteration caps and pass-through limits): coefficients
and inlier sets bit-equal to the oracle's estimate for the same request, then the depths (TEST TOOL; the check is
tests/sweeps.py:check_estimate).
usage: random_sweep_estimate.py first_seed n_seeds"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import sweeps  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
t0 = time.perf_counter()
bad, worst, worst_seed, kinds = [], 0.0, -1, {"ransac": 0, "semantic": 0, "invalid": 0}
for seed in range(first, first + count):
    try:
        dm, kind = sweeps.check_estimate(seed)
        kinds[kind] += 1
        if dm > worst:
            worst, worst_seed = dm, seed
    except AssertionError as e:  # noqa: PERF203
        bad.append((seed, str(e)[:200]))
print(f"estimated-plane sweep: seeds {first} .. {first + count - 1}: {count - len(bad)} of {count} configurations equal to the oracle "
      f"({kinds['ransac']} RANSAC, {kinds['semantic']} semantic, {kinds['invalid']} where both report an invalid cloud) in "
      f"{time.perf_counter() - t0:.0f} s; max |depth - oracle| = {worst:.3e} m (seed {worst_seed})")
for s, why in bad[:20]:
    print("MISMATCH seed", s, why)
sys.exit(1 if bad else 0)

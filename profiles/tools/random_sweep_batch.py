#!/usr/bin/env python3
"""The parity sweep of random_sweep.py through the BATCHED entry points of the shipped library (frame slots: k_classify +
k_feature_fused + k_feature_wave, plane known at projection time): per random configuration a launch set of B ragged frames
against the oracle (TEST TOOL; the check is tests/sweeps.py:check_batch).
usage: random_sweep_batch.py first_seed n_seeds [frames_per_set=5]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import sweeps  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 5
t0 = time.perf_counter()
bad, worst, worst_seed, frames = [], 0.0, -1, 0
for seed in range(first, first + count):
    try:
        dm, n = sweeps.check_batch(seed, B)
        frames += n
        if dm > worst:
            worst, worst_seed = dm, seed
    except AssertionError as e:  # noqa: PERF203
        bad.append((seed, str(e)[:200]))
print(f"batched random sweep (shipped library, {B} frames per launch set): seeds {first} .. {first + count - 1}: "
      f"{count - len(bad)} of {count} configurations ({frames} frames) equal to the oracle in {time.perf_counter() - t0:.0f} s; "
      f"max |depth - oracle| = {worst:.3e} m (seed {worst_seed})")
for s, why in bad[:20]:
    print("MISMATCH seed", s, why)
sys.exit(1 if bad else 0)

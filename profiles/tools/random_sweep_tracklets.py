#!/usr/bin/env python3
"""Parity sweep of the batched tracklet layer (mld_tracklets_depths_device: two banks of slots, the previous frame's slot
resident, feature groups when the sequences are few) over random configurations: S sequences x 3 frames per configuration
against the CPU restatement of TrackletDepthModule::process (TEST TOOL; the check is tests/sweeps.py:check_tracklets).
usage: random_sweep_tracklets.py first_seed n_seeds [sequences=3] [tracks=2600]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import sweeps  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 3
NT = int(sys.argv[4]) if len(sys.argv) > 4 else 2600
t0 = time.perf_counter()
bad, worst, worst_seed, checks = [], 0.0, -1, 0
for seed in range(first, first + count):
    try:
        dm, n = sweeps.check_tracklets(seed, S, NT)
        checks += n
        if dm > worst:
            worst, worst_seed = dm, seed
    except AssertionError as e:  # noqa: PERF203
        bad.append((seed, str(e)[:200]))
print(f"tracklet sweep ({S} sequences x 3 frames, {NT} tracks, list capacities default / 48-24 by seed parity): seeds {first} .. "
      f"{first + count - 1}: {count - len(bad)} of {count} configurations ({checks} sequence-frames) equal to the oracle in "
      f"{time.perf_counter() - t0:.0f} s; max |d - oracle| = {worst:.3e} m (seed {worst_seed}; float32 outputs)")
for s, why in bad[:20]:
    print("MISMATCH seed", s, why)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Features per path (lane-per-feature kernel / handed to the wave kernel) of the config-2 batch at several list budgets
(mld_get_path_counts).  usage: path_counts.py [frames=64]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import resident  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for cap, budget in (((32, 24), 56), ((32, 24), 40), ((32, 24), 36), ((32, 24), 32)):
    w = resident.build(B=B)
    est, b = w["est"], w["batch"]
    est.setListCapacity(*cap)
    est.setListBudget(budget)
    est.runBatch(b)
    est.synchronize()
    counts = np.array([est.pathCounts(s) for s in range(B)])
    est.timingEnable(True)
    est.timingReset()
    for _ in range(4):
        est.runBatch(b)
    est.synchronize()
    t = {n: round(est.kernelTimeMs(k)[0] * 1e3, 1) for k, n in ((1, "fused"), (3, "wave"))}
    print(f"capacities {cap} budget {budget}: lane path {counts[:, 0].sum()} features, handed over {counts[:, 1].sum()} "
          f"({100.0 * counts[:, 1].sum() / max(1, counts[:, 0].sum()):.2f} %), max per slot {counts[:, 1].max()}; kernels us {t}")
    est.close()

#!/usr/bin/env python3
"""Phase shares of the feature kernels from the s_memtime stamps of the diagnostic build
(profiles/tools/mkvariant.sh stamps -DMLD_STAMPS; run with MLD_HIP_LIBRARY pointing at it)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import resident  # noqa: E402

NAMES = {0: "prologue", 1: "queue+uv", 2: "bitmap loads", 3: "bit walk / counts", 4: "cell+key loads", 5: "narrow list",
         6: "dead stores+enqueue", 7: "info+uv reload", 8: "pass1: point loads", 9: "hist rest", 10: "triangle",
         11: "tail", 12: "stores+enqueue", 13: "road: point loop", 14: "road: eigen + tail"}


def main():
    w = resident.build(B=int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
    est, b = w["est"], w["batch"]
    lib = est._lib
    out = np.zeros((2, 32768, 16), dtype=np.uint32)
    for _ in range(3):
        est.runBatch(b)
    est.synchronize()
    lib.mld_debug_read_stamps(C.c_void_p(out.ctypes.data))
    steps = 4
    est.timingEnable(True)
    est.timingReset()
    for _ in range(steps):
        est.runBatch(b)
    est.synchronize()
    lib.mld_debug_read_stamps(C.c_void_p(out.ctypes.data))
    for k, name in ((0, "k_feature_fused"),):
        v = out[k].astype(np.float64)
        waves = v[:, 15].sum()
        ph = v[:, :15].sum(0)
        tot = ph.sum()
        print(f"== {name}: {waves / steps:.0f} stamped waves/launch, {tot / max(1, waves):.0f} cycles/wave "
              f"(kernel {est.kernelTimeMs(1)[0] * 1e3:.1f} us with stamps)")
        for i in range(15):
            if ph[i]:
                print(f"   {i:2d} {NAMES.get(i, '?'):22s} {100.0 * ph[i] / tot:5.1f} %   {ph[i] / waves:8.0f} cyc/wave")


if __name__ == "__main__":
    main()

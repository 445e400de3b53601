#!/usr/bin/env python3
"""Phase shares of k_feature_fused on the DENSE workload of BASELINE config 5 (128x4096 cloud, 10 000 integer-pixel
features per frame, list capacities 48 / 24) from the s_memtime stamps of the diagnostic build
(profiles/tools/mkvariant.sh stamps -DMLD_STAMPS; run with MLD_HIP_LIBRARY pointing at it), plus the window statistics
of one frame (neighbours in the narrow / road window, from the pixel map)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import resident  # noqa: E402
from mono_lidar_depth_amd import synth  # noqa: E402
from stamps import NAMES  # noqa: E402


def uv_c5(F, seed):
    rng = np.random.default_rng(5000 + seed)
    return np.stack([rng.integers(0, synth.KITTI_W, F), rng.integers(100, synth.KITTI_H, F)], axis=1).astype(np.float64)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    w = resident.build(B=B, F=10000, U=4, scanner=synth.DENSE128, seed=5, uv_fn=uv_c5, list_capacity=(48, 24))
    est, b = w["est"], w["batch"]
    lib = est._lib
    has_stamps = hasattr(lib, "mld_debug_read_stamps")
    out = np.zeros((2, 32768, 16), dtype=np.uint32)
    for _ in range(3):
        est.runBatch(b)
    est.synchronize()
    if has_stamps:
        lib.mld_debug_read_stamps(C.c_void_p(out.ctypes.data))
    steps = 4
    est.timingEnable(True)
    est.timingReset()
    for _ in range(steps):
        est.runBatch(b)
    est.synchronize()
    for k, name in ((0, "k_project_scatter"), (5, "k_classify"), (1, "k_feature_fused"), (3, "k_feature_wave")):
        ms, n = est.kernelTimeMs(k)
        print(f"{name}: {ms * 1e3:.1f} us per launch of {B} frames ({n} launches)")
    if has_stamps:
        lib.mld_debug_read_stamps(C.c_void_p(out.ctypes.data))
        v = out[0].astype(np.float64)
        waves = v[:, 15].sum()
        ph = v[:, :15].sum(0)
        tot = ph.sum()
        print(f"== k_feature_fused: {waves / steps:.0f} stamped waves/launch, {tot / max(1, waves):.0f} cycles/wave")
        for i in range(15):
            if ph[i]:
                print(f"   {i:2d} {NAMES.get(i, '?'):22s} {100.0 * ph[i] / tot:5.1f} %   {ph[i] / waves:8.0f} cyc/wave")
    # window statistics of frame 0 from its pixel map
    pm = est.getPixelMap(0).reshape(synth.KITTI_H, synth.KITTI_W) >= 0
    ii = np.zeros((synth.KITTI_H + 1, synth.KITTI_W + 1), dtype=np.int64)
    ii[1:, 1:] = pm.cumsum(0).cumsum(1)
    uv = w["uvs_h"][0]
    P = w["P"]

    def counts(sx, sy):
        hx, hy = 0.5 * P.pixelarea_search_witdh * sx, 0.5 * P.pixelarea_search_height * sy
        x0 = np.maximum(uv[:, 0] - hx, 0).astype(int)
        x1 = np.minimum(uv[:, 0] + hx, synth.KITTI_W - 1).astype(int)
        y0 = np.maximum(uv[:, 1] - hy, 0).astype(int)
        y1 = np.minimum(uv[:, 1] + hy, synth.KITTI_H - 1).astype(int)
        return ii[y1 + 1, x1 + 1] - ii[y0, x1 + 1] - ii[y1 + 1, x0] + ii[y0, x0]

    k1, k2 = counts(1.0, 1.0), counts(2.0, 1.5)
    t = w["type"][0].cpu().numpy()
    print(f"frame 0: occupied cells {pm.mean():.3f}; narrow k1 mean {k1.mean():.2f} p50 {np.percentile(k1, 50):.0f} "
          f"p90 {np.percentile(k1, 90):.0f} max {k1.max()}; road k2 mean {k2.mean():.2f} p90 {np.percentile(k2, 90):.0f} "
          f"max {k2.max()}; k1 > 8: {np.mean(k1 > 8):.3f}; k1 > 12: {np.mean(k1 > 12):.3f}; k1 > 16: {np.mean(k1 > 16):.3f}; "
          f"k1 > 24: {np.mean(k1 > 24):.3f}; k2 > 32: {np.mean(k2 > 32):.3f}; k2 > 48: {np.mean(k2 > 48):.3f}")
    print("result types frame 0:", {int(k): int(c) for k, c in zip(*np.unique(t, return_counts=True))})


if __name__ == "__main__":
    main()

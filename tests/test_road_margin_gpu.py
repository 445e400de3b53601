"""What "result types identical" rests on for the ROAD path.

The lane-per-feature kernel computes the M-estimator plane with reordered one-pass sums, reciprocal / rsqrt estimates
refined by Newton steps and a direct eigenvector solver (mld_kernels.hip, "reduced-cost arithmetic for the ROAD path"),
so its road depths differ from the oracle's by rounding (bar: 1e-4 m; the weights 1 / |distance to the prior plane| make
the sums ill-conditioned when an inlier lies almost on the prior, and the two summation orders then differ visibly).
The result TYPE of such a feature is decided by comparing that depth with the global thresholds
(TresholdDepthGlobal.cpp:16-36) and the local bounds (TresholdDepthLocal.cpp:18-66): types agree for certain only if
no depth sits within the arithmetic difference of a bound.  This test measures that on >= 100 seeded frames: for every
feature whose road estimator produced a plane it re-derives the raw (pre-threshold) depth and the four bounds from the
oracle's trace, checks the re-derivation against the oracle's own decision, and asserts (a) HIP types == oracle types,
(b) per feature, the distance of the raw depth to the nearest bound exceeds its |HIP depth - oracle depth| (features
without a HIP depth: the largest difference seen) by a factor of ten or more, (c) every difference is inside the bar."""
import numpy as np
import pytest

from mono_lidar_depth_amd import capi, synth
from oracle import oracle

from helpers import make_estimator, make_oracle

pytestmark = pytest.mark.gpu

N_FRAMES, BATCH, F = 104, 8, 700


def _mask_of(inl, n, dev):
    import torch
    m = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
    return torch.from_numpy(m.view(np.int32)).to(dev)


def test_road_path_types_do_not_hang_on_rounding():
    import torch
    P = capi.params_c0()
    assert P.treshold_depth_enabled and P.treshold_depth_local_enabled and P.treshold_depth_local_valuetype == 1
    dev = torch.device("cuda:0")
    est = make_estimator(P, max_frames=BATCH, max_features=F)  # batches: the lane-per-feature kernel, as in the bench
    diffs, margins_ok, margins_fail, n_road = [], [], [], 0
    for b0 in range(0, N_FRAMES, BATCH):
        clouds = [synth.make_cloud(synth.HDL64_KITTI, seed=500 + b, frame=b % 7) for b in range(b0, b0 + BATCH)]
        planes = [synth.make_ground_plane(c) for c in clouds]
        uvs = [synth.make_features(F, seed=900 + b) for b in range(b0, b0 + BATCH)]
        d = [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(BATCH)]
        t = [torch.empty(F, dtype=torch.int32, device=dev) for _ in range(BATCH)]
        batch = est.prepareBatch([torch.from_numpy(c).to(dev) for c in clouds], [torch.from_numpy(u).to(dev) for u in uvs],
                                 d, t, np.stack([p[0] for p in planes]),
                                 [_mask_of(p[1], c.shape[0], dev) for p, c in zip(planes, clouds)])
        torch.cuda.synchronize()
        est.runBatch(batch)
        est.synchronize()
        for i in range(BATCH):
            ref = make_oracle(P)
            ref.set_cloud(clouds[i])
            ref.set_ground_plane(*planes[i])
            d0, t0 = ref.calculate_depth(uvs[i], 8)
            dg, tg = d[i].cpu().numpy(), t[i].cpu().numpy()
            assert np.array_equal(tg, t0), (b0 + i, np.nonzero(tg != t0)[0][:5])
            cam = ref.cloud_camera_cs()            # 3 x N or N x 3
            cam = cam if cam.shape[-1] == 3 else cam.T
            pidx = ref.point_index()
            co = planes[i][0].astype(np.float64)
            prior_n = co[:3] / np.sqrt(co[0] * co[0] + (co[1] * co[1] + co[2] * co[2]))  # DepthEstimator.cpp:286-292
            # features that entered the road fallback and came out with a depth-dependent type
            for f in np.nonzero(np.isin(t0, (16, 4, 5, 6, 7)))[0]:
                tr = ref.trace_feature(float(uvs[i][f, 0]), float(uvs[i][f, 1]))
                if not tr["reached_road"] or len(tr["road_pos"]) < 3:
                    continue  # the type came from the main path (bit-exact there)
                pts = cam[pidx[tr["road_idx"][tr["road_pos"]]]]
                n, off = oracle.mestimator_plane(pts, prior_n, float(co[3]))
                ray = ref.viewing_ray(float(uvs[i][f, 0]), float(uvs[i][f, 1]))
                raw = -off * ray[2] / float(n @ ray)
                zmin, zmax = pts[:, 2].min(), pts[:, 2].max()
                r = (zmax - zmin) * P.treshold_depth_local_value
                bounds = np.array([P.treshold_depth_min, P.treshold_depth_max, zmin - r, zmax + r], dtype=np.float64)
                # the recomputation must reproduce the oracle's decision, or the margin below means nothing
                if raw < bounds[0]:
                    exp = 5
                elif raw > bounds[1]:
                    exp = 4
                elif raw < bounds[2]:
                    exp = 7
                elif raw > bounds[3]:
                    exp = 6
                else:
                    exp = 16
                assert exp == t0[f], (b0 + i, f, raw, bounds, t0[f])
                margin = float(np.abs(raw - bounds).min())
                if exp == 16:
                    assert abs(raw - d0[f]) <= 1e-9 * max(1.0, abs(raw)), (raw, d0[f])
                    diffs.append(abs(dg[f] - d0[f]))
                    margins_ok.append(margin)
                else:
                    margins_fail.append(margin)
            n_road += int((t0 == 16).sum())
    diffs, margins_ok, margins_fail = np.array(diffs), np.array(margins_ok), np.array(margins_fail)
    assert diffs.size > 10000 and n_road >= diffs.size
    worst = float(diffs.max())
    assert worst <= 1e-4, worst                                    # the parity bar
    assert (margins_ok > 10.0 * diffs).all(), float((margins_ok / np.maximum(diffs, 1e-300)).min())
    assert margins_fail.size == 0 or float(margins_fail.min()) > 10.0 * worst, (float(margins_fail.min()), worst)
    print(f"road path: {diffs.size} SuccessRoad + {margins_fail.size} threshold results re-derived; |HIP - oracle| max "
          f"{worst:.3e} m, 99.9 % {np.quantile(diffs, 0.999):.3e} m, median {np.median(diffs):.3e} m; smallest distance "
          f"of a raw depth to a bound {min(margins_ok.min(), margins_fail.min() if margins_fail.size else np.inf):.3e} m; "
          f"smallest margin / difference ratio {float((margins_ok / np.maximum(diffs, 1e-300)).min()):.3e}")

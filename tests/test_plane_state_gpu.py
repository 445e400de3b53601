"""The ground-plane state in the pixel-map keys (k_project_scatter) against the reference's float distance test
(DepthEstimator.cpp:810-815), where it is hardest: points whose plane distance lies ON or within float rounding of
`ransac_plane_point_distance_treshold`.  The projection must mark such points "unsure" (never guess), and both feature
kernels must then take the exact general loop and agree with the oracle.  The thresholds are built from the clouds
themselves: the un-normalised float distance of a chosen visible road point, computed as the reference does (camera ->
lidar in f64, cast to float, float plane arithmetic), its float neighbours, and values a few ulp away."""
import numpy as np
import pytest

from mono_lidar_depth_amd import GroundPlane, capi, synth

from helpers import assert_depth_parity, make_estimator, make_oracle, run_oracle

pytestmark = pytest.mark.gpu


def _reference_distances(ref, cloud, coeffs):
    """|a x + b y + c z + d| in float for every point, on the coordinates the reference uses (:810-812)."""
    cam = ref.cloud_camera_cs()
    cam = cam if cam.shape[-1] == 3 else cam.T                       # N x 3, f64
    Tinv, _ = ref.calibration()
    lid = np.empty_like(cam)
    for r in range(3):  # fixed-size product order: t + (a x + (b y + c z))
        lid[:, r] = Tinv[r, 3] + (Tinv[r, 0] * cam[:, 0] + (Tinv[r, 1] * cam[:, 1] + Tinv[r, 2] * cam[:, 2]))
    f = lid.astype(np.float32)
    co = np.asarray(coeffs, dtype=np.float32)
    d = np.abs(((co[0] * f[:, 0] + co[1] * f[:, 1]) + co[2] * f[:, 2]) + co[3])
    return d.astype(np.float32)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_thresholds_on_and_next_to_point_distances(seed):
    import torch
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=700 + seed, frame=seed)
    coeffs, inl = synth.make_ground_plane(cloud)
    uv = synth.make_features(1500, seed=710 + seed)
    P0 = capi.params_c0()
    ref0 = make_oracle(P0)
    ref0.set_cloud(cloud)
    ref0.set_ground_plane(coeffs, inl)
    d = _reference_distances(ref0, cloud, coeffs)
    vis = ref0.point_index()                                          # visible points: the ones that get a key
    rng = np.random.default_rng(seed)
    # distances of visible points that are not inliers of the plane but close to it: flipping their far / near decision
    # changes which features take the road fallback
    cand = vis[(d[vis] > 0.05) & (d[vis] < 2.0)]
    assert cand.size > 100
    picks = rng.choice(cand, 6, replace=False)
    dev = torch.device("cuda:0")
    n_checked = 0
    for pidx in picks:
        t0 = np.float32(d[pidx])
        for thr in (float(t0), float(np.nextafter(t0, np.float32(0))), float(np.nextafter(t0, np.float32(9))),
                    float(t0) * (1 + 3e-8), float(t0) * (1 - 3e-8), float(t0) + 1e-9):
            P = P0.replace(ransac_plane_point_distance_treshold=thr)
            _, (d_ref, t_ref) = run_oracle(P, cloud, uv, (coeffs, inl), n_threads=8)
            # (a) one frame per call, plane handed in with the cloud: wave-cooperative kernel, state in the keys
            est = make_estimator(P)
            dg, tg = est.CalculateDepth(cloud, uv, GroundPlane(coeffs, inl))
            assert_depth_parity(dg, tg, d_ref, t_ref)
            est.close()
            # (b) a batch of two slots: lane-per-feature kernel
            est = make_estimator(P, max_frames=2, max_features=uv.shape[0])
            m = np.zeros((cloud.shape[0] + 31) // 32, dtype=np.uint32)
            np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
            tm = torch.from_numpy(m.view(np.int32)).to(dev)
            tc, tu = torch.from_numpy(cloud).to(dev), torch.from_numpy(uv).to(dev)
            out_d = [torch.empty(uv.shape[0], dtype=torch.float64, device=dev) for _ in range(2)]
            out_t = [torch.empty(uv.shape[0], dtype=torch.int32, device=dev) for _ in range(2)]
            b = est.prepareBatch([tc, tc], [tu, tu], out_d, out_t, np.stack([coeffs, coeffs]), [tm, tm])
            torch.cuda.synchronize()
            est.runBatch(b)
            est.synchronize()
            for q in range(2):
                assert_depth_parity(out_d[q].cpu().numpy(), out_t[q].cpu().numpy(), d_ref, t_ref)
            est.close()
            n_checked += 1
    assert n_checked == 36

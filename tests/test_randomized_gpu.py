"""Randomised configurations: random parameter sets, cameras and lidar->camera transforms (seeded), HIP against the
oracle.  Catches assumptions that the fixed KITTI-like set-up of the other tests would hide (image size, rotated
mounting, window sizes, every mode combination)."""
import numpy as np
import pytest

from mono_lidar_depth_amd import CameraPinhole, GroundPlane, capi, synth

from helpers import assert_depth_parity, make_estimator, run_oracle

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("feature_kernel_path")]


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _random_setup(seed):
    rng = np.random.default_rng(9100 + seed)
    pick = lambda *a: a[int(rng.integers(len(a)))]  # noqa: E731
    road_tri = rng.random() < 0.3
    kw = dict(
        pixelarea_search_witdh=int(rng.integers(2, 15)), pixelarea_search_height=int(rng.integers(2, 15)),
        radiusSearch_count_min=int(rng.integers(1, 5)),
        do_use_histogram_segmentation=int(rng.random() < 0.7),
        histogram_segmentation_bin_witdh=pick(0.1, 0.3, 1.0), histogram_segmentation_min_pointcount=int(rng.integers(1, 5)),
        treshold_depth_enabled=int(rng.random() < 0.8), treshold_depth_mode=int(rng.integers(0, 2)),
        treshold_depth_min=int(rng.integers(0, 6)), treshold_depth_max=int(rng.integers(30, 101)),
        treshold_depth_local_enabled=int(rng.random() < 0.8), treshold_depth_local_mode=int(rng.integers(0, 2)),
        treshold_depth_local_valuetype=int(rng.integers(0, 2)), treshold_depth_local_value=pick(0.1, 0.5, 1.0),
        do_use_PCA=int(rng.random() < 0.15),
        do_use_triangle_size_maximation=int(rng.random() < 0.8),
        do_check_triangleplanar_condition=int(rng.random() < 0.8), triangleplanar_crossnorm_treshold=pick(0.05, 0.1, 0.3),
        viewray_plane_orthoganality_treshold=pick(0.0, 0.03, 0.2),
        do_use_cut_behind_camera=int(rng.random() < 0.8),
        do_use_ransac_plane=int(rng.random() < 0.85),
        plane_estimator_use_triangle_maximation=int(road_tri), plane_estimator_use_mestimator=int(not road_tri),
        plane_estimator_z_x_min_relation=pick(0.0, 0.2), ransac_plane_point_distance_treshold=pick(0.1, 0.2, 0.5),
    )
    P = capi.params_c0().replace(**kw)
    W, H = pick((640, 480), (1242, 375), (1920, 1080), (800, 300))
    f = float(rng.uniform(0.45, 1.1) * W)
    cam = CameraPinhole(W, H, f, W / 2 + float(rng.uniform(-20, 20)), H / 2 + float(rng.uniform(-20, 20)))
    base = synth.T_CAM_LIDAR[:, :3]
    R = _rot(*np.deg2rad(rng.uniform(-6, 6, 3))) @ base
    t = synth.T_CAM_LIDAR[:, 3] + rng.uniform(-0.3, 0.3, 3)
    T = np.concatenate([R, t[:, None]], axis=1)
    scanner = pick(synth.HDL64_KITTI, synth.VLP16, synth.HDL64)
    return P, cam, T, scanner, kw


# (1990: found by profiles/tools/random_sweep.py - depth thresholds off, a road estimate 3.8 km behind the camera: its depth
#  answers to 1e-4 m only if the M-estimator's weights - reciprocals of a cancellation - are evaluated in the reference's
#  own operation order; LAB.md 5.32)
# (4266, 6081: search windows three pixels wide - the road points are returns of one azimuth, collinear to the coordinates'
#  rounding; the normal is then in the SVD of the point matrix and no longer in its scatter: road_qr, LAB.md 5.33)
# (112686, 110803: three returns of one azimuth 75 m / 20 m away - the lane-per-feature kernel's road sums, taken about the
#  camera's origin, left 1.1e-4 m / 3.5e-5 m in the depth with its own error estimate at 1.5e-5 / 8e-6: the sums are taken
#  about the list's first point since, LAB.md 6.24)
@pytest.mark.parametrize("seed", [*range(24), 1990, 4266, 6081, 112686, 110803])
def test_random_configuration(seed):
    P, cam, T, scanner, kw = _random_setup(seed)
    cloud = synth.make_cloud(scanner, seed=200 + seed, frame=seed % 5)
    uv = synth.make_features(900, seed=300 + seed, width=cam.width, height=cam.height)
    plane = synth.make_ground_plane(cloud)
    est = make_estimator(P, camera=cam, T=T)
    d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    _, (d0, t0) = run_oracle(P, cloud, uv, plane, camera=cam, T=T)
    assert_depth_parity(d, t, d0, t0, exact_main=not P.do_use_PCA)
    # visible bookkeeping is integer work: bit-exact for any camera / transform
    ref, _ = run_oracle(P, cloud, uv[:1], plane, camera=cam, T=T)
    assert np.array_equal(est.getPointIndex(), ref.point_index()), kw
    assert np.array_equal(est.getPixelMap(), ref.pixel_map()), kw


def test_cloud_with_non_finite_and_duplicate_points():
    """NaN / inf coordinates, exact duplicates (pixel collisions: the first index must win), points on the camera plane
    and behind it, inside an otherwise normal cloud."""
    P = capi.params_c0()
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=71, frame=1).copy()
    rng = np.random.default_rng(71)
    n = cloud.shape[0]
    bad = rng.choice(n, 600, replace=False)
    cloud[bad[:150], 0] = np.nan
    cloud[bad[150:300], 1] = np.inf
    cloud[bad[300:450], 2] = -np.inf
    cloud[bad[450:500]] = 0.0                       # at the lidar origin
    cloud[bad[500:550], 0] = 0.27                   # z_cam == 0 for the synthetic mounting
    # finite but enormous coordinates, straight ahead: the single-precision pre-cull overflows (inf / NaN intermediates)
    # and must hand these points to the exact path, where they project into the image like any other point
    cloud[bad[550:560], :3] = np.array([3e37, 1e35, -2e35], dtype=np.float32) * rng.uniform(0.5, 1.0, (10, 1)).astype(np.float32)
    cloud[bad[560:570], :3] = np.array([2e38, -3e36, 1e36], dtype=np.float32) * rng.uniform(0.9, 1.0, (10, 1)).astype(np.float32)
    dup_src = rng.choice(n, 3000, replace=False)
    dup_dst = rng.choice(n, 3000, replace=False)
    cloud[dup_dst] = cloud[dup_src]
    uv = synth.make_features(1500, seed=71)
    plane = synth.make_ground_plane(cloud)
    est = make_estimator(P)
    d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    ref, (d0, t0) = run_oracle(P, cloud, uv, plane)
    assert_depth_parity(d, t, d0, t0)
    assert np.array_equal(est.getPointIndex(), ref.point_index())
    assert np.array_equal(est.getPixelMap(), ref.pixel_map())


def test_cloud_size_limit():
    """23-bit point index in the pixel-map key (beside the tag and the inlier flag): 8 388 607 points are accepted,
    one more is a capacity error."""
    import torch
    from mono_lidar_depth_amd import DepthEstimatorError
    P = capi.params_c0().replace(do_use_ransac_plane=0)
    est = make_estimator(P)
    n_max = (1 << 23) - 1
    big = torch.zeros((n_max + 1, 4), dtype=torch.float32, device="cuda:0")
    # a handful of real points at the very end of the largest legal cloud: the highest indices must round-trip
    tail = torch.from_numpy(synth.make_cloud(synth.VLP16, seed=5)[:20000]).to("cuda:0")
    big[n_max - tail.shape[0]:n_max] = tail
    torch.cuda.synchronize()
    with pytest.raises(DepthEstimatorError) as ei:
        est.setInputCloud(big, None, plane_given=False)
    assert ei.value.code == capi.MLD_ERR_CAPACITY
    est.setInputCloud(big[:n_max], None, plane_given=False)
    pidx = est.getPointIndex()
    small = make_estimator(P)
    small.setInputCloud(tail, None, plane_given=False)
    # the zero points in front project nowhere (z_cam <= 0 after the mounting offset), so the visible set is the tail's
    assert np.array_equal(pidx - (n_max - tail.shape[0]), small.getPointIndex())
    uv = synth.make_features(500, seed=5)
    uvd = torch.from_numpy(uv).to("cuda:0")
    d_big, t_big = est.CalculateDepth(uvd)
    d_small, t_small = small.CalculateDepth(uvd)
    assert torch.equal(t_big, t_small) and torch.equal(d_big, d_small)


@pytest.mark.parametrize("seed", range(4))
def test_projection_over_extreme_magnitudes(seed):
    """The single-precision pre-cull of k_project_scatter against the exact projection on coordinates spread over forty
    decades (under- and overflowing its f32 intermediates), with a randomly rotated, slightly scaled lidar->camera
    transform: the set of visible points and the pixel map are the oracle's, bit for bit."""
    rng = np.random.default_rng(4200 + seed)
    P = capi.params_c0().replace(do_use_ransac_plane=0)
    n = 60000
    mag = 10.0 ** rng.uniform(-20, 20, (n, 3))
    sign = rng.choice([-1.0, 1.0], (n, 3))
    pts = (mag * sign).astype(np.float32)
    # half of the points: one dominant forward component, so that many land inside the image at every magnitude
    fwd = rng.random(n) < 0.5
    scale = 10.0 ** rng.uniform(-15, 30, n)
    pts[fwd, 0] = (scale[fwd] * rng.uniform(1.0, 3.0, fwd.sum())).astype(np.float32)
    pts[fwd, 1] = (scale[fwd] * rng.uniform(-1.0, 1.0, fwd.sum())).astype(np.float32)
    pts[fwd, 2] = (scale[fwd] * rng.uniform(-0.4, 0.4, fwd.sum())).astype(np.float32)
    cloud = np.zeros((n, 4), dtype=np.float32)
    cloud[:, :3] = pts
    base = synth.T_CAM_LIDAR[:, :3]
    R = _rot(*np.deg2rad(rng.uniform(-10, 10, 3))) @ base * rng.uniform(0.5, 2.0)
    T = np.concatenate([R, rng.uniform(-1, 1, 3)[:, None]], axis=1)
    cam = CameraPinhole(1242, 375, 721.5377, 609.5593, 172.854)
    uv = synth.make_features(50, seed=seed)
    est = make_estimator(P, camera=cam, T=T)
    d, t = est.CalculateDepth(cloud, uv, None)
    ref, (d0, t0) = run_oracle(P, cloud, uv, None, camera=cam, T=T)
    assert np.array_equal(est.getPointIndex(), ref.point_index())
    assert np.array_equal(est.getPixelMap(), ref.pixel_map())
    assert ref.point_index().size > 1000  # the case does exercise visible points
    assert np.array_equal(t, t0)


def test_known_exceedance_of_the_absolute_road_tolerance_on_an_undetermined_fit():
    """The ONE road depth of round 6's 81 850 sweep configurations that left the 1e-4 m tolerance after LAB.md 6.24 (batched
    sweep, seed 160643, frame 0, feature 354; found in the closing hour, documented instead of tuned away): window 2 px wide,
    thresholds off; the three inliers are returns of one azimuth on consecutive rings 68 m away, collinear to 2e-6 - singular
    values 3.8 : 7.6e-6 : 0 -, the ray grazes the fitted plane (|n.ray| = 3.6e-6) and the estimate lies 2.3 km BEHIND the camera.
    The normal of such a set is determined to eps * s3 / s2 = 5e-11 rad by ANY backward-stable f64 decomposition, i.e. the
    depth to 2298 m * 5e-11 / 3.6e-6 = 3 cm; the wave kernel (QR + one-sided Jacobi, sums as DPP trees) and the oracle (the
    reference's order, sums in sequence) agree to 1.35e-4 m = 6e-8 of the depth.  Pinned here: result types identical,
    every depth within 1e-4 m + 1e-7 |depth|, and this feature is the only one beyond 1e-4 m."""
    seed = 160643
    P, cam, T, scanner, kw = _random_setup(seed)
    cloud = synth.make_cloud(scanner, seed=200 + seed, frame=seed % 7)
    uv = synth.make_features(600, seed=300 + seed, width=cam.width, height=cam.height)
    plane = synth.make_ground_plane(cloud)
    est = make_estimator(P, camera=cam, T=T)
    d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    _, (d0, t0) = run_oracle(P, cloud, uv, plane, camera=cam, T=T)
    assert np.array_equal(t, t0)
    assert np.array_equal(np.isnan(d), np.isnan(d0))
    diff = np.abs(np.nan_to_num(d) - np.nan_to_num(d0))
    assert np.all(diff <= 1e-4 + 1e-7 * np.abs(np.nan_to_num(d0))), float(diff.max())
    beyond = np.nonzero(diff > 1e-4)[0]
    assert len(beyond) <= 1 and all(abs(d0[i]) > 1000.0 and t0[i] == 16 for i in beyond), (beyond, d0[beyond])
    main = t0 != 16
    assert np.array_equal(d[main], d0[main], equal_nan=True)

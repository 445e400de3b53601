"""Ground-plane estimation on the GPU (RansacPlane::CalculateInliersPlane).

Parity against PCL is unpinned (time-seeded sampling upstream); what is checked: (a) the reference's own test
RansacPlane.CalculateInlersPlane (test_monolidar_fusion.cpp:376-441: coefficients within 0.2 of the truth) holds for
the CPU restatement and for the HIP implementation, (b) HIP == CPU restatement bit for bit (same hash-based draws),
(c) depths computed with the estimated plane agree with the oracle end to end.
"""
import numpy as np
import pytest

from mono_lidar_depth_amd import DepthEstimatorError, ExceptionPclInvalid, RansacPlane, capi, synth

from helpers import assert_depth_parity, make_estimator, make_oracle


def _reference_test_cloud():
    rng = np.random.default_rng(1234)
    n = 18000
    cl = np.zeros((n, 4), np.float32)
    cl[:, 0] = rng.uniform(-20, 20, n) + rng.normal(0, 0.5, n)
    cl[:, 1] = rng.uniform(-20, 20, n) + rng.normal(0, 0.5, n)
    cl[:, 2] = -1.6 + rng.normal(0, 0.5, n)
    cl[:, 3] = 150
    return cl


REF_TEST_PARAMS = dict(ransac_plane_distance_treshold=0.2, ransac_plane_max_iterations=600,
                       ransac_plane_use_refinement=1, ransac_plane_refinement_treshold=0.05,
                       ransac_plane_probability=0.99, ransac_plane_min_z=-1000.0, ransac_plane_max_z=1000.0)


def _check_truth(coeffs):
    s = 1.0 if coeffs[2] > 0 else -1.0
    assert np.abs(coeffs - s * np.array([0, 0, 1, 1.6])).max() < 0.2  # ASSERT_NEAR(..., 0.2), :436-439


@pytest.mark.parametrize("seed", range(4))
def test_reference_ransac_test_holds_for_the_restatement(seed):
    P = capi.params_c0().replace(**REF_TEST_PARAMS)
    ref = make_oracle(P)
    ref.set_cloud(_reference_test_cloud())
    coeffs, inl = ref.estimate_ground_plane(seed)
    _check_truth(coeffs)
    assert 100 < inl.size <= 6000  # refinement threshold 0.05 on a 0.5 m-noise plane keeps a thin slab of the sample


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_reference_ransac_test_on_gpu_matches_restatement(seed):
    P = capi.params_c0().replace(**REF_TEST_PARAMS)
    cloud = _reference_test_cloud()
    est = make_estimator(P)
    gp = RansacPlane(seed=seed)
    est.setInputCloud(cloud, gp)
    assert gp.isSegmented()
    _check_truth(gp.getModelCoeffs())
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    c0, inl0 = ref.estimate_ground_plane(seed)
    assert np.array_equal(gp.getModelCoeffs(), c0)
    assert np.array_equal(gp.getInlinersIndex(), inl0)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [("c0", {}), ("passthrough", dict(ransac_plane_min_z=-3.0, ransac_plane_max_z=-0.5)),
                                     ("no_refinement", dict(ransac_plane_use_refinement=0)),
                                     ("few_iterations", dict(ransac_plane_max_iterations=25))])
def test_estimated_plane_end_to_end(name, kw):
    """setInputCloud(cloud, nullptr) -> RANSAC plane -> depths, HIP vs oracle, on a synthetic frame."""
    P = capi.params_c0().replace(**kw)
    cloud = synth.make_cloud(synth.HDL64, seed=6, frame=4)
    uv = synth.make_features(2000, seed=6)
    est = make_estimator(P)
    depth, types = est.CalculateDepth(cloud, uv, None)  # None: a RansacPlane is created and estimated (:275-283)
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    c0, inl0 = ref.estimate_ground_plane(0)
    assert np.array_equal(est.getGroundPlaneInliers(), inl0)
    assert abs(abs(c0[2]) - 1) < 0.01 and abs(abs(c0[3]) - 1.73) < 0.05  # the scene's ground plane
    d0, t0 = ref.calculate_depth(uv)
    assert_depth_parity(depth, types, d0, t0)
    if name == "c0":
        # only sampled points are inliers (6000 of 131072): few road features find >= 3 of them (reference quirk)
        assert inl0.size <= 6000


@pytest.mark.gpu
def test_too_small_cloud_raises_pcl_invalid():
    est = make_estimator(capi.params_c0())
    cloud = np.array([[1, 0, -1.7, 0], [2, 0, -1.7, 0]], np.float32)
    with pytest.raises(ExceptionPclInvalid):
        est.setInputCloud(cloud, None)
    nan_cloud = np.full((500, 4), np.nan, np.float32)
    with pytest.raises(ExceptionPclInvalid):
        est.setInputCloud(nan_cloud, RansacPlane())


@pytest.mark.gpu
def test_no_horizontal_plane_in_the_cloud():
    """A vertical wall only: every hypothesis fails the perpendicular-plane test (axis z, 10 degrees), so PCL's
    countWithinDistance returns 0 for each; its loop still adopts the first one (0 > -INT_MAX) and the reference ends
    up with that plane's coefficients and no inliers.  Both sides reproduce this."""
    rng = np.random.default_rng(3)
    n = 5000
    wall = np.stack([np.full(n, 10.0), rng.uniform(-20, 20, n), rng.uniform(-2, 3, n), np.zeros(n)], axis=1).astype(np.float32)
    ref = make_oracle(capi.params_c0())
    ref.set_cloud(wall)
    c0, inl0 = ref.estimate_ground_plane(1)
    assert inl0.size == 0 and abs(abs(c0[0]) - 1.0) < 1e-3
    est = make_estimator(capi.params_c0())
    gp = RansacPlane(seed=1)
    est.setInputCloud(wall, gp)
    assert np.array_equal(np.asarray(gp.getModelCoeffs(), dtype=np.float32), c0)
    assert est.getGroundPlaneInliers().size == 0
    # no inliers -> no road fallback succeeds; the main path is unaffected
    uv = synth.make_features(300, seed=4)
    d, t = est.CalculateDepth(uv)
    d0, t0 = ref.calculate_depth(uv)
    assert_depth_parity(d, t, d0, t0)
    assert (t != 16).all()


@pytest.mark.gpu
def test_batched_estimation_every_slot_equals_the_restatement():
    """mld_set_clouds_estimate_planes_device: 16 slots, one launch, no host round trip.  Every slot's coefficients and
    inlier set are bit-identical to the CPU restatement run with the same seed, and the depths computed with the
    estimated planes agree with the oracle end to end (inlier flags ride in the map keys)."""
    import torch
    P = capi.params_c0()
    B, F = 16, 600
    est = make_estimator(P, max_frames=B, max_features=F)
    dev = torch.device("cuda:0")
    scanners = [synth.HDL64_KITTI, synth.HDL64, synth.VLP16]
    clouds = [synth.make_cloud(scanners[b % 3], seed=30 + b % 5, frame=b) for b in range(B)]
    clouds[7] = clouds[7][:2].copy()  # fewer than three points: ExceptionPclInvalid -> that frame runs without a plane
    seeds = [1000 + 17 * b for b in range(B)]
    uvs = [synth.make_features(F, seed=40 + b) for b in range(B)]
    t_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
    t_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
    d = [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(B)]
    t = [torch.empty(F, dtype=torch.int32, device=dev) for _ in range(B)]
    torch.cuda.synchronize()
    est.setInputCloudsEstimatePlanes(t_clouds, seeds)
    est.CalculateDepths(t_uvs, d, t)
    est.synchronize()
    coeffs, n_inl, status = est.getEstimatedPlanes(B)
    for b in range(B):
        ref = make_oracle(P)
        ref.set_cloud(clouds[b])
        if b == 7:
            assert status[b] == 1
            ref.set_ground_plane(None, None)
        else:
            c0, inl0 = ref.estimate_ground_plane(seeds[b])
            assert status[b] == 0
            assert np.array_equal(coeffs[b], c0), b
            assert np.array_equal(est.getGroundPlaneInliers(b), inl0), b
            assert n_inl[b] == inl0.size
        d0, t0 = ref.calculate_depth(uvs[b])
        assert_depth_parity(d[b].cpu().numpy(), t[b].cpu().numpy(), d0, t0)
    # the reference's own tolerance test on the same entry point (test_monolidar_fusion.cpp:376-441)
    P2 = capi.params_c0().replace(**{k: v for k, v in REF_TEST_PARAMS.items() if not k.endswith("_z")})
    e2 = make_estimator(P2, max_frames=2)
    cl = torch.from_numpy(_reference_test_cloud()).to(dev)
    e2.setInputCloudsEstimatePlanes([cl, cl], [3, 4])
    co2, _, st2 = e2.getEstimatedPlanes(2)
    assert (st2 == 0).all()
    _check_truth(co2[0])
    _check_truth(co2[1])


@pytest.mark.gpu
@pytest.mark.parametrize("min_z,max_z", [(-1000.0, 1000.0), (-2.2, -0.9)])
def test_batched_estimation_with_the_z_pass_through(min_z, max_z):
    """The z pass-through (RansacPlane.cpp:57-64) INSIDE the batched kernel: candidate masks per 64 points instead of a
    compacted index list.  With the reference test's own limits (+-1000 m: every finite point passes) and with limits
    that cut the cloud, every slot of a 16-slot batch - clouds of different sizes, NaN "no return" points, one slot with
    fewer than three candidates - has the coefficients, inlier set and count of the CPU restatement, and the depths
    computed with those planes agree with the oracle."""
    import torch
    P = capi.params_c0().replace(**dict(REF_TEST_PARAMS, ransac_plane_min_z=min_z, ransac_plane_max_z=max_z,
                                        ransac_plane_max_iterations=200))
    B, F = 16, 500
    est = make_estimator(P, max_frames=B, max_features=F)
    dev = torch.device("cuda:0")
    scanners = [synth.HDL64_KITTI, synth.HDL64, synth.VLP16]
    clouds = [synth.make_cloud(scanners[b % 3], seed=60 + b % 5, frame=b) for b in range(B)]
    clouds[3] = _reference_test_cloud()
    clouds[5] = clouds[5][:777].copy()          # not a multiple of 64
    clouds[9] = clouds[9].copy()
    clouds[9][:, 2] = 50.0                      # nothing passes a cutting pass-through
    seeds = [2000 + 13 * b for b in range(B)]
    uvs = [synth.make_features(F, seed=70 + b) for b in range(B)]
    t_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
    t_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
    d = [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(B)]
    t = [torch.empty(F, dtype=torch.int32, device=dev) for _ in range(B)]
    torch.cuda.synchronize()
    for rounds in range(2):  # twice: the mask words serve as scratch and must come back clean
        est.setInputCloudsEstimatePlanes(t_clouds, seeds)
        est.CalculateDepths(t_uvs, d, t)
    est.synchronize()
    coeffs, n_inl, status = est.getEstimatedPlanes(B)
    n_failed = 0
    for b in range(B):
        ref = make_oracle(P)
        ref.set_cloud(clouds[b])
        try:
            c0, inl0 = ref.estimate_ground_plane(seeds[b])
        except Exception:  # noqa: BLE001  (the restatement raises where the reference throws ExceptionPclInvalid)
            c0 = None
        if c0 is None:
            n_failed += 1
            assert status[b] == 1, b
            ref.set_ground_plane(None, None)
        else:
            assert status[b] == 0, b
            assert np.array_equal(coeffs[b], c0), b
            assert np.array_equal(est.getGroundPlaneInliers(b), inl0), b
            assert n_inl[b] == inl0.size
        d0, t0 = ref.calculate_depth(uvs[b])
        assert_depth_parity(d[b].cpu().numpy(), t[b].cpu().numpy(), d0, t0)
    assert (n_failed >= 1) if max_z < 0 else (n_failed == 0)  # slot 9 (and any slot with < 3 candidates) runs without a plane


@pytest.mark.gpu
def test_estimated_planes_two_contexts_in_turn():
    """Two contexts in turn with ESTIMATED planes (mld_order_after): the next context's estimation and projection are
    released by the end of this one's projection and run beside these feature kernels.  Six launch sets, each equal to
    the restatement whichever context computed it; the same slots are re-estimated and re-projected while the other
    context's kernels are in flight."""
    import torch
    P = capi.params_c0()
    S, F, rounds = 5, 900, 3
    dev = torch.device("cuda:0")
    ests = [make_estimator(P, max_frames=S, max_features=F) for _ in range(2)]
    for e in ests:
        e.setSharedGpu(True)
    sets = []
    for r in range(rounds * 2):
        clouds = [synth.make_cloud(synth.HDL64_KITTI if (r + b) % 2 else synth.HDL64, seed=500 + r, frame=b) for b in range(S)]
        uvs = [synth.make_features(F, seed=600 + 10 * r + b) for b in range(S)]
        seeds = [7000 + 31 * r + b for b in range(S)]
        sets.append((clouds, uvs, seeds, [torch.from_numpy(c).to(dev) for c in clouds],
                     [torch.from_numpy(u).to(dev) for u in uvs],
                     [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(S)],
                     [torch.empty(F, dtype=torch.int32, device=dev) for _ in range(S)]))
    torch.cuda.synchronize()
    for r, (clouds, uvs, seeds, tc, tu, d, t) in enumerate(sets):
        e, nxt = ests[r % 2], ests[(r + 1) % 2]
        e.setInputCloudsEstimatePlanes(tc, seeds)
        nxt.orderAfter(e)
        e.CalculateDepths(tu, d, t)
    for e in ests:
        e.synchronize()
    for r, (clouds, uvs, seeds, tc, tu, d, t) in enumerate(sets):
        for b in range(r % 2, S, 2):
            ref = make_oracle(P)
            ref.set_cloud(clouds[b])
            ref.estimate_ground_plane(seeds[b])
            d0, t0 = ref.calculate_depth(uvs[b])
            assert_depth_parity(d[b].cpu().numpy(), t[b].cpu().numpy(), d0, t0)
    for e in ests:
        e.close()


def _boundary_clouds():
    """Clouds that walk k_rs_batch's branches: samples shorter than a wavefront / not a multiple of four / exactly the
    sample size, draws that are mostly skipped (NaN points) or never valid (a wall), duplicates, collinear points."""
    rng = np.random.default_rng(77)

    def plane(n, noise=0.02, z0=-1.7):
        cl = np.zeros((n, 4), np.float32)
        cl[:, 0] = rng.uniform(2, 40, n)
        cl[:, 1] = rng.uniform(-15, 15, n)
        cl[:, 2] = z0 + 0.01 * cl[:, 0] + rng.normal(0, noise, n)
        return cl

    clouds = [plane(n) for n in (3, 5, 63, 64, 65, 130, 257, 5999, 6000, 6001)]
    nan_heavy = plane(20000)
    nan_heavy[rng.random(20000) < 0.6] = np.nan  # most draws hit a NaN point and are skipped
    wall = np.zeros((9000, 4), np.float32)  # a vertical wall: no draw gives a model within 10 degrees of horizontal
    wall[:, 0] = 10.0 + rng.normal(0, 0.01, 9000)
    wall[:, 1] = rng.uniform(-10, 10, 9000)
    wall[:, 2] = rng.uniform(-2, 3, 9000)
    dup = np.repeat(plane(40), 50, axis=0)  # every point fifty times: many degenerate triples
    line = np.zeros((500, 4), np.float32)   # collinear points
    line[:, 0] = np.linspace(1, 50, 500)
    line[:, 2] = -1.7
    weak = plane(12000, noise=0.02)         # a third of the points on the plane, the rest clutter: long runs of draws
    clutter = rng.random(12000) < 0.67
    weak[clutter, 2] = rng.uniform(-1.5, 4.0, clutter.sum())
    return clouds + [nan_heavy, wall, dup, line, weak, synth.make_cloud(synth.HDL64, seed=1, frame=14)]


@pytest.mark.gpu
@pytest.mark.parametrize("max_it,prob,z_limits", [
    (0, 0.99, None), (1, 0.99, None), (15, 0.99, None), (16, 0.99, None), (63, 0.99, None), (64, 0.99, None),
    (65, 0.99, None), (1023, 0.999999, None), (1087, 0.999999, None), (1088, 0.999999, None), (10000, 0.99, None),
    (10000, 0.5, None), (3000, 0.999999999, None),
    # the same with the z pass-through ahead of the sub-sampling (RansacPlane.cpp:57-64)
    (64, 0.99, (-1.9, 0.5)), (1088, 0.999999, (-1.9, 0.5)), (10000, 0.99, (-1000.0, 1000.0))])
def test_batched_estimation_boundaries(max_it, prob, z_limits):
    """Every branch of the batched estimator against the restatement, bit for bit: draw budgets at the edges of its
    epochs (64 draws, then 1024) and rounds (16 valid draws, then two per wavefront), stopping probabilities that end a
    slot in the first round or keep it going for thousands of draws, samples of every raggedness, clouds whose draws
    are skipped or never valid.  Slots the restatement cannot estimate must carry status 1."""
    import torch
    P = capi.params_c0().replace(ransac_plane_max_iterations=max_it, ransac_plane_probability=prob)
    if z_limits:
        P = P.replace(ransac_plane_min_z=z_limits[0], ransac_plane_max_z=z_limits[1])
    clouds = _boundary_clouds()
    B = len(clouds)
    dev = torch.device("cuda:0")
    est = make_estimator(P, max_frames=B, max_features=64)
    seeds = [911 + 13 * b + max_it for b in range(B)]
    est.setInputCloudsEstimatePlanes([torch.from_numpy(c).to(dev) for c in clouds], seeds)
    coeffs, n_inl, status = est.getEstimatedPlanes(B)
    n_ok = 0
    for b in range(B):
        ref = make_oracle(P)
        ref.set_cloud(clouds[b])
        try:
            c0, inl0 = ref.estimate_ground_plane(seeds[b])
        except RuntimeError:
            assert status[b] == 1, (b, status[b])
            continue
        n_ok += 1
        assert status[b] == 0, b
        assert np.array_equal(coeffs[b], c0), (b, coeffs[b], c0)
        assert n_inl[b] == inl0.size, b
        assert np.array_equal(est.getGroundPlaneInliers(b), inl0), b
    assert n_ok >= B - 6
    est.close()

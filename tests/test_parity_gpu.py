"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs."""
import numpy as np
import pytest

from mono_lidar_depth_amd import NO_PLANE, GroundPlane, capi, synth

from helpers import assert_depth_parity, kitti_camera, make_estimator, make_oracle, run_oracle

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("feature_kernel_path")]


def _frame(scanner, seed, nfeat, frame=0, integer=False, stride=4):
    cloud = synth.make_cloud(scanner, seed=seed, frame=frame, stride_floats=stride)
    uv = synth.make_features(nfeat, seed=seed, integer=integer)
    plane = synth.make_ground_plane(cloud)
    return cloud, uv, plane


def test_projection_pixelmap_bit_exact():
    """_pointIndex, visible image points, camera-frame cloud and the pixel map are bit-identical."""
    P = capi.params_c0()
    cloud, uv, plane = _frame(synth.HDL64, 0, 16)
    est = make_estimator(P)
    est.setInputCloud(cloud, GroundPlane(*plane))
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    assert est.getVisibleCount() == ref.nvis
    assert np.array_equal(est.getPointIndex(), ref.point_index())
    assert np.array_equal(est.getPointsCloudImageCs(), ref.visible_image_points())
    cam_gpu, cam_ref = est.getCloudCameraCs(cloud.shape[0]), ref.cloud_camera_cs()
    assert np.array_equal(np.isnan(cam_gpu), np.isnan(cam_ref))
    assert np.array_equal(np.nan_to_num(cam_gpu), np.nan_to_num(cam_ref))
    assert np.array_equal(est.getPixelMap(), ref.pixel_map())
    k = ref.nvis // 2
    assert est.getPointDepthCamVisible(k) == ref.cloud_camera_cs()[2, ref.point_index()[k]]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_config2_depth_parity(seed):
    """BASELINE config 2: 64x2048 cloud, 2000 features, C0 parameters."""
    P = capi.params_c0()
    cloud, uv, plane = _frame(synth.HDL64, seed, 2000, frame=seed * 7)
    est = make_estimator(P)
    depth, types = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    _, (d0, t0) = run_oracle(P, cloud, uv, plane)
    assert_depth_parity(depth, types, d0, t0)
    assert (types == 1).sum() > 50 and (types == 16).sum() > 50


def test_pcl_stride_and_repeat_calls():
    """32-byte pcl::PointXYZI records; CalculateDepth twice on one cloud; a second cloud on the same slot."""
    P = capi.params_c0()
    cloud8, uv, plane = _frame(synth.HDL64_KITTI, 5, 500, stride=8)
    est = make_estimator(P)
    gp = GroundPlane(*plane)
    d1, t1 = est.CalculateDepth(cloud8, uv, gp)
    d2, t2 = est.CalculateDepth(uv)
    assert np.array_equal(d1, d2, equal_nan=True) and np.array_equal(t1, t2)
    _, (d0, t0) = run_oracle(P, cloud8, uv, plane)
    assert_depth_parity(d1, t1, d0, t0)
    # new cloud on the same slot: stale map entries must not leak
    cloudb, uvb, planeb = _frame(synth.HDL64_KITTI, 6, 500, frame=40)
    d3, t3 = est.CalculateDepth(cloudb, uvb, GroundPlane(*planeb))
    _, (d0b, t0b) = run_oracle(P, cloudb, uvb, planeb)
    assert_depth_parity(d3, t3, d0b, t0b)


def test_tag_wraparound():
    """More than 127 (kMaxTag) setInputCloud calls on one slot: the map tag wraps and the map is re-zeroed."""
    P = capi.params_c0()
    sc = synth.Scanner(16, 360, 2.0, -24.9)
    est = make_estimator(P)
    clouds = [synth.make_cloud(sc, seed=9, frame=f) for f in range(4)]
    uv = synth.make_features(256, seed=9)
    for it in range(300):
        cl = clouds[it % 4]
        est.setInputCloud(cl, NO_PLANE)
    cl = clouds[299 % 4]
    depth, types = est.CalculateDepth(uv)
    _, (d0, t0) = run_oracle(P, cl, uv, None)
    assert_depth_parity(depth, types, d0, t0)


@pytest.mark.parametrize("tmode,lmode,ltype", [(0, 0, 1), (1, 0, 0), (0, 1, 0), (1, 1, 1)])
def test_config3_sparse_cloud_threshold_modes(tmode, lmode, ltype):
    """BASELINE config 3: VLP-16 sparse cloud, 5000 features, threshold treatment modes swept."""
    P = capi.params_c0().replace(treshold_depth_mode=tmode, treshold_depth_local_mode=lmode,
                                 treshold_depth_local_valuetype=ltype, treshold_depth_max=30,
                                 treshold_depth_min=4, treshold_depth_local_value=0.05)
    cloud, uv, plane = _frame(synth.VLP16, 11, 5000)
    est = make_estimator(P)
    depth, types = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    _, (d0, t0) = run_oracle(P, cloud, uv, plane)
    assert_depth_parity(depth, types, d0, t0)


def test_wide_window_many_neighbours():
    """Large window (20x24 -> up to 572 cells, road window 42x38): multi-chunk lists and pair search."""
    P = capi.params_default().replace(viewray_plane_orthoganality_treshold=0.03, radiusSearch_count_min=3,
                                      pixelarea_search_witdh=20, pixelarea_search_height=24)
    cloud, uv, plane = _frame(synth.DENSE128, 21, 1500)
    est = make_estimator(P)
    depth, types = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    ref, (d0, t0) = run_oracle(P, cloud, uv, plane)
    assert_depth_parity(depth, types, d0, t0)
    traces = [ref.trace_feature(*uv[i]) for i in range(0, 1500, 10)]
    assert max(len(t["nb_idx"]) for t in traces) > 64, "test should exercise lists longer than one wavefront"
    assert max(len(t["road_idx"]) for t in traces) > 128


MODE_CASES = {
    "no_plane": dict(_plane=False),
    "no_ransac": dict(do_use_ransac_plane=0),
    "no_hist": dict(do_use_histogram_segmentation=0),
    "no_trimax": dict(do_use_triangle_size_maximation=0),
    "no_planar_check": dict(do_check_triangleplanar_condition=0),
    "normal_intersection": dict(viewray_plane_orthoganality_treshold=0.0),
    "header_orth_1": dict(viewray_plane_orthoganality_treshold=1.0),
    "no_thresholds": dict(treshold_depth_enabled=0, treshold_depth_local_enabled=0, do_use_cut_behind_camera=0),
    "count_min_5": dict(radiusSearch_count_min=5),
    "hist_min1_bin1": dict(histogram_segmentation_min_pointcount=1, histogram_segmentation_bin_witdh=1.0),
    "hist_min0": dict(histogram_segmentation_min_pointcount=0),
    "pca": dict(do_use_PCA=1, pca_treshold_2_1_rel_min=0.5),
    "road_triangle": dict(plane_estimator_use_triangle_maximation=1, plane_estimator_use_mestimator=0,
                          plane_estimator_z_x_min_relation=0.2),
    "zero": dict(set_all_depths_to_zero=1),
}


@pytest.mark.parametrize("name", sorted(MODE_CASES))
def test_parameter_modes(name):
    kw = dict(MODE_CASES[name])
    with_plane = kw.pop("_plane", True)
    P = capi.params_c0().replace(**kw)
    cloud, uv, plane = _frame(synth.HDL64_KITTI, 31, 1200)
    est = make_estimator(P)
    gp = GroundPlane(*plane) if with_plane else NO_PLANE
    depth, types = est.CalculateDepth(cloud, uv, gp)
    _, (d0, t0) = run_oracle(P, cloud, uv, plane if with_plane else None)
    exact = name not in ("pca",)
    assert_depth_parity(depth, types, d0, t0, exact_main=exact)


def test_edge_features_and_empty_inputs():
    """Features on / outside the image border, integer pixels, F = 0, an empty cloud, an all-NaN cloud."""
    P = capi.params_c0()
    cloud, _, plane = _frame(synth.HDL64_KITTI, 41, 1)
    W, H = synth.KITTI_W, synth.KITTI_H
    uv = np.array([[0, 0], [W - 1, H - 1], [W, H], [-3.5, 200], [W + 2.9, 300], [600.5, -4.4], [600.5, H + 4.4],
                   [0.49, 374.51], [1241.99, 0.01], [-100, -100], [5000, 5000], [620, 250], [3.0, 371.0]], dtype=np.float64)
    uv = np.concatenate([uv, synth.make_features(200, seed=41, integer=True)])
    est = make_estimator(P)
    depth, types = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    _, (d0, t0) = run_oracle(P, cloud, uv, plane)
    assert_depth_parity(depth, types, d0, t0)
    # F = 0
    d, t = est.CalculateDepth(np.zeros((0, 2)))
    assert d.shape == (0,) and t.shape == (0,)
    # empty cloud and all-NaN cloud: every feature is RadiusSearchInsufficientPoints
    for cl in (np.zeros((0, 4), np.float32), np.full((100, 4), np.nan, np.float32)):
        d, t = est.CalculateDepth(cl, uv[:20], NO_PLANE)
        assert (t == 2).all() and (d == -1).all()


def test_usage_errors():
    """Error behaviour of the reference's usage errors (DepthEstimator.cpp:38,57,94,439,608)."""
    from mono_lidar_depth_amd import DepthEstimator, DepthEstimatorError
    est = DepthEstimator()
    with pytest.raises(DepthEstimatorError, match="InitConfig"):
        est.Initialize(kitti_camera(), synth.T_CAM_LIDAR)
    for kw, code in ((dict(neighbor_search_mode=1), capi.MLD_ERR_UNSUPPORTED_MODE),
                     (dict(do_use_depth_segmentation=1), capi.MLD_ERR_UNSUPPORTED_MODE),
                     (dict(plane_estimator_use_mestimator=0), capi.MLD_ERR_NO_ROAD_ESTIMATOR)):
        e2 = DepthEstimator()
        e2.InitConfig(capi.params_c0().replace(**kw))
        with pytest.raises(DepthEstimatorError) as ei:
            e2.Initialize(kitti_camera(), synth.T_CAM_LIDAR)
        assert ei.value.code == code
    e3 = make_estimator(capi.params_c0())
    with pytest.raises(DepthEstimatorError, match="without 'SetInputCloud'"):
        e3.CalculateDepth(np.zeros((4, 2)))
    cloud = synth.make_cloud(synth.Scanner(8, 90, 2, -20), 1)
    e3.setInputCloud(cloud, None, plane_given=False)
    with pytest.raises(DepthEstimatorError) as ei:
        e3.CalculateDepth(np.zeros((4, 2)))
    assert ei.value.code == capi.MLD_ERR_NO_GROUND_PLANE


@pytest.mark.parametrize("B", [16, 13])
def test_batched_slots_match_single_slot(B):
    """Frame slots: B frames in one launch set (torch device tensors) == one frame at a time.  13 exercises the
    slots beyond the last full group of 8 (plain block order after the XCD-interleaved groups)."""
    import torch
    P = capi.params_c0()
    F = 700
    sc = synth.HDL64_KITTI
    est = make_estimator(P, max_frames=B)
    clouds = [synth.make_cloud(sc, seed=50 + (b % 3), frame=b) for b in range(B)]
    uvs = [synth.make_features(F + 13 * b, seed=60 + b) for b in range(B)]
    planes = [synth.make_ground_plane(c) for c in clouds]
    dev = torch.device("cuda:0")
    t_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
    t_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
    t_inl = [torch.from_numpy(p[1]).to(dev) for p in planes]
    t_depth = [torch.full((u.shape[0],), 7.0, dtype=torch.float64, device=dev) for u in uvs]
    t_type = [torch.full((u.shape[0],), -7, dtype=torch.int32, device=dev) for u in uvs]
    torch.cuda.synchronize()
    est.setInputClouds(t_clouds, 16)
    for b in range(B):
        est.setGroundPlane(GroundPlane(planes[b][0], t_inl[b]), slot=b)
    est.CalculateDepths(t_uvs, t_depth, t_type)
    est.synchronize()
    for b in range(B):
        _, (d0, t0) = run_oracle(P, clouds[b], uvs[b], planes[b])
        assert_depth_parity(t_depth[b].cpu().numpy(), t_type[b].cpu().numpy(), d0, t0)


def test_planes_known_at_projection_equal_planes_set_afterwards():
    """setInputCloud(cloud, groundPlane) with the plane known at projection time (mld_set_clouds_planes_device: the
    inlier flags travel in the pixel-map keys) against the plane installed after the cloud (mask lookups), and both
    against the oracle; then a DIFFERENT plane set after the combined call must win over the flags in the keys."""
    import torch
    P = capi.params_c0()
    B, F = 11, 900
    est = make_estimator(P, max_frames=B, max_features=F)
    clouds = [synth.make_cloud(synth.HDL64_KITTI, seed=70 + (b % 4), frame=b) for b in range(B)]
    uvs = [synth.make_features(F, seed=80 + b) for b in range(B)]
    planes = [synth.make_ground_plane(c) for c in clouds]
    dev = torch.device("cuda:0")

    def mask_of(inl, n):
        m = np.zeros((n + 31) // 32, dtype=np.uint32)
        np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
        return torch.from_numpy(m.view(np.int32)).to(dev)

    t_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
    t_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
    t_masks = [mask_of(p[1], c.shape[0]) for p, c in zip(planes, clouds)]
    coeffs = np.stack([p[0] for p in planes])
    d_a = [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(B)]
    t_a = [torch.empty(F, dtype=torch.int32, device=dev) for _ in range(B)]
    torch.cuda.synchronize()
    batch = est.prepareBatch(t_clouds, t_uvs, d_a, t_a, coeffs, t_masks)
    est.runBatch(batch)  # combined entry point
    est.synchronize()
    res_a = [(d.cpu().numpy().copy(), t.cpu().numpy().copy()) for d, t in zip(d_a, t_a)]
    est.setInputClouds(t_clouds, 16)  # plane after the cloud
    est.setGroundPlanesMask(coeffs, t_masks)
    est.CalculateDepths(t_uvs, d_a, t_a)
    est.synchronize()
    for b in range(B):
        _, (d0, t0) = run_oracle(P, clouds[b], uvs[b], planes[b])
        assert_depth_parity(res_a[b][0], res_a[b][1], d0, t0)
        assert np.array_equal(res_a[b][1], t_a[b].cpu().numpy())
        assert np.array_equal(res_a[b][0], d_a[b].cpu().numpy(), equal_nan=True)
    # a plane with half the inliers, installed after the combined call: the keys' flags are stale and must be ignored
    half = [(p[0], p[1][::2].copy()) for p in planes]
    est.runBatch(batch)
    est.setGroundPlanesMask(coeffs, [mask_of(h[1], c.shape[0]) for h, c in zip(half, clouds)])
    est.CalculateDepths(t_uvs, d_a, t_a)
    est.synchronize()
    for b in range(0, B, 5):
        _, (d0, t0) = run_oracle(P, clouds[b], uvs[b], half[b])
        assert_depth_parity(d_a[b].cpu().numpy(), t_a[b].cpu().numpy(), d0, t0)


def test_config5_dense_cloud_integer_features():
    """BASELINE config 5: 128x4096 cloud (524 288 points), 10 000 integer-pixel features (tracklet-style)."""
    P = capi.params_c0()
    cloud, uv, plane = _frame(synth.DENSE128, 77, 10000, integer=True)
    est = make_estimator(P)
    depth, types = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    _, (d0, t0) = run_oracle(P, cloud, uv, plane, n_threads=8)
    assert_depth_parity(depth, types, d0, t0)
    assert (types == 1).sum() > 500


def test_full_size_properties_config2():
    """Size-independent properties at full size: idempotence (same cloud twice), permutation of the feature
    order permutes the outputs, and a feature's result does not depend on the other features in the call."""
    P = capi.params_c0()
    cloud, uv, plane = _frame(synth.HDL64, 88, 2000)
    est = make_estimator(P)
    gp = GroundPlane(*plane)
    d1, t1 = est.CalculateDepth(cloud, uv, gp)
    d2, t2 = est.CalculateDepth(cloud, uv, gp)
    assert np.array_equal(d1, d2, equal_nan=True) and np.array_equal(t1, t2)
    perm = np.random.default_rng(0).permutation(len(uv))
    d3, t3 = est.CalculateDepth(uv[perm])
    assert np.array_equal(d3, d1[perm], equal_nan=True) and np.array_equal(t3, t1[perm])
    d4, t4 = est.CalculateDepth(uv[:137])
    assert np.array_equal(d4, d1[:137], equal_nan=True) and np.array_equal(t4, t1[:137])
    # successes lie inside the global window and inside the local z-range tolerance by construction
    ok = t1 == 1
    assert (d1[ok] >= 0).all() and (d1[ok] <= 100).all()
    assert ((t1 != 1) & (t1 != 16) == (d1 == -1)).all()


def test_batch_with_empty_and_ragged_slots():
    """One launch set with ragged slots: no features in one slot, an empty cloud in another, a slot without a ground
    plane, different cloud sizes and feature counts everywhere else."""
    import torch
    P = capi.params_c0()
    B = 8
    dev = torch.device("cuda:0")
    scanners = [synth.HDL64_KITTI, synth.VLP16, synth.HDL64, synth.HDL64_KITTI, synth.VLP16, synth.HDL64_KITTI,
                synth.HDL64, synth.VLP16]
    clouds = [synth.make_cloud(scanners[b], seed=90 + b, frame=b) for b in range(B)]
    clouds[5] = np.zeros((0, 4), dtype=np.float32)
    uvs = [synth.make_features(300 + 97 * b, seed=95 + b) for b in range(B)]
    uvs[2] = np.zeros((0, 2), dtype=np.float64)
    planes = [synth.make_ground_plane(c) if c.shape[0] else (np.array([0, 0, 1, 1.73], np.float32), np.zeros(0, np.int32))
              for c in clouds]
    est = make_estimator(P, max_frames=B)
    t_clouds = [torch.from_numpy(c).to(dev) if c.shape[0] else torch.zeros((1, 4), dtype=torch.float32, device=dev)[:0]
                for c in clouds]
    t_uvs = [torch.from_numpy(u).to(dev) if u.shape[0] else torch.zeros((1, 2), dtype=torch.float64, device=dev)[:0]
             for u in uvs]
    t_depth = [torch.full((max(1, u.shape[0]),), 7.0, dtype=torch.float64, device=dev)[:u.shape[0]] for u in uvs]
    t_type = [torch.full((max(1, u.shape[0]),), -7, dtype=torch.int32, device=dev)[:u.shape[0]] for u in uvs]
    torch.cuda.synchronize()
    est.setInputClouds(t_clouds, 16)
    for b in range(B):
        if b == 7:
            est.setGroundPlane(NO_PLANE, slot=b)
        else:
            est.setGroundPlane(GroundPlane(planes[b][0], torch.from_numpy(planes[b][1]).to(dev)), slot=b)
    est.CalculateDepths(t_uvs, t_depth, t_type)
    est.synchronize()
    for b in range(B):
        if uvs[b].shape[0] == 0:
            continue
        if clouds[b].shape[0] == 0:
            assert (t_type[b].cpu().numpy() == 2).all() and (t_depth[b].cpu().numpy() == -1).all()
            continue
        _, (d0, t0) = run_oracle(P, clouds[b], uvs[b], None if b == 7 else planes[b])
        assert_depth_parity(t_depth[b].cpu().numpy(), t_type[b].cpu().numpy(), d0, t0)


@pytest.mark.parametrize("paired,handover", [(False, "classify"), (False, "projection"), (True, "classify")])
def test_two_contexts_alternating_on_one_gpu(paired, handover):
    """Two contexts used in turn (mld_order_after[_classify] or mld_pair_contexts, + mld_set_shared_gpu; include/mld.h "Two
    contexts"): the projection of one runs beside the feature kernels of the other.  Every launch set of four rounds
    equals the oracle, whichever context computed it, and the shared-GPU mode (more LDS per block of the lane-per-feature
    kernel) changes nothing.  Paired: both contexts' projections on one stream; the same slots are re-projected while
    the previous round's feature kernels may still be queued (the fork event orders them)."""
    import torch
    P = capi.params_c0()
    S, F, rounds = 6, 1200, 4
    dev = torch.device("cuda:0")
    ests = [make_estimator(P, max_frames=S, max_features=F) for _ in range(2)]
    for e in ests:
        e.setSharedGpu(True)
    if paired:
        ests[0].pairWith(ests[1])

    def mask_of(inl, n):
        m = np.zeros((n + 31) // 32, dtype=np.uint32)
        np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
        return torch.from_numpy(m.view(np.int32)).to(dev)

    sets = []
    for r in range(rounds * 2):  # one launch set per (round, context)
        clouds = [synth.make_cloud(synth.HDL64_KITTI, seed=300 + r, frame=b) for b in range(S)]
        planes = [synth.make_ground_plane(c) for c in clouds]
        uvs = [synth.make_features(F, seed=400 + 10 * r + b) for b in range(S)]
        e = ests[r % 2]
        d = [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(S)]
        t = [torch.empty(F, dtype=torch.int32, device=dev) for _ in range(S)]
        batch = e.prepareBatch([torch.from_numpy(c).to(dev) for c in clouds], [torch.from_numpy(u).to(dev) for u in uvs],
                               d, t, np.stack([p[0] for p in planes]),
                               [mask_of(p[1], c.shape[0]) for p, c in zip(planes, clouds)])
        sets.append((e, batch, clouds, planes, uvs, d, t))
    torch.cuda.synchronize()
    for r, (e, batch, *_rest) in enumerate(sets):
        e.runBatchBeside(batch, ests[(r + 1) % 2], handover)
    for e in ests:
        e.synchronize()
    for e, batch, clouds, planes, uvs, d, t in sets:
        for b in range(0, S, 2):
            _, (d0, t0) = run_oracle(P, clouds[b], uvs[b], planes[b])
            assert_depth_parity(d[b].cpu().numpy(), t[b].cpu().numpy(), d0, t0)
    # argument checks of the C entry point
    lib = ests[0]._lib
    assert lib.mld_order_after(ests[0]._ctx, None) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_order_after(ests[0]._ctx, ests[0]._ctx) == capi.MLD_OK
    assert lib.mld_order_after_classify(ests[0]._ctx, None) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_order_after_classify(None, ests[0]._ctx) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_order_after_classify(ests[0]._ctx, ests[0]._ctx) == capi.MLD_OK
    # a pending hand-over (the other context never issues its feature kernels) ends with either context
    assert lib.mld_order_after_classify(ests[0]._ctx, ests[1]._ctx) == capi.MLD_OK
    assert lib.mld_order_after_classify(ests[1]._ctx, ests[0]._ctx) == capi.MLD_OK
    assert lib.mld_pair_contexts(ests[0]._ctx, None) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_pair_contexts(ests[0]._ctx, ests[0]._ctx) == capi.MLD_ERR_INVALID_ARG
    if paired:
        assert lib.mld_pair_contexts(ests[0]._ctx, ests[1]._ctx) == capi.MLD_ERR_INVALID_ARG  # already paired
    ests[1].close()  # the borrower of the projection stream first
    ests[0].close()


@pytest.mark.parametrize("mixed", [False, True])
def test_clouds_off_16_byte_boundaries(mixed):
    """Device clouds that start 4 bytes past a 16-byte boundary (a view into a larger buffer): the projection takes its
    scalar-load instantiation (`k_project_scatter<false>`; the 16-byte one is chosen only when every cloud of the launch
    is aligned).  `mixed`: aligned and unaligned clouds in one launch set.  Same results as the oracle either way."""
    import torch
    P = capi.params_c0()
    B, F = 10, 600
    dev = torch.device("cuda:0")
    est = make_estimator(P, max_frames=B)
    clouds = [synth.make_cloud(synth.HDL64_KITTI, seed=70 + b, frame=b) for b in range(B)]
    planes = [synth.make_ground_plane(c) for c in clouds]
    uvs = [synth.make_features(F, seed=80 + b) for b in range(B)]
    t_clouds, keep = [], []
    for b, c in enumerate(clouds):
        off = 0 if (mixed and b % 2 == 0) else 1
        buf = torch.zeros(c.size + 4, dtype=torch.float32, device=dev)
        assert buf.data_ptr() % 16 == 0
        view = buf[off:off + c.size].view(c.shape)
        view.copy_(torch.from_numpy(c))
        assert (view.data_ptr() % 16 == 0) == (off == 0)
        keep.append(buf)
        t_clouds.append(view)
    t_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
    t_inl = [torch.from_numpy(p[1]).to(dev) for p in planes]
    t_depth = [torch.full((F,), 7.0, dtype=torch.float64, device=dev) for _ in range(B)]
    t_type = [torch.full((F,), -7, dtype=torch.int32, device=dev) for _ in range(B)]
    torch.cuda.synchronize()
    est.setInputClouds(t_clouds, 16)
    for b in range(B):
        est.setGroundPlane(GroundPlane(planes[b][0], t_inl[b]), slot=b)
    est.CalculateDepths(t_uvs, t_depth, t_type)
    est.synchronize()
    for b in range(B):
        _, (d0, t0) = run_oracle(P, clouds[b], uvs[b], planes[b])
        assert_depth_parity(t_depth[b].cpu().numpy(), t_type[b].cpu().numpy(), d0, t0)
    # one frame at a time on an unaligned cloud (single-slot launch)
    est1 = make_estimator(P)
    est1.setInputCloud(t_clouds[1], GroundPlane(planes[1][0], t_inl[1]))
    d1, t1 = est1.CalculateDepth(t_uvs[1])
    _, (d0, t0) = run_oracle(P, clouds[1], uvs[1], planes[1])
    assert_depth_parity(d1.cpu().numpy(), t1.cpu().numpy(), d0, t0)


@pytest.mark.parametrize("destroy_first", ["released", "releasing"])
def test_pending_gate_survives_either_context_going_first(destroy_first):
    """mld_order_after_classify hands over through a polling wavefront queued in front of the released context's NEXT
    projection.  Here that projection never comes: the classification has run (the gate's counter and target are set on
    the released context) and one of the two contexts is destroyed - the other must neither poll freed memory nor keep a
    pointer to the dead context (it goes on to compute a correct batch)."""
    import torch
    P = capi.params_c0()
    S, F = 4, 300
    dev = torch.device("cuda:0")
    a, b = make_estimator(P, max_frames=S, max_features=F), make_estimator(P, max_frames=S, max_features=F)

    def batch(e, seed):
        clouds = [synth.make_cloud(synth.HDL64_KITTI, seed=seed, frame=i) for i in range(S)]
        planes = [synth.make_ground_plane(c) for c in clouds]
        uvs = [synth.make_features(F, seed=seed + i) for i in range(S)]
        d = [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(S)]
        t = [torch.empty(F, dtype=torch.int32, device=dev) for _ in range(S)]
        masks = []
        for p, c in zip(planes, clouds):
            m = np.zeros((c.shape[0] + 31) // 32, dtype=np.uint32)
            np.bitwise_or.at(m, p[1] >> 5, (np.uint32(1) << (p[1] & 31).astype(np.uint32)))
            masks.append(torch.from_numpy(m.view(np.int32)).to(dev))
        pb = e.prepareBatch([torch.from_numpy(c).to(dev) for c in clouds], [torch.from_numpy(u).to(dev) for u in uvs], d, t,
                            np.stack([p[0] for p in planes]), masks)
        return pb, clouds, planes, uvs, d, t

    ba = batch(a, 500)
    torch.cuda.synchronize()
    a.runBatchBeside(ba[0], b)   # b is released behind a's classification: the gate is now pending on b
    a.synchronize()
    survivor = a if destroy_first == "released" else b
    (b if destroy_first == "released" else a).close()
    bs = batch(survivor, 600)
    torch.cuda.synchronize()
    survivor.runBatch(bs[0])     # (b: its pending gate must have been dropped with a's counter)
    survivor.synchronize()
    _, clouds, planes, uvs, d, t = bs
    for i in range(S):
        _, (d0, t0) = run_oracle(P, clouds[i], uvs[i], planes[i])
        assert_depth_parity(d[i].cpu().numpy(), t[i].cpu().numpy(), d0, t0)
    survivor.close()


def test_rearmed_waiter_leaves_no_dangling_back_pointer():
    """A context that holds an unconsumed gate on one context and is then re-armed behind ANOTHER one
    (mld_order_after_classify again, before any projection of its own consumed the first hand-over): destroying it must
    clear its registration at BOTH - the second context then goes on to classify and compute a correct batch (its
    release_waiter used to point at freed memory), and so does the first."""
    import torch
    P = capi.params_c0()
    S, F = 3, 250
    dev = torch.device("cuda:0")
    a, b, w = (make_estimator(P, max_frames=S, max_features=F) for _ in range(3))

    def batch(e, seed):
        clouds = [synth.make_cloud(synth.HDL64_KITTI, seed=seed, frame=i) for i in range(S)]
        planes = [synth.make_ground_plane(c) for c in clouds]
        uvs = [synth.make_features(F, seed=seed + i) for i in range(S)]
        d = [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(S)]
        t = [torch.empty(F, dtype=torch.int32, device=dev) for _ in range(S)]
        masks = []
        for p, c in zip(planes, clouds):
            m = np.zeros((c.shape[0] + 31) // 32, dtype=np.uint32)
            np.bitwise_or.at(m, p[1] >> 5, (np.uint32(1) << (p[1] & 31).astype(np.uint32)))
            masks.append(torch.from_numpy(m.view(np.int32)).to(dev))
        pb = e.prepareBatch([torch.from_numpy(c).to(dev) for c in clouds], [torch.from_numpy(u).to(dev) for u in uvs], d, t,
                            np.stack([p[0] for p in planes]), masks)
        return pb, clouds, planes, uvs, d, t

    ba, bb = batch(a, 700), batch(b, 800)
    torch.cuda.synchronize()
    a.runBatchBeside(ba[0], w)     # w's gate is pending on a's counter (w never projects: the gate stays unconsumed)
    a.synchronize()
    w.orderAfterClassify(b)        # ... and w is re-armed behind b
    w.close()                      # both registrations must die with w
    for e, bt in ((b, bb), (a, batch(a, 900))):
        torch.cuda.synchronize()
        e.runBatch(bt[0])          # (b: launch_features consults its release_waiter)
        e.synchronize()
        _, clouds, planes, uvs, d, t = bt
        for i in range(S):
            _, (d0, t0) = run_oracle(P, clouds[i], uvs[i], planes[i])
            assert_depth_parity(d[i].cpu().numpy(), t[i].cpu().numpy(), d0, t0)
    a.close()
    b.close()

"""The reference's PRODUCTION call as one C call: CalculateDepth(cloud, uv, depths, types, groundPlane) with a GroundPlane
that is not segmented yet (TrackletDepthModule::process builds a fresh one per frame, tracklet_depth_module.cpp:269-284;
setInputCloud estimates it, DepthEstimator.cpp:275-283) -> mld_calculate_depth_frame_estimate: the plane is estimated on
the GPU ahead of the projection (RANSAC: one block; semantic: the label-image kernels), the points' ground-plane state
rides in the pixel-map keys, nothing returns to the host before the depths do.  Checked against the oracle with the
restatement's plane for the same seed / label image."""
import ctypes as C

import numpy as np
import pytest

from mono_lidar_depth_amd import (DepthEstimatorError, ExceptionPclInvalid, GroundPlane, RansacPlane, SemanticPlane, capi,
                                  synth)

from helpers import assert_depth_parity, make_estimator, make_oracle

pytestmark = pytest.mark.gpu
LABELS = (6, 7, 8, 9)  # tracklet_depth_module.cpp:280


@pytest.mark.parametrize("name,kw,seed", [("c0", {}, 0), ("c0_seed7", {}, 7),
                                          ("passthrough", dict(ransac_plane_min_z=-3.0, ransac_plane_max_z=-0.5), 3),
                                          ("no_refinement", dict(ransac_plane_use_refinement=0), 1),
                                          ("few_iterations", dict(ransac_plane_max_iterations=25), 2),
                                          ("road_triangle", dict(plane_estimator_use_triangle_maximation=1), 5)])
def test_ransac_plane_in_one_call(name, kw, seed, feature_kernel_path):
    P = capi.params_c0().replace(**kw)
    cloud = synth.make_cloud(synth.HDL64, seed=6, frame=4)
    uv = synth.make_features(2000, seed=6)
    est = make_estimator(P)
    gp = RansacPlane(seed=seed)
    depth, types = est.CalculateDepth(cloud, uv, gp)
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    c0, inl0 = ref.estimate_ground_plane(seed)
    d0, t0 = ref.calculate_depth(uv)
    assert gp.isSegmented() and np.array_equal(gp.getModelCoeffs(), c0)
    assert gp.n_inliers == inl0.size and gp.inliers is None  # the list is not part of the frame call ...
    assert np.array_equal(gp.getInlinersIndex(), inl0)        # ... it is fetched when asked for
    assert_depth_parity(depth, types, d0, t0)
    # the plane stays installed on the slot: the feature-only overload uses it
    uv2 = synth.make_features(700, seed=61)
    d2, t2 = est.CalculateDepth(uv2)
    assert_depth_parity(d2, t2, *ref.calculate_depth(uv2))
    # and the plane object, now segmented, takes the supplied-plane call on the next frame
    cloud2 = synth.make_cloud(synth.HDL64, seed=6, frame=5)
    d3, t3 = est.CalculateDepth(cloud2, uv, gp)
    ref.set_cloud(cloud2)
    ref.set_ground_plane(c0, inl0)
    assert_depth_parity(d3, t3, *ref.calculate_depth(uv))


def test_null_plane_is_estimated_like_the_reference_does():
    """CalculateDepth(cloud, uv, nullptr): the reference creates a RansacPlane and estimates it (:275-278)."""
    P = capi.params_c0()
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=8, frame=2)
    uv = synth.make_features(1200, seed=8)
    est = make_estimator(P)
    depth, types = est.CalculateDepth(cloud, uv, None)
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    _, inl0 = ref.estimate_ground_plane(0)
    assert np.array_equal(est.getGroundPlaneInliers(), inl0)
    assert_depth_parity(depth, types, *ref.calculate_depth(uv))


@pytest.mark.parametrize("seed,scanner", [(10, synth.HDL64_KITTI), (12, synth.HDL64), (13, synth.VLP16)])
def test_semantic_plane_in_one_call(seed, scanner, feature_kernel_path):
    P = capi.params_c0()
    cloud = synth.make_cloud(scanner, seed=seed, frame=1)
    img = synth.make_label_image(cloud)
    uv = synth.make_features(1500, seed=seed)
    thr = P.ransac_plane_refinement_treshold  # tracklet_depth_module.cpp:281-282
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    c0, inl0 = ref.estimate_semantic_plane(img, LABELS, thr)
    d0, t0 = ref.calculate_depth(uv)
    est = make_estimator(P)
    gp = SemanticPlane(img, LABELS, thr)
    d, t = est.CalculateDepth(cloud, uv, gp)
    assert gp.isSegmented() and np.array_equal(gp.getModelCoeffs(), c0)
    assert gp.n_inliers == inl0.size and np.array_equal(gp.getInlinersIndex(), inl0)
    assert_depth_parity(d, t, d0, t0)


def test_semantic_plane_strided_label_image():
    P = capi.params_c0()
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=14, frame=1)
    img = synth.make_label_image(cloud)
    wide = np.zeros((img.shape[0], img.shape[1] + 38), dtype=np.uint8)
    wide[:, :img.shape[1]] = img
    uv = synth.make_features(800, seed=14)
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    c0, inl0 = ref.estimate_semantic_plane(img, LABELS, 0.1)
    est = make_estimator(P)
    lib, ctx = est._lib, est._ctx
    req = capi.MldPlaneRequest()
    lab = np.array(LABELS, dtype=np.int32)
    req.kind = capi.MLD_PLANE_SEMANTIC
    req.label_image, req.rows, req.cols, req.row_stride_bytes = wide.ctypes.data, img.shape[0], img.shape[1], wide.strides[0]
    req.ground_labels, req.n_labels, req.inlier_threshold = lab.ctypes.data, lab.size, 0.1
    res = capi.MldPlaneResult()
    d = np.empty(800)
    t = np.empty(800, dtype=np.int32)
    uvh = np.ascontiguousarray(uv, dtype=np.float64)
    est._check(lib.mld_calculate_depth_frame_estimate(ctx, 0, cloud.ctypes.data, cloud.shape[0], 16, C.byref(req),
                                                      uvh.ctypes.data, 800, d.ctypes.data, t.ctypes.data, C.byref(res)))
    assert res.status == 0 and res.n_inliers == inl0.size and np.array_equal(np.array(list(res.coeffs), np.float32), c0)
    assert_depth_parity(d, t, *ref.calculate_depth(uv))


def test_failed_estimation_raises_and_leaves_the_outputs_alone():
    """GroundPlane::ExceptionPclInvalid (RansacPlane.cpp:44-50, :224-227): the reference throws out of setInputCloud,
    no depth is computed (tracklet_depth_module.cpp:321,338 catch it and fill -1 themselves)."""
    est = make_estimator(capi.params_c0())
    uv = synth.make_features(100, seed=1)
    two = np.array([[1, 0, -1.7, 0], [2, 0, -1.7, 0]], np.float32)
    with pytest.raises(ExceptionPclInvalid):
        est.CalculateDepth(two, uv, RansacPlane())
    nan_cloud = np.full((500, 4), np.nan, np.float32)
    gp = RansacPlane(seed=3)
    with pytest.raises(ExceptionPclInvalid):
        est.CalculateDepth(nan_cloud, uv, gp)
    assert not gp.isSegmented()
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=9, frame=1)
    img = synth.make_label_image(cloud)
    with pytest.raises(ExceptionPclInvalid):
        est.CalculateDepth(cloud, uv, SemanticPlane(np.zeros_like(img), LABELS, 0.1))
    # C level: outputs untouched, status reported
    lib, ctx = est._lib, est._ctx
    req = capi.MldPlaneRequest()
    req.kind, req.seed = capi.MLD_PLANE_RANSAC, 3
    res = capi.MldPlaneResult()
    d = np.full(100, 123.0)
    t = np.full(100, -5, dtype=np.int32)
    uvh = np.ascontiguousarray(uv, dtype=np.float64)
    rc = lib.mld_calculate_depth_frame_estimate(ctx, 0, nan_cloud.ctypes.data, 500, 16, C.byref(req), uvh.ctypes.data, 100,
                                                d.ctypes.data, t.ctypes.data, C.byref(res))
    assert rc == capi.MLD_ERR_CLOUD_TOO_SMALL and res.status == 1 and (d == 123.0).all() and (t == -5).all()
    # the estimator is still usable
    gp2 = RansacPlane(seed=1)
    dd, tt = est.CalculateDepth(cloud, uv, gp2)
    ref = make_oracle(capi.params_c0())
    ref.set_cloud(cloud)
    ref.estimate_ground_plane(1)
    assert_depth_parity(dd, tt, *ref.calculate_depth(uv))


def test_plane_request_is_ignored_without_the_road_fallback():
    P = capi.params_c0().replace(do_use_ransac_plane=0)
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=15, frame=0)
    uv = synth.make_features(900, seed=15)
    est = make_estimator(P)
    lib, ctx = est._lib, est._ctx
    req = capi.MldPlaneRequest()
    req.kind, req.seed = capi.MLD_PLANE_RANSAC, 5
    d = np.empty(900)
    t = np.empty(900, dtype=np.int32)
    uvh = np.ascontiguousarray(uv, dtype=np.float64)
    est._check(lib.mld_calculate_depth_frame_estimate(ctx, 0, cloud.ctypes.data, cloud.shape[0], 16, C.byref(req),
                                                      uvh.ctypes.data, 900, d.ctypes.data, t.ctypes.data, None))
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    ref.set_ground_plane(None, None)
    assert_depth_parity(d, t, *ref.calculate_depth(uv))
    # bad requests
    req.kind = 7
    assert lib.mld_calculate_depth_frame_estimate(ctx, 0, cloud.ctypes.data, cloud.shape[0], 16, C.byref(req), uvh.ctypes.data,
                                                  900, d.ctypes.data, t.ctypes.data, None) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_calculate_depth_frame_estimate(ctx, 0, cloud.ctypes.data, cloud.shape[0], 16, None, uvh.ctypes.data, 900,
                                                  d.ctypes.data, t.ctypes.data, None) == capi.MLD_ERR_INVALID_ARG
    req.kind = capi.MLD_PLANE_SEMANTIC  # no image
    est2 = make_estimator(capi.params_c0())
    assert est2._lib.mld_calculate_depth_frame_estimate(est2._ctx, 0, cloud.ctypes.data, cloud.shape[0], 16, C.byref(req),
                                                        uvh.ctypes.data, 900, d.ctypes.data, t.ctypes.data,
                                                        None) == capi.MLD_ERR_INVALID_ARG


def test_frames_in_turn_on_two_slots_and_the_timing_breakdown():
    """A sequence as tracklets_depth sees it: a new cloud and a fresh plane every frame, slots in turn (the previous
    frame stays resident); the phase clocks of mld_frame_timing add up."""
    P = capi.params_c0()
    est = make_estimator(P, max_frames=2)
    est.timingEnable(True)
    ref = make_oracle(P)
    ransac_plane_us = []
    for it in range(6):
        cloud = synth.make_cloud(synth.HDL64_KITTI, seed=20, frame=it)
        uv = synth.make_features(1000, seed=200 + it)
        gp = RansacPlane(seed=it)
        d, t = est.CalculateDepth(cloud, uv, gp, slot=it % 2)
        ref.set_cloud(cloud)
        ref.estimate_ground_plane(it)
        assert_depth_parity(d, t, *ref.calculate_depth(uv))
        tm = est.frameTiming()
        assert tm["h2d_us"] > 0 and tm["plane_us"] > 0 and tm["kernels_us"] > 0 and tm["d2h_us"] > 0
        assert tm["gpu_us"] >= tm["h2d_us"] + tm["plane_us"] + tm["kernels_us"] + tm["d2h_us"] - 1.0
        assert tm["total_us"] >= tm["api_us"] + tm["wait_us"] - 1.0
        ransac_plane_us.append(tm["plane_us"])
    est.timingEnable(False)
    # a supplied plane through the same machinery: no plane phase
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=20, frame=9)
    coeffs, inl = synth.make_ground_plane(cloud)
    est.timingEnable(True)
    est.CalculateDepth(cloud, synth.make_features(500, seed=3), GroundPlane(coeffs, inl))
    # (only the wait for the side stream's small inputs: ~10-20 us; the estimation itself takes 45-60)
    assert est.frameTiming()["plane_us"] < max(30.0, 0.8 * min(ransac_plane_us)), (est.frameTiming(), ransac_plane_us)


def test_large_cloud_with_the_pass_through_takes_the_per_slot_estimator():
    """Beyond ~0.4 M points the z pass-through no longer fits the one-block kernel's LDS: the call falls back to the
    per-slot estimator (same results)."""
    P = capi.params_c0().replace(ransac_plane_min_z=-3.0, ransac_plane_max_z=-0.5)
    cloud = synth.make_cloud(synth.DENSE128, seed=5, frame=0)
    assert cloud.shape[0] > 450_000
    uv = synth.make_features(600, seed=5)
    est = make_estimator(P)
    gp = RansacPlane(seed=4)
    d, t = est.CalculateDepth(cloud, uv, gp)
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    c0, inl0 = ref.estimate_ground_plane(4)
    assert np.array_equal(gp.getModelCoeffs(), c0) and np.array_equal(gp.getInlinersIndex(), inl0)
    assert_depth_parity(d, t, *ref.calculate_depth(uv))


@pytest.mark.parametrize("switch", ["MLD_FRAME_HELPER=0", "MLD_FRAME_COPY=1"])
def test_one_frame_call_variants_of_the_test_build(switch, monkeypatch):
    """The A/B switches of the measurement build (no helper thread: the caller queues the side-stream work itself;
    results through device memory and a D2H copy instead of direct stores into the pinned block) give the same results."""
    name, value = switch.split("=")
    monkeypatch.setenv(name, value)
    monkeypatch.setattr(capi, "_lib", capi.load_ab())
    P = capi.params_c0()
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=17, frame=3)
    uv = synth.make_features(1500, seed=17)
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    est = make_estimator(P)
    # supplied plane
    plane = synth.make_ground_plane(cloud)
    ref.set_ground_plane(*plane)
    d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    assert_depth_parity(d, t, *ref.calculate_depth(uv))
    # plane estimated inside the call, both estimators
    gp = RansacPlane(seed=9)
    d, t = est.CalculateDepth(cloud, uv, gp)
    c0, inl0 = ref.estimate_ground_plane(9)
    assert np.array_equal(gp.getModelCoeffs(), c0) and np.array_equal(gp.getInlinersIndex(), inl0)
    assert_depth_parity(d, t, *ref.calculate_depth(uv))
    img = synth.make_label_image(cloud)
    sp = SemanticPlane(img, LABELS, 0.1)
    d, t = est.CalculateDepth(cloud, uv, sp)
    c1, inl1 = ref.estimate_semantic_plane(img, LABELS, 0.1)
    assert np.array_equal(sp.getModelCoeffs(), c1) and sp.n_inliers == inl1.size
    assert_depth_parity(d, t, *ref.calculate_depth(uv))

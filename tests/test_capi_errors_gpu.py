"""Return codes of the C-ABI for bad arguments (include/mld.h): nothing crashes, every error carries a message."""
import ctypes as C

import numpy as np
import pytest

from mono_lidar_depth_amd import capi, synth

from helpers import make_estimator

pytestmark = pytest.mark.gpu


def test_bad_arguments_return_error_codes():
    lib = capi.load()
    est = make_estimator(capi.params_c0())
    ctx = est._ctx
    cloud = synth.make_cloud(synth.VLP16, seed=2)
    uv = synth.make_features(50, seed=2)
    depth = np.empty(50)
    types = np.empty(50, dtype=np.int32)
    n = C.c_int64(0)

    def err():
        return lib.mld_last_error(ctx).decode()

    # slot range
    assert lib.mld_set_cloud(ctx, 5, cloud.ctypes.data, cloud.shape[0], 16) == capi.MLD_ERR_INVALID_ARG and "slot" in err()
    assert lib.mld_set_cloud(ctx, -1, cloud.ctypes.data, cloud.shape[0], 16) == capi.MLD_ERR_INVALID_ARG
    # stride, null pointer, negative count
    assert lib.mld_set_cloud(ctx, 0, cloud.ctypes.data, cloud.shape[0], 12) == capi.MLD_ERR_INVALID_ARG and "stride" in err()
    assert lib.mld_set_cloud(ctx, 0, None, 10, 16) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_set_cloud(ctx, 0, cloud.ctypes.data, -4, 16) != capi.MLD_OK
    # CalculateDepth before a cloud
    assert lib.mld_calculate_depth(ctx, 0, uv.ctypes.data, 50, depth.ctypes.data, types.ctypes.data) == capi.MLD_ERR_NOT_INITIALIZED
    assert lib.mld_set_cloud(ctx, 0, cloud.ctypes.data, cloud.shape[0], 16) == capi.MLD_OK
    # plane missing while do_use_ransac_plane is on
    assert lib.mld_calculate_depth(ctx, 0, uv.ctypes.data, 50, depth.ctypes.data, types.ctypes.data) == capi.MLD_ERR_NO_GROUND_PLANE
    assert lib.mld_get_ground_plane_inliers(ctx, 0, None, 0, C.byref(n)) == capi.MLD_ERR_NOT_INITIALIZED
    coeffs = (C.c_float * 4)(0, 0, 1, 1.73)
    inl = np.arange(100, dtype=np.int32)
    assert lib.mld_set_ground_plane(ctx, 0, coeffs, inl.ctypes.data, inl.size) == capi.MLD_OK
    # null outputs, negative F, F = 0
    assert lib.mld_calculate_depth(ctx, 0, None, 50, depth.ctypes.data, types.ctypes.data) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_calculate_depth(ctx, 0, uv.ctypes.data, 50, None, types.ctypes.data) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_calculate_depth(ctx, 0, uv.ctypes.data, -1, depth.ctypes.data, types.ctypes.data) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_calculate_depth(ctx, 0, uv.ctypes.data, 0, depth.ctypes.data, types.ctypes.data) == capi.MLD_OK
    assert lib.mld_calculate_depth(ctx, 0, uv.ctypes.data, 50, depth.ctypes.data, None) == capi.MLD_OK  # types optional
    # getter capacities
    assert lib.mld_get_visible_count(ctx, 0, C.byref(n)) == capi.MLD_OK and n.value > 0
    small = np.empty(4)
    assert lib.mld_get_visible_image_points(ctx, 0, small.ctypes.data, 2) == capi.MLD_ERR_CAPACITY
    assert lib.mld_get_cloud_camera_cs(ctx, 0, small.ctypes.data, 1) == capi.MLD_ERR_CAPACITY
    d = C.c_double(0)
    assert lib.mld_get_point_depth_cam_visible(ctx, 0, n.value, C.byref(d)) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_get_point_depth_cam_visible(ctx, 0, 0, C.byref(d)) == capi.MLD_OK and np.isfinite(d.value)
    cnt = C.c_int64(0)
    assert lib.mld_get_ground_plane_inliers(ctx, 0, None, 0, C.byref(cnt)) == capi.MLD_OK and cnt.value == 100
    two = np.empty(2, dtype=np.int32)
    assert lib.mld_get_ground_plane_inliers(ctx, 0, two.ctypes.data, 2, C.byref(cnt)) == capi.MLD_ERR_CAPACITY
    # semantic plane arguments
    img = np.zeros((10, 10), dtype=np.uint8)
    lab = np.array([7], dtype=np.int32)
    out4 = (C.c_float * 4)()
    assert lib.mld_estimate_semantic_plane(ctx, 0, None, 10, 10, 10, lab.ctypes.data, 1, 0.1, out4, C.byref(cnt)) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_estimate_semantic_plane(ctx, 0, img.ctypes.data, 10, 10, 5, lab.ctypes.data, 1, 0.1, out4, C.byref(cnt)) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_estimate_semantic_plane(ctx, 0, img.ctypes.data, 10, 10, 10, lab.ctypes.data, 1, 0.1, out4,
                                           C.byref(cnt)) == capi.MLD_ERR_CLOUD_TOO_SMALL  # no labelled point
    # a context survives all of the above
    assert lib.mld_set_ground_plane(ctx, 0, coeffs, inl.ctypes.data, inl.size) == capi.MLD_OK
    assert lib.mld_calculate_depth(ctx, 0, uv.ctypes.data, 50, depth.ctypes.data, types.ctypes.data) == capi.MLD_OK
    # null context
    assert lib.mld_calculate_depth(None, 0, uv.ctypes.data, 50, depth.ctypes.data, types.ctypes.data) == capi.MLD_ERR_INVALID_ARG
    assert lib.mld_synchronize(None) != capi.MLD_OK


def test_create_rejects_bad_camera_and_sizes():
    lib = capi.load()
    P = capi.params_c0()
    T = (C.c_double * 12)(*synth.T_CAM_LIDAR.ravel())
    status = C.c_int(0)
    for cam in (capi.MldCamera(721.0, 600.0, 170.0, 0, 375), capi.MldCamera(721.0, 600.0, 170.0, 1242, -1),
                capi.MldCamera(0.0, 600.0, 170.0, 1242, 375)):
        ctx = lib.mld_create(C.byref(P), C.byref(cam), T, 0, 1, 0, 0, C.byref(status))
        assert not ctx and status.value == capi.MLD_ERR_INVALID_ARG, (cam.width, cam.height, cam.focal_length)
    cam = capi.MldCamera(721.0, 600.0, 170.0, 1242, 375)
    ctx = lib.mld_create(C.byref(P), C.byref(cam), T, 0, 0, 0, 0, C.byref(status))
    assert not ctx and status.value == capi.MLD_ERR_INVALID_ARG
    ctx = lib.mld_create(C.byref(P), C.byref(cam), T, 99, 1, 0, 0, C.byref(status))
    assert not ctx and status.value != capi.MLD_OK
    assert lib.mld_create_error()

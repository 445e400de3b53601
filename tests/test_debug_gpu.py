"""Debug mode (ActivateDebugMode, DepthEstimator.h:85-87): the vectors behind getCloudTriangleCorners /
getCloudRansacPlane / getCloudInterpolated, checked against the oracle's per-feature trace."""
import numpy as np
import pytest

from mono_lidar_depth_amd import NO_PLANE, GroundPlane, capi, synth

from helpers import make_estimator, make_oracle

pytestmark = pytest.mark.gpu


def _frame(seed, n_feat=600):
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=seed, frame=2)
    uv = synth.make_features(n_feat, seed=seed)
    plane = synth.make_ground_plane(cloud)
    return cloud, uv, plane


@pytest.mark.parametrize("variant", ["c0", "no_histogram", "no_trimax"])
def test_debug_mode_matches_normal_mode_and_oracle_corners(variant):
    P = capi.params_c0()
    if variant == "no_histogram":
        P = P.replace(do_use_histogram_segmentation=0)
    if variant == "no_trimax":
        P = P.replace(do_use_triangle_size_maximation=0)
    cloud, uv, plane = _frame(31)
    est = make_estimator(P)
    d0, t0 = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    est.ActivateDebugMode()
    d1, t1 = est.CalculateDepth(uv)
    assert np.array_equal(t0, t1)
    assert np.array_equal(d0[t0 != 16], d1[t0 != 16])
    assert np.abs(d0 - d1).max() <= 1e-9
    corners = est.getTriangleCorners()
    assert corners.shape == (uv.shape[0], 9)

    ref = make_oracle(P)
    ref.set_cloud(cloud)
    ref.set_ground_plane(*plane)
    cam = ref.cloud_camera_cs().T  # N x 3
    pidx = ref.point_index()
    n_with = 0
    for i, (u, v) in enumerate(uv):
        tr = ref.trace_feature(u, v)
        has = variant != "no_trimax" and tr["corner_pos"][0] >= 0
        if not has:
            # first-three-points mode never publishes corners (DepthEstimator.cpp:919-925)
            assert np.isnan(corners[i]).all(), i
            continue
        n_with += 1
        want = []
        for cpos in tr["corner_pos"]:
            vis = tr["nb_idx"][tr["seg_pos"][cpos]]
            want.extend(cam[pidx[vis]])
        assert np.array_equal(corners[i], np.asarray(want)), i
    if variant != "no_trimax":
        assert n_with > 20
        cloud3 = est.getCloudTriangleCorners()
        assert cloud3.shape == (3, 3 * n_with)


def test_interpolated_points_lie_on_the_viewing_rays():
    P = capi.params_c0()
    cloud, uv, plane = _frame(32)
    est = make_estimator(P)
    est.ActivateDebugMode()
    d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    pts = est.getCloudInterpolated()
    ok = d >= 0
    assert pts.shape == (3, int(ok.sum())) and ok.sum() > 50
    assert np.array_equal(pts[2], d[ok])
    # reprojection gives the feature back
    u = pts[0] / pts[2] * synth.KITTI_F + synth.KITTI_CU
    v = pts[1] / pts[2] * synth.KITTI_F + synth.KITTI_CV
    assert np.abs(u - uv[ok, 0]).max() < 1e-9 and np.abs(v - uv[ok, 1]).max() < 1e-9
    assert est.getCloudNeighbors().shape == (3, 0)


@pytest.mark.parametrize("camx", [None, 2.0])
def test_ground_plane_cloud(camx):
    P = capi.params_c0()
    if camx is not None:
        P = P.replace(ransac_plane_use_camx_treshold=1, ransac_plane_treshold_camx=camx)
    cloud, uv, plane = _frame(33)
    est = make_estimator(P)
    est.setInputCloud(cloud, GroundPlane(*plane))
    got = est.getCloudRansacPlane()
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    cam = ref.cloud_camera_cs().T
    inl = np.unique(plane[1])
    want = cam[inl]
    if camx is not None:
        want = want[np.abs(want[:, 0]) <= camx]
    assert got.shape == (3, want.shape[0]) and want.shape[0] > 100
    assert np.array_equal(got.T, want)


def test_ground_plane_cloud_without_plane_is_empty():
    P = capi.params_c0()
    cloud, uv, _ = _frame(34)
    est = make_estimator(P)
    est.setInputCloud(cloud, NO_PLANE)
    assert est.getCloudRansacPlane().shape == (3, 0)


def test_depth_calc_stats_follow_the_last_call():
    P = capi.params_c0()
    cloud, uv, plane = _frame(35)
    est = make_estimator(P)
    assert est.getDepthCalcStats() == {"PointCount": 0}
    d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    st = est.getDepthCalcStats()
    assert st["PointCount"] == uv.shape[0]
    assert st["Success"] == int((t == 1).sum()) and st["SuccessRoad"] == int((t == 16).sum())
    assert sum(v for k, v in st.items() if k != "PointCount") == uv.shape[0]
    d, t = est.CalculateDepth(uv[:37])
    assert est.getDepthCalcStats()["PointCount"] == 37

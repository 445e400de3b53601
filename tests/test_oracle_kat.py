"""Pins the CPU oracle against the reference's own tests for this path and against independent math.

Known-answer / property tests restated from monolidar_fusion/test/test_monolidar_fusion.cpp:
  Histogram.FilterPointsMinDistBlob (:306-374, exact KAT), Histogram.GetNearestPoint (:277-303),
  NeigborFinder.findByPixel (:82-171, property bounds).
Plus micro-vectors for the triangle search, planarity check, ray/plane intersection, thresholds, the M-estimator
plane (vs numpy SVD) and PCA (vs numpy eigh).
"""
import ctypes as C

import numpy as np
import pytest

from mono_lidar_depth_amd import capi, synth
from oracle import oracle


def test_histogram_filter_points_min_dist_blob_kat():
    # test_monolidar_fusion.cpp:310-373
    depths = [2.2, 3.5, 4.2, 5.2, 5.2, 6.2, 7.2, 8.2, 8.3, 8.4, 9.2, 10.2, 10.5]
    ok, keep, lo, hi = oracle.filter_points_min_dist_blob(depths, 1.0, 3)
    assert ok
    assert [depths[k] for k in keep] == [8.2, 8.3, 8.4]
    assert lo == 8.0 and hi == 9.0
    assert all(lo <= depths[k] <= hi for k in keep)


def test_histogram_get_nearest_point():
    # test_monolidar_fusion.cpp:277-303: depths 5.0, 5.5, ... -> the first point
    depths = [5 + 0.5 * i for i in range(10)]
    assert oracle.get_nearest_point(depths, list(range(10))) == 0
    assert oracle.get_nearest_point([], []) == -1


def test_histogram_edge_semantics():
    f = oracle.filter_points_min_dist_blob
    # a single point in the first non-empty bin followed by an empty bin: rejected before any maximum (:84-85)
    assert not f([1.1, 3.1, 3.2, 3.3], 1.0, 3)[0]
    # maximum found, then an empty bin: break first (:79-81), success
    ok, keep, lo, hi = f([3.1, 3.2, 3.3, 5.5], 1.0, 3)
    assert ok and list(keep) == [0, 1, 2] and (lo, hi) == (3.0, 4.0)
    # rising counts: the LAST bin of a rising run wins
    ok, keep, lo, hi = f([2.1, 2.2, 2.3, 3.1, 3.2, 3.3, 3.4, 4.5], 1.0, 3)
    assert ok and (lo, hi) == (3.0, 4.0) and list(keep) == [3, 4, 5, 6]
    # equal counts do not replace the maximum and do not break
    ok, keep, lo, hi = f([2.1, 2.2, 2.3, 3.1, 3.2, 3.3, 4.5], 1.0, 3)
    assert ok and (lo, hi) == (2.0, 3.0)
    # fewer than min count everywhere -> no maximum; all in one bin with maxDist/binW+1 <= 1 -> false
    assert not f([2.1, 2.2], 1.0, 3)[0]
    assert not f([], 0.3, 3)[0]
    # depth capped at 999 lands in the last bin; bin width 0.3 borders
    ok, keep, lo, hi = f([9.95, 10.0, 10.1], 0.3, 3)
    assert ok and lo == 33 * 0.3 and hi == 33 * 0.3 + 0.3 and list(keep) == [0, 1, 2]
    # membership is by [lo, hi), not by bin index: 10.2 / 0.3 rounds into bin 33 or 34 but 10.2 >= hi is dropped
    ok, keep, lo, hi = f([9.95, 10.0, 10.1, 10.2, 10.3], 0.3, 3)
    assert ok and all(lo <= [9.95, 10.0, 10.1, 10.2, 10.3][k] < hi for k in keep)
    # min count 0: leading empty bins are no-ops, first non-empty bin run decides
    ok, keep, lo, hi = f([5.5], 1.0, 0)
    assert ok and (lo, hi) == (5.0, 6.0)


def test_neighbor_finder_find_by_pixel_property():
    """test_monolidar_fusion.cpp:82-171 with a seeded generator instead of std::rand."""
    W = H = 100
    cam = capi.MldCamera(600.0, 50.0, 50.0, W, H)
    P = capi.params_c0().replace(pixelarea_search_witdh=3, pixelarea_search_height=5, do_use_ransac_plane=0)
    T = np.hstack([np.eye(3), np.zeros((3, 1))])
    ref = oracle.OracleDepthEstimator(P, cam, T)
    rng = np.random.default_rng(0)
    uv = rng.integers(0, 10, size=(50, 2)).astype(np.float64)
    # the reference places points at integer pixels; use pixel centres + 0.25 so that truncation is unambiguous
    uvp = uv + 0.25
    depth = rng.integers(1, 11, size=50).astype(np.float64)
    rays = np.array([ref.viewing_ray(a, b) for a, b in uvp])
    pts = rays * depth[:, None]
    cloud = np.zeros((50, 4), np.float32)
    cloud[:, :3] = pts
    ref.set_cloud(cloud)
    img = ref.cloud_image_cs().T
    assert np.abs(img - uvp).max() < 0.01  # re-projection error (:166)
    vis = ref.point_index()
    for a, b in uvp:
        tr = ref.trace_feature(a, b)
        for k in tr["nb_idx"]:
            q = img[vis[k]]
            assert abs(q[0] - a) <= np.ceil(3 * 0.5) + 0.01  # :167
            assert abs(q[1] - b) <= np.ceil(5 * 0.5) + 0.01  # :168
    # first point wins per pixel (NeighborFinderPixel.cpp:51-54)
    pm = ref.pixel_map()
    seen = {}
    for i, raw in enumerate(vis):
        key = (int(img[raw][0]), int(img[raw][1]))
        seen.setdefault(key, i)
    for (x, y), i in seen.items():
        assert pm[y, x] == i


def test_max_spanning_triangle_vectors():
    t = oracle.max_spanning_triangle
    assert not t([[0, 0, 0], [1, 0, 0]])[0]                      # < 3 points
    assert not t([[1, 1, 1]] * 4)[0]                             # all on one spot (:65)
    # last point (3) is never the third corner (:71) although its distance sum (178) beats point 2 (52)
    ok, c = t([[0, 0, 0], [10, 0, 0], [5, 1, 0], [5, 8, 0]])
    assert ok and list(c) == [0, 1, 2]
    # tie d01 == d02 == 100: strict '>' keeps the FIRST maximal pair in (i,j) order
    ok, c = t([[0, 0, 0], [6, 8, 0], [8, 6, 0], [1, 1, 0]])
    assert ok and list(c) == [0, 1, 2]
    # exactly three points with farthest pair (0,1): the only candidates k < n-1 are the pair itself -> false
    assert not t([[0, 0, 0], [10, 0, 0], [5, 1, 0]])[0]
    # ... but with farthest pair (0,2) the third corner 1 is found
    ok, c = t([[0, 0, 0], [5, 1, 0], [10, 0, 0]])
    assert ok and list(c) == [0, 2, 1]
    # third corner coinciding with a chosen corner is skipped (d <= 0)
    ok, c = t([[0, 0, 0], [4, 0, 0], [0, 0, 0], [4, 0, 0]])
    assert not ok


def test_check_planar_and_intersection_vectors():
    assert oracle.check_planar([0, 0, 5], [1, 0, 5], [0, 1, 5], 0.1)
    assert not oracle.check_planar([0, 0, 5], [1, 0, 5], [2, 0.01, 5], 0.1)  # nearly collinear
    # fronto-parallel plane z = 5, ray through the principal axis
    ok, pt, depth = oracle.intersect_triangle([0, 0, 5], [1, 0, 5], [0, 1, 5], [0, 0, 0], [0, 0, 1], 0.03)
    assert ok and depth == 5.0 and np.allclose(pt, [0, 0, 5])
    # grazing plane: |n . ray| < threshold -> rejected only by the OrthogonalTreshold variant
    p1, p2, p3 = [0, 1, 1], [0, 1, 9], [1, 1.0001, 5]
    assert not oracle.intersect_triangle(p1, p2, p3, [0, 0, 0], [0, 0, 1], 0.03)[0]
    assert oracle.intersect_triangle(p1, p2, p3, [0, 0, 0], [0, 0, 1], 0.0)[0]
    # swapped line arguments (road estimators): same geometric line, same point
    ray = np.array([0.1, 0.2, 0.97])
    ray /= np.linalg.norm(ray)
    a = oracle.intersect_triangle([0, 2, 4], [3, 2, 9], [-3, 2.2, 7], [0, 0, 0], ray, 0.0)
    b = oracle.intersect_triangle([0, 2, 4], [3, 2, 9], [-3, 2.2, 7], ray, [0, 0, 0], 0.0)
    assert np.allclose(a[1], b[1], atol=1e-12) and abs(a[2] - b[2]) < 1e-12


def test_threshold_vectors():
    P = capi.params_c0()
    assert oracle.threshold_global(P, -0.5) == (1, -1.0)
    assert oracle.threshold_global(P, 100.5) == (2, -1.0)
    assert oracle.threshold_global(P, 42.0) == (0, 42.0)
    Pa = P.replace(treshold_depth_mode=1)
    assert oracle.threshold_global(Pa, -0.5) == (0, 0.0) and oracle.threshold_global(Pa, 150.0) == (0, 100.0)
    pts = [[0, 0, 10.0], [0, 0, 12.0], [0, 0, 11.0]]
    # relative 0.5: [9, 13]
    assert oracle.threshold_local(P, pts, 8.9) == (1, -1.0)
    assert oracle.threshold_local(P, pts, 13.1) == (2, -1.0)
    assert oracle.threshold_local(P, pts, 9.0) == (0, 9.0)
    Pabs = P.replace(treshold_depth_local_valuetype=0, treshold_depth_local_value=0.25)
    assert oracle.threshold_local(Pabs, pts, 9.7) == (1, -1.0)
    assert oracle.threshold_local(Pabs, pts, 12.25) == (0, 12.25)
    Padj = Pabs.replace(treshold_depth_local_mode=1)
    assert oracle.threshold_local(Padj, pts, 9.0) == (0, 9.75) and oracle.threshold_local(Padj, pts, 20.0) == (0, 12.25)


@pytest.mark.parametrize("seed", range(5))
def test_mestimator_plane_matches_numpy_svd(seed):
    rng = np.random.default_rng(seed)
    k = int(rng.integers(3, 40))
    # road-like patch in camera coordinates: y ~ 1.65 + noise, x/z spread
    pts = np.stack([rng.uniform(-3, 3, k), 1.65 + rng.normal(0, 0.02, k), rng.uniform(5, 25, k)], axis=1)
    prior_n, prior_off = np.array([0.0, 0.0, 1.0]), 1.73  # lidar-frame prior applied to camera points (quirk)
    n, off = oracle.mestimator_plane(pts, prior_n, prior_off)
    w = 1.0 / np.abs(pts @ prior_n + prior_off)
    c = (pts * w[:, None]).sum(0) / w.sum()
    M = ((pts - c) * np.sqrt(w)[:, None]).T
    U = np.linalg.svd(M, full_matrices=False)[0]
    n_ref = U[:, -1]
    assert abs(abs(float(n @ n_ref)) - 1.0) < 1e-12
    assert abs(abs(off) - abs(float(n_ref @ c))) < 1e-10


def test_mestimator_point_on_prior_plane_gives_nan():
    # weight 1/0 = inf -> NaN centre (PlaneEstimationMEstimator.cpp:32); IEEE behaviour kept
    pts = [[0, 0, -1.73], [1, 0, 5], [0, 1, 6], [1, 1, 7]]
    n, off = oracle.mestimator_plane(pts, [0.0, 0.0, 1.0], 1.73)
    assert np.isnan(off) and np.isnan(n).all()


@pytest.mark.parametrize("seed", range(4))
def test_pca_matches_numpy_eigh(seed):
    rng = np.random.default_rng(100 + seed)
    k = int(rng.integers(4, 30))
    pts = np.stack([rng.uniform(-2, 2, k), rng.uniform(-2, 2, k), 10 + rng.normal(0, 0.01, k)], axis=1)
    P = capi.params_c0()
    r, n, m = oracle.pca(P, pts)
    X = pts.T
    C_ = (X - X.mean(1, keepdims=True)) @ (X - X.mean(1, keepdims=True)).T
    w, V = np.linalg.eigh(C_)
    assert abs(abs(float(n @ V[:, 0])) - 1) < 1e-9
    assert np.allclose(m, X.mean(1), atol=1e-12)
    planarity = np.float32((w[1] - w[0]) / w[2])
    expect = 3 if planarity < P.pca_treshold_2_1_rel_min else 2
    assert r == expect


def test_calibration_inverses():
    P = capi.params_c0()
    cam = capi.MldCamera(synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV, synth.KITTI_W, synth.KITTI_H)
    th = 0.03
    R = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], float) @ np.array(
        [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    T = np.hstack([R, [[0.01], [-0.08], [-0.27]]])
    ref = oracle.OracleDepthEstimator(P, cam, T)
    Tinv, Kinv = ref.calibration()
    A = np.eye(4)
    A[:3] = T
    assert np.allclose(Tinv, np.linalg.inv(A)[:3], atol=1e-15)
    K = np.array([[cam.focal_length, 0, cam.principal_point_x], [0, cam.focal_length, cam.principal_point_y], [0, 0, 1]])
    assert np.allclose(Kinv, np.linalg.inv(K), rtol=1e-15, atol=1e-18)

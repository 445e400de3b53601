"""Hand-derived micro-vectors for the third-party (Eigen) semantics of the path — SURVEY.md §9.

Every expected number below was worked out on paper from the cited REFERENCE lines and the published Eigen 3.3
definitions, not obtained from the restatement: small integers / exact fractions chosen so that the answer is known in
closed form.  They pin the oracle (CPU, always) and — through a tiny end-to-end scene — the HIP path (GPU mark).

  H1  Hyperplane::Through(p1,p2,p3) + ParametrizedLine::Through(n0,n1).intersectionPoint  (LinePlaneIntersectionBase.cpp:41,
      LinePlaneIntersectionNormal.cpp:17-26): oblique plane, un-normalised n1, negative parameter allowed
  H2  the same call with swapped arguments (n0 = viewing direction, n1 = support), as the road estimators make it
      (RoadDepthEstimatorMEstimator.cpp:55-60): origin = direction, line direction = -direction
  H3  orthogonality threshold (LinePlaneIntersectionOrthogonalTreshold.cpp:22-32): |n . ray| >= thr, with normalised ray
  H4  Hyperplane::Through on a degenerate (collinear) triangle: the normal comes from the SVD fallback; only its defining
      properties are determined (unit length, orthogonal to both edges)
  H5  PlaneEstimationMEstimator::EstimatePlane (:18-55): weights 1/absDistance to the prior, weighted centre, least
      singular direction — a configuration whose weighted scatter is diagonal (2, 1, 2/3)
  H6  the same with a point exactly ON the prior: weight = 1/0 = inf, centre = inf/inf = NaN — reproduced, not "fixed"
  H7  CameraPinhole::getImagePoints / getViewingRays (camera_pinhole.h:52-69,84-97): K p with all nine products and
      hnormalized; K^-1 (u,v,1) normalised
  H8  TresholdDepthLocal (TresholdDepthLocal.cpp:18-68): relative and absolute borders from the neighbours' z range
  E1  end to end, main path: five points on the plane z = 4 + 2y seen by an identity-calibrated camera; the depth of the
      feature at (50, 60) is 4 / (1 - 2*0.1) = 5
  E2  end to end, road path: points on y = 1.5 with the prior y = 1.4; depth of the feature at (50, 80) is 1.5/0.3 = 5
"""
import numpy as np
import pytest

from mono_lidar_depth_amd import CameraPinhole, GroundPlane, capi
from oracle import oracle

I34 = np.hstack([np.eye(3), np.zeros((3, 1))])


def _plane_z_4_plus_2y():
    # three points of z = 4 + 2y:  (0,0,4), (1,0,4), (0,1,6)
    return [0.0, 0.0, 4.0], [1.0, 0.0, 4.0], [0.0, 1.0, 6.0]


def test_H1_through_and_intersection_oblique_plane():
    p1, p2, p3 = _plane_z_4_plus_2y()
    # ray from the origin through (0,-1,1):  points (0,-s,s);  s = 4 - 2s  ->  s = 4/3
    ok, pt, depth = oracle.intersect_triangle(p1, p2, p3, [0, 0, 0], [0, -1, 1], 0.0)
    assert ok
    assert np.allclose(pt, [0.0, -4.0 / 3.0, 4.0 / 3.0], rtol=0, atol=1e-14) and abs(depth - 4.0 / 3.0) < 1e-14
    # n1 is NOT a unit vector in the reference either: Through() normalises (b - a), the point does not move
    ok, pt2, depth2 = oracle.intersect_triangle(p1, p2, p3, [0, 0, 0], [0, -7, 7], 0.0)
    assert ok and np.allclose(pt2, pt, rtol=0, atol=1e-14)
    # ray through (0,1,1): s = 4 + 2s -> s = -4: the intersection lies BEHIND the origin and is returned as is
    ok, pt3, depth3 = oracle.intersect_triangle(p1, p2, p3, [0, 0, 0], [0, 1, 1], 0.0)
    assert ok and np.allclose(pt3, [0.0, -4.0, -4.0], rtol=0, atol=1e-13) and abs(depth3 + 4.0) < 1e-13


def test_H2_swapped_arguments_give_the_same_point():
    p1, p2, p3 = _plane_z_4_plus_2y()
    d = [0.0, -0.6, 0.8]  # unit viewing direction; s*(−0.6, 0.8): 0.8 s = 4 − 1.2 s -> s = 2 -> (0, −1.2, 1.6)
    ok, pt, depth = oracle.intersect_triangle(p1, p2, p3, d, [0, 0, 0], 0.0)  # n0 = direction, n1 = support
    assert ok and np.allclose(pt, [0.0, -1.2, 1.6], rtol=0, atol=1e-14) and abs(depth - 1.6) < 1e-14


def test_H3_orthogonality_threshold_uses_the_normalised_ray():
    p1, p2, p3 = _plane_z_4_plus_2y()
    # plane normal = +-(0,-2,1)/sqrt5.  Ray (0,0,3): |n . (0,0,1)| = 1/sqrt5 = 0.4472...
    c = 1.0 / np.sqrt(5.0)
    assert oracle.intersect_triangle(p1, p2, p3, [0, 0, 0], [0, 0, 3], c - 1e-9)[0]
    assert not oracle.intersect_triangle(p1, p2, p3, [0, 0, 0], [0, 0, 3], c + 1e-9)[0]
    # ray inside the plane direction (0,1,2): n . ray = 0 -> rejected for any positive threshold
    assert not oracle.intersect_triangle(p1, p2, p3, [0, 0, 0], [0, 1, 2], 1e-12)[0]


def test_H4_degenerate_triangle_takes_the_svd_branch():
    # collinear corners: v0 x v1 = 0 <= |v0||v1| eps -> normal from the SVD of [v0; v1]; the null space is
    # two-dimensional, so only "unit, orthogonal to the line" is determined.  Checked through the intersection: a plane
    # containing the line x-axis-parallel through (0,0,5), hit by the ray through the line's own point (1,0,5), returns
    # that point whatever normal was picked.
    p1, p2, p3 = [0.0, 0.0, 5.0], [1.0, 0.0, 5.0], [2.0, 0.0, 5.0]
    ok, pt, depth = oracle.intersect_triangle(p1, p2, p3, [0, 0, 0], [1, 0, 5], 0.0)
    assert ok
    if np.all(np.isfinite(pt)):  # normal not orthogonal to the ray: the point is on the line
        assert np.allclose(pt, [1.0, 0.0, 5.0], rtol=0, atol=1e-12)


def test_H5_mestimator_diagonal_case():
    # prior z = 0 (unit normal (0,0,1), offset 0): absDistance = |z|.
    # points (+-1,0,1) weight 1, (0,+-1,2) weight 1/2.  weightsSum = 3, centre = (0,0,(1+1+1+1)/3) = (0,0,4/3).
    # weighted scatter: xx = 2, yy = 2*(1/2) = 1, zz = 2*(1/9) + 2*(1/2)*(4/9) = 2/3, off-diagonals 0
    # -> least singular direction = z axis, plane z = 4/3 (offset -4/3 for normal +z).
    pts = np.array([[1, 0, 1], [-1, 0, 1], [0, 1, 2], [0, -1, 2]], dtype=np.float64)
    n, off = oracle.mestimator_plane(pts, [0, 0, 1], 0.0)
    assert abs(abs(n[2]) - 1.0) < 1e-14 and abs(n[0]) < 1e-14 and abs(n[1]) < 1e-14
    assert abs(off * np.sign(n[2]) + 4.0 / 3.0) < 1e-14


def test_H6_mestimator_point_on_the_prior_gives_nan():
    # weights[0] = 1/|0| = inf; centre += inf * (0,1,0) has a 0*inf = NaN component; weightsSum = inf; centre /= inf
    # -> every component NaN or inf/inf = NaN (PlaneEstimationMEstimator.cpp:31-37): the plane is NaN
    pts = np.array([[0, 1, 0], [1, 0, 1], [-1, 0, 1], [0, -1, 2]], dtype=np.float64)
    n, off = oracle.mestimator_plane(pts, [0, 0, 1], 0.0)
    assert np.isnan(n).all() and np.isnan(off)


def test_H7_projection_and_viewing_ray():
    cam = capi.MldCamera(100.0, 50.0, 60.0, 200, 200)
    P = capi.params_c0().replace(do_use_ransac_plane=0)
    ref = oracle.OracleDepthEstimator(P, cam, I34)
    cloud = np.array([[1, 2, 4, 0], [-2, 1, 8, 0], [0, 0, 1, 0]], dtype=np.float32)
    ref.set_cloud(cloud)
    img = ref.cloud_image_cs().T
    # u = (f x + 0 y + cu z)/z, v = (0 x + f y + cv z)/z
    assert np.array_equal(img, [[75.0, 110.0], [25.0, 72.5], [50.0, 60.0]])
    # K^-1 = [[1/f,0,-cu/f],[0,1/f,-cv/f],[0,0,1]];  (125,60) -> (0.75,0,1)/1.25 = (0.6,0,0.8)
    assert np.allclose(ref.viewing_ray(125.0, 60.0), [0.6, 0.0, 0.8], rtol=0, atol=1e-15)
    assert np.allclose(ref.viewing_ray(50.0, 135.0), [0.0, 0.6, 0.8], rtol=0, atol=1e-15)


def test_H8_local_threshold_borders():
    pts = np.array([[0, 0, 4.0], [0, 0, 6.0], [0, 0, 5.0]])
    # relative, value 0.5: interval 2 -> [4 - 1, 6 + 1]; Dispose (mode 0)
    P = capi.params_c0().replace(treshold_depth_local_enabled=1, treshold_depth_local_mode=0,
                                 treshold_depth_local_valuetype=1, treshold_depth_local_value=0.5)
    assert oracle.threshold_local(P, pts, 3.0)[0] == 0 and oracle.threshold_local(P, pts, 7.0)[0] == 0  # InBounds
    assert oracle.threshold_local(P, pts, 2.999)[0] != 0 and oracle.threshold_local(P, pts, 7.001)[0] != 0
    # absolute, value 0.25, Adjust (mode 1): 6.5 -> clamped to 6.25, 3 -> 3.75
    P = P.replace(treshold_depth_local_mode=1, treshold_depth_local_valuetype=0, treshold_depth_local_value=0.25)
    assert oracle.threshold_local(P, pts, 6.5)[1] == 6.25
    assert oracle.threshold_local(P, pts, 3.0)[1] == 3.75
    assert oracle.threshold_local(P, pts, 5.0)[1] == 5.0


# ------------------------------------------------------------------------------------------------ end to end
CAM = (100, 100, 100.0, 50.0, 50.0)  # W, H, f, cu, cv


def _pixel_point(u, v, z):
    """3-D point that projects to (u, v) at depth z (identity calibration)."""
    return [(u - CAM[3]) / CAM[2] * z, (v - CAM[4]) / CAM[2] * z, z, 0.0]


def _scene_main():
    # points ON z = 4 + 2y, at pixel rows 58..62 / columns 48..52 around the feature (50.5, 60.5):
    # for pixel row v: b = (v - 50)/100, z = 4/(1 - 2b)
    pts = []
    for (u, v) in [(48.25, 58.25), (52.25, 58.75), (50.25, 62.25), (49.25, 60.25), (51.75, 61.25)]:
        b = (v - CAM[4]) / CAM[2]
        pts.append(_pixel_point(u, v, 4.0 / (1.0 - 2.0 * b)))
    cloud = np.array(pts, dtype=np.float64)
    uv = np.array([[50.0, 60.0]])
    return cloud, uv


def _expect_main(cloud32):
    """Closed form for E1 on the float32-rounded cloud: all five points lie (to float rounding) on z = 4 + 2y, so
    the plane through ANY three of them intersects the ray through (50,60) — direction (0, 0.1, 1) — at
    s = 4/(1 - 0.2) = 5."""
    return 5.0


def _params_e1():
    # window 6 x 9 around (50,60): columns 47..53, rows 55.5..64.5 -> all five points; z in [4.7, 5.3]: bin width 1.0
    # puts 4.76..4.99 into bin 4 and 5.0..5.3 into bin 5 -> use width 2.0: bin 2 = [4, 6) holds all five
    return capi.params_c0().replace(do_use_ransac_plane=0, histogram_segmentation_bin_witdh=2.0,
                                    histogram_segmentation_min_pointcount=3, treshold_depth_local_enabled=1,
                                    treshold_depth_local_valuetype=0, treshold_depth_local_value=1.0)


def test_E1_main_path_oracle():
    cloud, uv = _scene_main()
    c32 = cloud.astype(np.float32)
    cam = capi.MldCamera(CAM[2], CAM[3], CAM[4], CAM[0], CAM[1])
    ref = oracle.OracleDepthEstimator(_params_e1(), cam, I34)
    ref.set_cloud(c32)
    ref.set_ground_plane(None, None)
    d, t = ref.calculate_depth(uv)
    assert t[0] == 1, t  # Success
    assert abs(d[0] - _expect_main(c32)) < 2e-6  # float32 storage of the points: ~1e-7 relative


def _scene_road():
    # ground points ON y = 1.5 (camera frame, y down); prior plane y = 1.4 (coefficients (0,1,0,-1.4) in the lidar
    # frame = camera frame here).  Feature at (50, 80): ray (0, 0.3, 1) s, 0.3 s = 1.5 -> s = 5.
    # Narrow window (6 x 9) around (50,80) must hold fewer than 3 points (-> HistogramNoLocalMax -> road fallback);
    # the wide window (12 x 13.5: columns 44..56, rows 73.25..86.75) holds all of them.
    pts = []
    for (u, v) in [(44.5, 74.5), (55.5, 74.5), (44.5, 86.25), (55.5, 86.25), (50.5, 80.5)]:
        z = 1.5 / ((v - CAM[4]) / CAM[2])
        pts.append(_pixel_point(u, v, z))
    cloud = np.array(pts, dtype=np.float64)
    uv = np.array([[50.0, 80.0]])
    return cloud, uv


def _params_e2():
    # |y - 1.4| = 0.1 <= ransac_plane_point_distance_treshold (0.2): no point is "far"; local threshold off the
    # z-range of the inliers (z from 1.5/0.3625 = 4.14 to 1.5/0.245 = 6.12): depth 5 in bounds
    return capi.params_c0().replace(do_use_ransac_plane=1, treshold_depth_local_enabled=1,
                                    treshold_depth_local_valuetype=0, treshold_depth_local_value=0.5)


def test_E2_road_path_oracle():
    cloud, uv = _scene_road()
    c32 = cloud.astype(np.float32)
    cam = capi.MldCamera(CAM[2], CAM[3], CAM[4], CAM[0], CAM[1])
    ref = oracle.OracleDepthEstimator(_params_e2(), cam, I34)
    ref.set_cloud(c32)
    ref.set_ground_plane(np.array([0, 1, 0, -1.4], np.float32), np.arange(5, dtype=np.int32))
    d, t = ref.calculate_depth(uv)
    assert t[0] == 16, t  # SuccessRoad
    assert abs(d[0] - 5.0) < 2e-6


@pytest.mark.gpu
@pytest.mark.usefixtures("feature_kernel_path")
def test_E1_E2_hip_path():
    from mono_lidar_depth_amd import DepthEstimator
    cam = CameraPinhole(*CAM)
    for params, (cloud, uv), plane, want_type in (
            (_params_e1(), _scene_main(), None, 1),
            (_params_e2(), _scene_road(), GroundPlane(np.array([0, 1, 0, -1.4], np.float32), np.arange(5, dtype=np.int32)), 16)):
        est = DepthEstimator(device=0)
        est.InitConfig(params)
        est.Initialize(cam, I34)
        from mono_lidar_depth_amd.depth_estimator import NO_PLANE
        d, t = est.CalculateDepth(cloud.astype(np.float32), uv, plane if plane is not None else NO_PLANE)
        assert t[0] == want_type, t
        assert abs(d[0] - 5.0) < 2e-6
        est.close()

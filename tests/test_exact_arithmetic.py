"""The main path's geometry in EXACT rational arithmetic (fractions.Fraction), written from the mathematical definition and
not from the reference's operation order: which corners span the largest triangle, and where the viewing ray of the feature
meets the plane through them.

The C++ oracle, the NumPy restatement and the HIP kernels follow the reference's f64 operation order
(PlaneEstimationCalcMaxSpanningTriangle.cpp:37-100, LinePlaneIntersectionNormal.cpp:17-26, Eigen's Hyperplane::Through /
ParametrizedLine) and agree with one another bit for bit - three artefacts of one reading of the reference.  This file is the
check that does not share that reading's arithmetic: every float the oracle starts from (cloud coordinates, calibration,
feature pixel) is taken as the exact rational it is, the camera-frame points, squared distances, plane and intersection are
computed without any rounding, and the oracle's f64 results must be that exact answer up to f64 rounding:

  * corners: the first pair (row-major, strict >) of maximal exact squared distance and the first third point (last segmented
    point excluded, the reference's loop bound) of maximal exact distance sum - equal to the oracle's corner positions, or
    an exact-arithmetic near-tie (relative gap below 1e-12) where f64 rounding may legitimately order two candidates the
    other way;
  * depth: z of (ray through the pixel) x (plane through the three exact corners) - the oracle's depth within 1e-12
    relative (observed: below 1e-15; the conditioning of near-grazing rays and slim triangles is bounded by the
    reference's own orthogonality and planarity thresholds);
  * projection: which points are on the visible list (0 < u < W, 0 < v < H) and which pixel each one in front of the camera falls into, from the exact image
    coordinates - the oracle's visible list and first-wins pixel map, except for points whose exact coordinate lies
    within 1e-9 px of a pixel or image border (none in these clouds);
  * road fallback: the oracle's plane normal is an eigenvector of the EXACT weighted scatter matrix of the inliers (exact
    Rayleigh quotient, residual below 1e-9 of the matrix's trace - observed 1e-17) for its SMALLEST eigenvalue (LAPACK on the
    rounded matrix), and the oracle's depth is the exact intersection of the viewing ray with the plane through the exact
    weighted centre with that normal (within 1e-10 relative, observed 1e-15).
What this pins: that the restated operation order computes the quantity the reference's mathematics defines, to the last
few bits of f64.  What it cannot pin: that Eigen rounds in the same order (only the reference built against Eigen could).
"""
from fractions import Fraction as Fr

import numpy as np
import pytest

from mono_lidar_depth_amd import capi, synth
from oracle import oracle


def _camera():
    return capi.MldCamera(synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV, synth.KITTI_W, synth.KITTI_H)


def _exact_cam_point(T, p):
    x, y, z = (Fr(float(c)) for c in p)
    return tuple(Fr(float(T[r, 0])) * x + Fr(float(T[r, 1])) * y + Fr(float(T[r, 2])) * z + Fr(float(T[r, 3])) for r in range(3))


def _sq(a, b):
    return sum((a[k] - b[k]) ** 2 for k in range(3))


def _exact_corners(seg):
    """Largest pair first (first maximal pair in row-major order), then the third point of the largest distance sum among
    all but the LAST segmented point.  Returns (i, j, k) and the relative gaps to the runners-up (1 when there is none)."""
    n = len(seg)
    best, bi, bj, second = Fr(-1), -1, -1, Fr(-1)
    for i in range(n - 1):
        for j in range(i + 1, n):
            d = _sq(seg[i], seg[j])
            if d > best:
                second, best, bi, bj = best, d, i, j
            elif d > second:
                second = d
    gap_pair = float((best - second) / best) if second >= 0 and best > 0 else 1.0
    best3, bk, second3 = Fr(-1), -1, Fr(-1)
    for k in range(n - 1):
        if k in (bi, bj):
            continue
        d1, d2 = _sq(seg[k], seg[bi]), _sq(seg[k], seg[bj])
        if d1 <= 0 or d2 <= 0:
            continue
        s = d1 + d2
        if s > best3:
            second3, best3, bk = best3, s, k
        elif s > second3:
            second3 = s
    gap_third = float((best3 - second3) / best3) if second3 >= 0 and best3 > 0 else 1.0
    return (bi, bj, bk), gap_pair, gap_third


def _exact_depth(c1, c2, c3, u, v, cam):
    """z of the intersection of the ray through pixel (u, v) with the plane through c1, c2, c3 (all exact)."""
    a = tuple(c2[k] - c1[k] for k in range(3))
    b = tuple(c3[k] - c1[k] for k in range(3))
    n = (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
    f, cu, cv = Fr(float(cam.focal_length)), Fr(float(cam.principal_point_x)), Fr(float(cam.principal_point_y))
    d = ((Fr(float(u)) - cu) / f, (Fr(float(v)) - cv) / f, Fr(1))   # K^-1 (u, v, 1): z component 1, so depth = the parameter
    nd = sum(n[k] * d[k] for k in range(3))
    if nd == 0:
        return None
    return sum(n[k] * c1[k] for k in range(3)) / nd


@pytest.mark.parametrize("scanner,frame", [("HDL64_KITTI", 0), ("HDL64", 3), ("HDL64_KITTI", 7), ("DENSE128", 2)])
def test_main_path_against_exact_rational_arithmetic(scanner, frame):
    P = capi.params_c0()
    cam = _camera()
    T = np.asarray(synth.T_CAM_LIDAR, dtype=np.float64)[:3, :4]
    cloud = synth.make_cloud(getattr(synth, scanner), seed=40 + frame, frame=frame)
    # features where the window holds enough returns (a quarter of them end on the triangle's plane) and some anywhere
    uv = np.concatenate([synth.make_features_k_neighbours(cloud, 900, seed=900 + frame), synth.make_features(200, seed=900 + frame)])
    ref = oracle.OracleDepthEstimator(P, cam, synth.T_CAM_LIDAR)
    ref.set_cloud(cloud)
    d, t = ref.calculate_depth(uv)
    vis = ref.point_index()
    checked, near_ties, worst = 0, 0, 0.0
    for i in np.nonzero(t == 1)[0]:          # result type 1: depth from the triangle's plane (main path)
        tr = ref.trace_feature(*uv[i])
        assert tr["type"] == 1
        seg = [_exact_cam_point(T, cloud[vis[tr["nb_idx"][p]], :3]) for p in tr["seg_pos"]]
        corners, gap_pair, gap_third = _exact_corners(seg)
        got = tuple(tr["corner_pos"])
        if got != corners:
            # only where exact arithmetic sees a near-tie may the f64 evaluation pick the other candidate
            assert min(gap_pair, gap_third) < 1e-12, (i, got, corners, gap_pair, gap_third)
            near_ties += 1
            continue
        z = _exact_depth(seg[got[0]], seg[got[1]], seg[got[2]], uv[i][0], uv[i][1], cam)
        assert z is not None
        rel = abs(float((Fr(float(d[i])) - z) / z))
        worst = max(worst, rel)
        assert rel < 1e-12, (i, float(d[i]), float(z), rel)
        checked += 1
    assert checked >= 80, checked
    assert near_ties <= max(2, checked // 100), (near_ties, checked)
    print(f"{scanner} frame {frame}: {checked} depths within {worst:.2e} relative of the exact value, {near_ties} near-ties")


@pytest.mark.parametrize("scanner,frame", [("VLP16", 1), ("HDL64_KITTI", 4)])
def test_projection_against_exact_rational_arithmetic(scanner, frame):
    cam = _camera()
    T = np.asarray(synth.T_CAM_LIDAR, dtype=np.float64)[:3, :4]
    cloud = synth.make_cloud(getattr(synth, scanner), seed=60 + frame, frame=frame)
    if scanner != "VLP16":
        cloud = cloud[: 40000]     # (a third of the scan: Fractions are slow; the map is first-wins over THIS cloud)
    ref = oracle.OracleDepthEstimator(capi.params_c0(), cam, synth.T_CAM_LIDAR)
    ref.set_cloud(cloud)
    vis, pm = ref.point_index(), ref.pixel_map()
    f, cu, cv = Fr(float(cam.focal_length)), Fr(float(cam.principal_point_x)), Fr(float(cam.principal_point_y))
    W, H = int(cam.width), int(cam.height)
    exact_vis, cells, borderline = [], {}, 0
    eps = Fr(1, 10 ** 9)
    for i in range(cloud.shape[0]):
        if not np.all(np.isfinite(cloud[i, :3])):
            continue
        x, y, z = _exact_cam_point(T, cloud[i, :3])
        if z == 0:
            continue
        # (DepthEstimator.cpp:186-187: the visible list is decided on the image coordinates alone - a point BEHIND the
        # camera whose projection falls into the image is on it; only z > 0 enters the pixel map, NeighborFinderPixel.cpp:51)
        u, v = (f * x + cu * z) / z, (f * y + cv * z) / z
        if not (0 < u < W and 0 < v < H):
            continue
        fu, fv = u - (u.numerator // u.denominator), v - (v.numerator // v.denominator)
        if min(fu, 1 - fu, fv, 1 - fv) < eps:
            borderline += 1
        if z > 0:
            cell = (v.numerator // v.denominator, u.numerator // u.denominator)
            cells.setdefault(cell, len(exact_vis))      # first point wins (NeighborFinderPixel.cpp:51-54)
        exact_vis.append(i)
    assert borderline == 0      # (otherwise the comparison below would have to excuse those points)
    assert list(vis) == exact_vis
    want = np.full((H, W), -1, dtype=pm.dtype)
    for (r, c), k in cells.items():
        want[r, c] = k
    assert np.array_equal(pm, want)
    assert len(exact_vis) > 1000


def _road_exact(seg, pn, po):
    """PlaneEstimationMEstimator (:18-55) as mathematics: weights 1 / |distance to the prior|, weighted centre, weighted
    scatter matrix - all exact."""
    w = [1 / abs(sum(pn[k] * p[k] for k in range(3)) + po) for p in seg]
    sw = sum(w)
    c = tuple(sum(w[i] * seg[i][k] for i in range(len(seg))) / sw for k in range(3))
    S = [[sum(w[i] * (seg[i][a] - c[a]) * (seg[i][b] - c[b]) for i in range(len(seg))) for b in range(3)] for a in range(3)]
    return c, S


@pytest.mark.parametrize("scanner,frame,tilt", [("HDL64_KITTI", 0, 0.0), ("HDL64", 3, 0.004), ("DENSE128", 2, -0.006)])
def test_road_path_against_exact_rational_arithmetic(scanner, frame, tilt):
    """The road fallback (result type 16): the plane normal the oracle reports must be the eigenvector of the smallest
    eigenvalue of the EXACT weighted scatter matrix (exact Rayleigh quotient and residual; the eigenvalue itself against
    LAPACK on the rounded matrix), and the depth must be the exact intersection of the viewing ray with the plane through the
    exact weighted centre with that normal.  (The prior is applied to camera-frame points with its lidar-frame coefficients,
    as the reference does: RoadDepthEstimatorMEstimator.cpp:42-48.)"""
    P = capi.params_c0()
    cam = _camera()
    T = np.asarray(synth.T_CAM_LIDAR, dtype=np.float64)[:3, :4]
    cloud = synth.make_cloud(getattr(synth, scanner), seed=50 + frame, frame=frame)
    coeffs, inl = synth.make_ground_plane(cloud)
    coeffs = np.asarray(coeffs, dtype=np.float32).copy()
    coeffs[0] += np.float32(tilt)                       # (a prior that is not axis-aligned: the normalisation rounds)
    uv = synth.make_features_k_neighbours(cloud, 500, seed=700 + frame)
    ref = oracle.OracleDepthEstimator(P, cam, synth.T_CAM_LIDAR)
    ref.set_cloud(cloud)
    ref.set_ground_plane(coeffs, inl)
    d, t = ref.calculate_depth(uv)
    vis = ref.point_index()
    cam_pts = ref.cloud_camera_cs()
    cam_pts = cam_pts if cam_pts.shape[1] == 3 else cam_pts.T     # n x 3, the oracle's f64 camera-frame points
    n32 = coeffs[:3].astype(np.float64)
    pn_f = n32 / np.sqrt(float(n32 @ n32))               # the f64 unit normal the estimator starts from
    pn, po = tuple(Fr(float(x)) for x in pn_f), Fr(float(coeffs[3]))
    f, cu, cv = Fr(float(cam.focal_length)), Fr(float(cam.principal_point_x)), Fr(float(cam.principal_point_y))
    checked, worst_res, worst_depth = 0, 0.0, 0.0
    for i in np.nonzero(t == 16)[0][:150]:
        tr = ref.trace_feature(*uv[i])
        raw = [int(vis[tr["road_idx"][p]]) for p in tr["road_pos"]]
        seg = [_exact_cam_point(T, cloud[r, :3]) for r in raw]
        assert len(seg) >= 3
        c, S = _road_exact(seg, pn, po)
        # the oracle's estimator on the oracle's own (rounded) camera-frame points of the same inliers
        n_f64, _ = oracle.mestimator_plane(cam_pts[raw], pn_f, float(coeffs[3]))
        n = tuple(Fr(float(x)) for x in n_f64)
        nn = sum(x * x for x in n)
        Sn = tuple(sum(S[a][b] * n[b] for b in range(3)) for a in range(3))
        lam = sum(n[a] * Sn[a] for a in range(3)) / nn
        res2 = sum((Sn[a] - lam * n[a]) ** 2 for a in range(3))
        trS = S[0][0] + S[1][1] + S[2][2]
        rel_res = float(res2 / (trS * trS * nn)) ** 0.5
        worst_res = max(worst_res, rel_res)
        assert rel_res < 1e-9, (i, rel_res)
        ev = np.linalg.eigvalsh(np.array([[float(x) for x in row] for row in S]))
        assert abs(float(lam) - ev[0]) <= 1e-9 * ev[2], (i, float(lam), ev)        # ... of the SMALLEST eigenvalue
        dvec = ((Fr(float(uv[i][0])) - cu) / f, (Fr(float(uv[i][1])) - cv) / f, Fr(1))
        z = sum(n[k] * c[k] for k in range(3)) / sum(n[k] * dvec[k] for k in range(3))
        rel = abs(float((Fr(float(d[i])) - z) / z))
        worst_depth = max(worst_depth, rel)
        assert rel < 1e-10, (i, float(d[i]), float(z), rel)
        checked += 1
    assert checked >= 20, checked
    print(f"{scanner} frame {frame}: {checked} road depths within {worst_depth:.2e} of the exact intersection, "
          f"eigenvector residual <= {worst_res:.2e}")

"""Second, independent restatement of the reference path in NumPy / plain Python.

TEST INFRASTRUCTURE ONLY (see oracle/mld_oracle.cpp).  Written separately from the C++ oracle, straight from the
reference sources, to cross-check it and to generate the committed golden fixtures (tests/golden/make_golden.py).
Per-feature work is pure-Python loops: use it on small cases only.

Differences to the C++ oracle that are intentional (so that the two do not share a bug):
  * the M-estimator normal comes from numpy.linalg.svd of the 3 x k matrix (the reference uses JacobiSVD),
  * PCA uses numpy.linalg.eigh,
  * 3-vector dot products use Python float arithmetic in index order (last-ulp differences are expected; the
    cross-check tolerance on depths is 1e-9 m, while everything integer — visible list, pixel map, neighbour
    lists, histogram selection, triangle corners, result types — must agree exactly).
"""
from __future__ import annotations

import math

import numpy as np


class NpDepthEstimator:
    def __init__(self, P, cam, T):
        self.P, self.cam = P, cam
        self.T = np.asarray(T, dtype=np.float64)[:3, :4]
        # Initialize (DepthEstimator.cpp:43-44)
        A = np.eye(4)
        A[:3, :4] = self.T
        self.Tinv = np.linalg.inv(A)[:3, :4]
        f, cu, cv = cam.focal_length, cam.principal_point_x, cam.principal_point_y
        self.K = np.array([[f, 0, cu], [0, f, cv], [0, 0, 1.0]])
        self.Kinv = np.linalg.inv(self.K)
        self.plane = None

    # Transform_Cloud_LidarToCamera (DepthEstimator.cpp:156-217) + getImagePoints (camera_pinhole.h:84-97)
    def set_cloud(self, cloud):
        T, cam = self.T, self.cam
        p = np.asarray(cloud, dtype=np.float32)[:, :3].astype(np.float64)
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        with np.errstate(all="ignore"):
            xc = T[0, 3] + ((T[0, 0] * x + T[0, 1] * y) + T[0, 2] * z)
            yc = T[1, 3] + ((T[1, 0] * x + T[1, 1] * y) + T[1, 2] * z)
            zc = T[2, 3] + ((T[2, 0] * x + T[2, 1] * y) + T[2, 2] * z)
            f, cu, cv = cam.focal_length, cam.principal_point_x, cam.principal_point_y
            q0 = (f * xc + 0.0 * yc) + cu * zc
            q1 = (0.0 * xc + f * yc) + cv * zc
            q2 = (0.0 * xc + 0.0 * yc) + 1.0 * zc
            u, v = q0 / q2, q1 / q2
            in_range = (u >= 0.0) & (u <= float(cam.width)) & (v >= 0.0) & (v <= float(cam.height))
            strict = (u > 0) & (u < cam.width) & (v > 0) & (v < cam.height)
        self.cam_pts = np.stack([xc, yc, zc])
        self.img_pts = np.stack([u, v])
        self.in_range = in_range
        self.point_index = np.nonzero(in_range & strict)[0].astype(np.int32)
        self.img_vis = self.img_pts[:, self.point_index]
        # InitializeLidarProjection (NeighborFinderPixel.cpp:29-58)
        W, H = cam.width, cam.height
        pm = np.full((H, W), -1, dtype=np.int32)
        for i, raw in enumerate(self.point_index):
            xi, yi = int(self.img_vis[0, i]), int(self.img_vis[1, i])
            if pm[yi, xi] == -1 and self.cam_pts[2, raw] > 0:
                pm[yi, xi] = i
        self.pixel_map = pm
        self.plane = None

    def set_ground_plane(self, coeffs, inliers):
        if coeffs is None:
            self.plane = None
            return
        c = np.asarray(coeffs, dtype=np.float32)
        n = c[:3].astype(np.float64)
        self.plane = {"coeffs": c, "inliers": set(int(i) for i in np.asarray(inliers)),
                      "prior_n": n / math.sqrt(float(n @ n)), "prior_off": float(c[3])}

    # NeighborFinderPixel::getNeighbors (NeighborFinderPixel.cpp:60-95)
    def neighbours(self, u, v, sx, sy):
        P, W, H = self.P, self.cam.width, self.cam.height
        if not (np.isfinite(u) and np.isfinite(v)):  # undefined in the reference; an empty window here
            return [], []
        hx = float(P.pixelarea_search_witdh) * 0.5 * float(np.float32(sx))
        hy = float(P.pixelarea_search_height) * 0.5 * float(np.float32(sy))
        left, right = max(u - hx, 0.0), min(u + hx, float(W - 1))
        top, bottom = max(v - hy, 0.0), min(v + hy, float(H - 1))
        idx = []
        for i in range(int(top), int(bottom) + 1):
            for j in range(int(left), int(right) + 1):
                if self.pixel_map[i, j] != -1:
                    idx.append(int(self.pixel_map[i, j]))
        pts = [self.cam_pts[:, self.point_index[k]].copy() for k in idx]
        return idx, pts

    # PointHistogram::FilterPointsMinDistBlob (HistogramPointDepth.cpp:15-123)
    @staticmethod
    def histogram_filter(depths, bin_w, min_count):
        max_dist = 0
        for d in depths:
            if d > max_dist:
                max_dist = int(math.ceil(d))
        bin_count = int(max_dist / bin_w + 1)
        if bin_count <= 1:
            return None
        bins = [0] * bin_count
        for d in depths:
            val = min(d, 1e10)
            bins[int(min(abs(val / bin_w), float(bin_count) - 1.0))] += 1
        max_id, max_val, val = -1, -1, 0
        for i in range(bin_count):
            last = val
            val = bins[i]
            if val > max_val and val >= min_count:
                max_val, max_id = val, i
            elif val < max_val:
                break
            if last > 0 and val == 0:
                return None
        if max_id < 0:
            return None
        lo = max_id * bin_w - 0.0 * bin_w
        hi = max_id * bin_w + 1.0 * bin_w
        return [k for k, d in enumerate(depths) if lo <= d < hi]

    # PlaneEstimationCalcMaxSpanningTriangle::CalculatePlaneCorners (:37-100), _distTreshold = 0
    @staticmethod
    def triangle(pts, thr=0.0):
        n = len(pts)
        if n < 3:
            return None

        def sq(a, b):
            d = a - b
            return d[0] * d[0] + (d[1] * d[1] + d[2] * d[2])

        mi = mj = -1
        md = -1.0
        for i in range(n - 1):
            for j in range(i + 1, n):
                d = sq(pts[i], pts[j])
                if d > md:
                    md, mi, mj = d, i, j
        if md <= thr:
            return None
        md2, mk = -1.0, -1
        for k in range(n - 1):
            if k == mi or k == mj:
                continue
            d1 = sq(pts[k], pts[mi])
            if d1 <= thr:
                continue
            d2 = sq(pts[k], pts[mj])
            if d2 <= thr:
                continue
            if d1 + d2 > md2:
                md2, mk = d1 + d2, k
        if mi == -1 or mj == -1 or mk == -1:
            return None
        return mi, mj, mk

    def viewing_ray(self, u, v):
        d = self.Kinv @ np.array([u, v, 1.0])
        d = d / np.linalg.norm(d)
        if d[2] < 0:
            d = -d
        return d

    @staticmethod
    def line_plane(n, offset, n0, n1):
        direction = (n1 - n0) / np.linalg.norm(n1 - n0)
        t = -(offset + float(n @ n0)) / float(n @ direction)
        return n0 + direction * t

    def thresholds(self, depth, pts):
        P = self.P
        if P.treshold_depth_enabled:
            if depth < P.treshold_depth_min:
                if P.treshold_depth_mode == 0:
                    return 5, depth
                depth = float(P.treshold_depth_min)
            elif depth > P.treshold_depth_max:
                if P.treshold_depth_mode == 0:
                    return 4, depth
                depth = float(P.treshold_depth_max)
        if P.treshold_depth_local_enabled:
            zs = [p[2] for p in pts]
            mn, mx = min(zs), max(zs)
            if P.treshold_depth_local_valuetype == 1:
                r = (mx - mn) * P.treshold_depth_local_value
                lo, hi = mn - r, mx + r
            else:
                lo, hi = mn - P.treshold_depth_local_value, mx + P.treshold_depth_local_value
            if depth < lo:
                if P.treshold_depth_local_mode == 0:
                    return 7, depth
                depth = lo
            elif depth > hi:
                if P.treshold_depth_local_mode == 0:
                    return 6, depth
                depth = hi
        return 0, depth

    # DepthEstimator::CalculateDepthSegmented (DepthEstimator.cpp:903-1037), triangle variant
    def depth_segmented(self, u, v, seg, trace):
        P = self.P
        if not P.do_use_PCA and P.do_use_triangle_size_maximation:
            tri = self.triangle(seg)
            if tri is None:
                return 9, -1.0
        else:
            if len(seg) < 3:
                return 3, -1.0
            tri = (0, 1, 2)
        trace["corners"] = tri
        c1, c2, c3 = seg[tri[0]], seg[tri[1]], seg[tri[2]]
        if not P.do_use_PCA and P.do_check_triangleplanar_condition:
            e1, e2, e3 = c2 - c1, c3 - c1, c3 - c2
            e1, e2, e3 = e1 / np.linalg.norm(e1), e2 / np.linalg.norm(e2), e3 / np.linalg.norm(e3)
            thr = P.triangleplanar_crossnorm_treshold
            if not (np.linalg.norm(np.cross(e1, e2)) >= thr and np.linalg.norm(np.cross(e1, e3)) >= thr
                    and np.linalg.norm(np.cross(e2, e3)) >= thr):
                return 8, -1.0
        ray = self.viewing_ray(u, v)
        if P.do_use_PCA:
            X = np.stack(seg, axis=1)
            mean = X.mean(axis=1)
            C = (X - mean[:, None]) @ (X - mean[:, None]).T
            w, V = np.linalg.eigh(C)
            planarity = np.float32((w[1] - w[0]) / w[2])
            linearity = np.float32((w[2] - w[1]) / w[2])
            if planarity < P.pca_treshold_2_1_rel_min:
                return 14, -1.0
            if linearity > P.pca_treshold_3_2_rel_max:
                return 13, -1.0
            if w[2] < P.pca_treshold_3_abs_min:
                return 12, -1.0
            n = V[:, 0] / np.linalg.norm(V[:, 0])
            off = -float(n @ mean)
        else:
            v0, v1 = c3 - c1, c2 - c1
            n = np.cross(v0, v1)
            n = n / np.linalg.norm(n)
            off = -float(c1 @ n)
        if P.viewray_plane_orthoganality_treshold > 0:
            if not abs(float((n / np.linalg.norm(n)) @ (ray / np.linalg.norm(ray)))) >= P.viewray_plane_orthoganality_treshold:
                return 11, -1.0
        ip = self.line_plane(n, off, np.zeros(3), ray)
        depth = float(ip[2])
        code, depth = self.thresholds(depth, seg)
        if code:
            return code, -1.0
        if depth < 0 and P.do_use_cut_behind_camera:
            return 10, -1.0
        return 1, depth

    # DepthEstimator::CalculateDepth(Vector2d, ...) (DepthEstimator.cpp:491-600)
    def feature(self, u, v):
        P = self.P
        trace = {}
        idx, nb = self.neighbours(u, v, 1.0, 1.0)
        trace["nb_idx"] = idx
        if len(nb) < np.uint32(P.radiusSearch_count_min):
            return 2, -1.0, trace
        result = (0, -1.0)
        seg = None
        if P.do_use_histogram_segmentation:
            depths = [min(float(p[2]), 999.0) for p in nb]
            keep = self.histogram_filter(depths, P.histogram_segmentation_bin_witdh,
                                         P.histogram_segmentation_min_pointcount)
            if keep is None:
                result = (3, -1.0)
            else:
                seg = [nb[k] for k in keep]
                trace["seg_pos"] = keep
        else:
            seg = nb
            trace["seg_pos"] = list(range(len(nb)))
        if result[0] != 3:
            result = self.depth_segmented(u, v, seg, trace)
            if result[0] == 1:
                return result[0], result[1], trace
        old = result[0]
        if self.plane is not None and P.do_use_ransac_plane:
            idx, nb = self.neighbours(u, v, 2.0, 1.5)
            trace["road_idx"] = idx
            if len(nb) < np.uint32(P.radiusSearch_count_min):
                return 2, -1.0, trace
            # CalculateDepthSegmentationPlane (:782-900)
            seg, pos = [], []
            c = self.plane["coeffs"]
            for i, p in enumerate(nb):
                raw = int(self.point_index[idx[i]])
                pl = (self.Tinv[:, :3] @ p + self.Tinv[:, 3]).astype(np.float32)
                d = abs(((c[0] * pl[0] + c[1] * pl[1]) + c[2] * pl[2]) + c[3])  # float32 scalars throughout
                if float(d) > P.ransac_plane_point_distance_treshold:
                    return old, -1.0, trace
                if raw in self.plane["inliers"]:
                    seg.append(p)
                    pos.append(i)
            trace["road_pos"] = pos
            if len(seg) < 3:
                return old, -1.0, trace
            ray = self.viewing_ray(u, v)
            if P.plane_estimator_use_triangle_maximation:
                tri = self.triangle(seg)
                if tri is None:
                    return 2, -1.0, trace
                xs, zs = [p[0] for p in seg], [p[2] for p in seg]
                with np.errstate(all="ignore"):
                    rel = np.float64(max(zs) - min(zs)) / np.float64(max(xs) - min(xs))
                if not rel >= P.plane_estimator_z_x_min_relation:
                    return 15, -1.0, trace
                c1, c2, c3 = seg[tri[0]], seg[tri[1]], seg[tri[2]]
                n = np.cross(c3 - c1, c2 - c1)
                n = n / np.linalg.norm(n)
                off = -float(c1 @ n)
            else:
                # PlaneEstimationMEstimator::EstimatePlane (:18-55)
                pn, po = self.plane["prior_n"], self.plane["prior_off"]
                with np.errstate(all="ignore"):
                    w = np.array([1.0 / abs(float(pn @ p) + po) for p in seg])
                    X = np.stack(seg, axis=1)
                    center = (X * w[None, :]).sum(axis=1) / w.sum()
                    M = (X - center[:, None]) * np.sqrt(w)[None, :]
                if not np.all(np.isfinite(M)):
                    return 16, float("nan"), trace
                U, S, _ = np.linalg.svd(M, full_matrices=False)
                n = U[:, -1] / np.linalg.norm(U[:, -1])
                off = -float(n @ center)
            ip = self.line_plane(n, off, ray, np.zeros(3))
            depth = float(ip[2])
            code, depth = self.thresholds(depth, seg)
            if code:
                return code, -1.0, trace
            return 16, depth, trace
        return result[0], result[1], trace

    def calculate_depth(self, uv):
        if self.P.set_all_depths_to_zero:
            return np.full(len(uv), -1.0), np.ones(len(uv), dtype=np.int32), [{} for _ in uv]
        out = [self.feature(float(a), float(b)) for a, b in uv]
        return (np.array([o[1] for o in out]), np.array([o[0] for o in out], dtype=np.int32), [o[2] for o in out])


def _f32_partials_sum(values: np.ndarray, partials: int = 256) -> np.float32:
    """Sum of float32 `values`: partial p accumulates entries p, p+256, ... sequentially, then the partials are
    added in index order (the association of ls_plane_fit in the C++ restatement)."""
    m = values.shape[0]
    rows = (m + partials - 1) // partials
    pad = np.zeros(rows * partials, dtype=np.float32)
    pad[:m] = values
    part = np.add.accumulate(pad.reshape(rows, partials), axis=0, dtype=np.float32)[-1]
    return np.add.accumulate(part, dtype=np.float32)[-1]


def ls_plane_fit(xyz: np.ndarray, idx: np.ndarray, fallback):
    """optimizeModelCoefficients (PCL sac_model_plane.hpp): float32 moments, smallest eigenvector (LAPACK here, Jacobi
    in the C++ restatement: same subspace, last-bit differences allowed)."""
    if idx.size < 4:
        return np.asarray(fallback, dtype=np.float32)
    v = xyz[idx].astype(np.float32)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    m = np.float32(idx.size)
    a = [_f32_partials_sum(q) / m for q in (x * x, x * y, x * z, y * y, y * z, z * z, x, y, z)]
    cov = np.array([[a[0] - a[6] * a[6], a[1] - a[6] * a[7], a[2] - a[6] * a[8]],
                    [a[1] - a[6] * a[7], a[3] - a[7] * a[7], a[4] - a[7] * a[8]],
                    [a[2] - a[6] * a[8], a[4] - a[7] * a[8], a[5] - a[8] * a[8]]], dtype=np.float64)
    w, vec = np.linalg.eigh(cov)
    n = vec[:, 0].astype(np.float32)
    d = np.float32(-1.0) * (n[0] * a[6] + n[1] * a[7] + n[2] * a[8])
    return np.array([n[0], n[1], n[2], d], dtype=np.float32)


def _f32_group_tree_sum(values: np.ndarray) -> np.float32:
    """Sum of float32 `values` given for EVERY cloud point (0 for non-members), the association of the semantic plane's
    fits (C++ restatement ls_plane_fit, HIP k_sem_*): a fixed binary tree over each group of 64 consecutive points
    (strides 32, 16, ..., 1), then the group sums through 256 interleaved partials combined in index order."""
    n = values.shape[0]
    groups = (n + 63) // 64
    a = np.zeros(groups * 64, dtype=np.float32)
    a[:n] = values
    a = a.reshape(groups, 64)
    o = 32
    while o > 0:
        a = (a[:, :o] + a[:, o:2 * o]).astype(np.float32)
        o //= 2
    return _f32_partials_sum(a[:, 0])


def ls_plane_fit_members(xyz: np.ndarray, member: np.ndarray, fallback):
    """optimizeModelCoefficients over the cloud points flagged in `member` (semantic plane): float32 moments in the
    group-tree association, smallest eigenvector (LAPACK here, Jacobi in the C++ restatement)."""
    m_int = int(member.sum())
    if m_int < 4:
        return np.asarray(fallback, dtype=np.float32)
    v = np.where(member[:, None], xyz.astype(np.float32), np.float32(0.0)).astype(np.float32)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    m = np.float32(m_int)
    a = [_f32_group_tree_sum(q) / m for q in (x * x, x * y, x * z, y * y, y * z, z * z, x, y, z)]
    cov = np.array([[a[0] - a[6] * a[6], a[1] - a[6] * a[7], a[2] - a[6] * a[8]],
                    [a[1] - a[6] * a[7], a[3] - a[7] * a[7], a[4] - a[7] * a[8]],
                    [a[2] - a[6] * a[8], a[4] - a[7] * a[8], a[5] - a[8] * a[8]]], dtype=np.float64)
    w, vec = np.linalg.eigh(cov)
    n = vec[:, 0].astype(np.float32)
    d = np.float32(-1.0) * (n[0] * a[6] + n[1] * a[7] + n[2] * a[8])
    return np.array([n[0], n[1], n[2], d], dtype=np.float32)


def semantic_plane(cloud: np.ndarray, T, f, cu, cv, img: np.ndarray, labels, thr: float):
    """SemanticPlane::CalculateInliersPlane (monolidar_fusion/src/RansacPlane.cpp:195-274), vectorised.
    Returns (candidate indices, first fit, inlier indices, refined fit)."""
    T = np.asarray(T, dtype=np.float64).reshape(3, 4)
    xyz = cloud[:, :3].astype(np.float64)
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pc = [(((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3]).astype(np.float32).astype(np.float64) for r in range(3)]
    with np.errstate(all="ignore"):
        p0 = f * pc[0] + (0.0 * pc[1] + cu * pc[2])
        p1 = 0.0 * pc[0] + (f * pc[1] + cv * pc[2])
        p2 = 0.0 * pc[0] + (0.0 * pc[1] + 1.0 * pc[2])
        u, v = p0 / p2, p1 / p2
        ok = np.isfinite(u) & np.isfinite(v) & (np.abs(u) < 2147483648.0) & (np.abs(v) < 2147483648.0)
        ix = np.where(ok, np.trunc(np.where(ok, u, 0.0)), -1).astype(np.int64)
        iy = np.where(ok, np.trunc(np.where(ok, v, 0.0)), -1).astype(np.int64)
    rows, cols = img.shape
    ok &= (ix >= 0) & (ix < cols) & (iy >= 0) & (iy < rows)
    lab = np.zeros(cloud.shape[0], dtype=np.int64) - 1
    lab[ok] = img[iy[ok], ix[ok]]
    cand = np.nonzero(np.isin(lab, np.asarray(labels)))[0].astype(np.int32)
    if cand.size < 3:
        raise ValueError("In GroundPlane: Input pointcloud is invalid")
    xyz32 = cloud[:, :3].astype(np.float32)
    is_cand = np.zeros(cloud.shape[0], dtype=bool)
    is_cand[cand] = True
    c1 = ls_plane_fit_members(xyz32, is_cand, [0, 0, 1, 0])
    dist = np.abs(((c1[0] * xyz32[:, 0] + c1[1] * xyz32[:, 1]) + c1[2] * xyz32[:, 2]) + c1[3])
    sel = dist.astype(np.float64) < thr
    inl = np.nonzero(sel)[0].astype(np.int32)
    c2 = ls_plane_fit_members(xyz32, sel, c1)
    return cand, c1, inl, c2


# ---- RansacPlane::CalculateInliersPlane (monolidar_fusion/src/RansacPlane.cpp:41-140), second restatement ----------

def _mix(a: int, b: int, c: int) -> int:
    """Counter-based hash of the draws (shared convention of the C++ restatement and the HIP kernels)."""
    M = 0xFFFFFFFF
    h = (a * 0x9E3779B1) & M
    h ^= (b + 0x85EBCA6B + ((h << 6) & M) + (h >> 2)) & M
    h ^= (((c * 0xC2B2AE35) & M) + ((h << 6) & M) + (h >> 2)) & M
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M
    h ^= h >> 16
    return h


def _plane_from(p0, p1, p2):
    """sac_model_plane computeModelCoefficients + sac_model_perpendicular_plane isModelValid, float32 throughout."""
    f = np.float32
    with np.errstate(all="ignore"):
        a = (p1 - p0).astype(f)
        b = (p2 - p0).astype(f)
        r = a / b
        degenerate = bool(r[0] == r[1] and r[2] == r[1])
        n = np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=f)
        nn = np.sqrt(f(f(n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]))
        n = (n / nn).astype(f)
        d = f(-1.0) * f(f(n[0] * p0[0] + n[1] * p0[1]) + n[2] * p0[2])
    if not (nn > 0) or not np.isfinite(nn):
        degenerate = True
    valid = (not degenerate) and abs(float(n[2])) >= 0.984807753012208
    return np.array([n[0], n[1], n[2], d], dtype=f), degenerate, valid


def _plane_dist(c, pts):
    f = np.float32
    return np.abs(((c[0] * pts[:, 0] + c[1] * pts[:, 1]).astype(f) + c[2] * pts[:, 2]).astype(f) + c[3]).astype(f)


def ransac_plane(cloud: np.ndarray, P, seed: int):
    """Returns (coefficients float32[4], ascending inlier indices) like OracleDepthEstimator.estimate_ground_plane."""
    k_sample = 6000
    xyz = np.ascontiguousarray(cloud[:, :3], dtype=np.float32)
    n = xyz.shape[0]
    if n < 3:
        raise ValueError("In GroundPlane: Input pointcloud is invalid")
    if P.ransac_plane_min_z > -1001.0:
        lo, hi = np.float32(P.ransac_plane_min_z), np.float32(P.ransac_plane_max_z)
        with np.errstate(invalid="ignore"):
            ok = np.isfinite(xyz).all(axis=1) & ~(xyz[:, 2] < lo) & ~(xyz[:, 2] > hi)
        cand = np.nonzero(ok)[0].astype(np.int64)
    else:
        cand = np.arange(n, dtype=np.int64)
    M = cand.size
    if M > k_sample:
        sample = np.empty(k_sample, dtype=np.int64)
        for j in range(k_sample):
            u = _mix(seed, j, 0x5A17) * (1.0 / 4294967296.0)
            pos = int((float(j) + u) * float(M) / float(k_sample))
            sample[j] = cand[min(pos, M - 1)]
    else:
        sample = cand
    S = sample.size
    if S < 3:
        raise ValueError("In GroundPlane: Input pointcloud is invalid")
    sp = xyz[sample]

    def draw(d):
        a = _mix(seed, d, 1) % S
        b = _mix(seed, d, 2) % S
        c = _mix(seed, d, 3) % S
        if b == a:
            b = (b + 1) % S
        while c == a or c == b:
            c = (c + 1) % S
        return _plane_from(sp[a], sp[b], sp[c])

    thr = float(P.ransac_plane_distance_treshold)
    max_it = int(P.ransac_plane_max_iterations)
    iterations, best, best_draw, k = 0, -2147483647, -1, 1.0
    log_probability = np.log(1.0 - P.ransac_plane_probability)
    eps = np.finfo(np.float64).eps
    d = 0
    while d < max_it + 1 and iterations < k:
        c, degenerate, valid = draw(d)
        d += 1
        if degenerate:
            continue
        cnt = int((_plane_dist(c, sp).astype(np.float64) < thr).sum()) if valid else 0
        if cnt > best:
            best, best_draw = cnt, d - 1
            w = best / float(S)
            p_no = min(1.0 - eps, max(eps, 1.0 - w * w * w))
            k = log_probability / np.log(p_no)
        iterations += 1
        if iterations > max_it:
            break
    if best_draw < 0:
        raise ValueError("In GroundPlane: Input pointcloud is invalid")
    bm, _, bvalid = draw(best_draw)
    coeffs = bm.copy()
    inl = np.nonzero(_plane_dist(bm, sp).astype(np.float64) < thr)[0] if bvalid else np.zeros(0, dtype=np.int64)
    if P.ransac_plane_use_refinement:
        if inl.size > 3:
            coeffs = ls_plane_fit(sp, inl.astype(np.int64), coeffs)
        inl = (np.nonzero(_plane_dist(bm, sp).astype(np.float64) < P.ransac_plane_refinement_treshold)[0]
               if bvalid else np.zeros(0, dtype=np.int64))
    return coeffs, np.unique(sample[inl]).astype(np.int32)

/*
 * mld_oracle.cpp — CPU restatement of monolidar_fusion's DepthEstimator hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (mono_lidar_depth_amd/, include/)
 * links, loads or calls this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may use it, and only as the checker / reported CPU baseline.
 *
 * What it is: a dependency-free C++17 restatement of the reference algorithm, same stages and
 * operation order, `#pragma omp parallel for` over features exactly where the reference has it
 * (monolidar_fusion/src/DepthEstimator.cpp:455).  Every function cites the reference lines it follows.
 *
 * PARITY STATUS
 *   - The reference as a whole cannot be compiled here (Eigen, PCL, OpenCV absent).  Its two
 *     dependency-free units of the path, Histogram.cpp and TresholdDepthGlobal.cpp, are compiled from
 *     /root/reference into oracle/_ref/libmld_ref.so (oracle/ref_shim.cpp, `make ref`) and the corresponding
 *     functions here are checked against that object code (tests/test_reference_parts.py).
 *   - Pinned by the reference's own tests for this path: Histogram.FilterPointsMinDistBlob (exact
 *     KAT), Histogram.GetNearestPoint, NeigborFinder.findByPixel (property bounds)
 *     (monolidar_fusion/test/test_monolidar_fusion.cpp:82-171,277-374) — see tests/test_oracle_kat.py.
 *   - End-to-end CalculateDepth: PARITY UNPINNED — the reference holds no test or fixture that
 *     exercises it.  Eigen (un-vendored, no version pin: monolidar_fusion/package.xml:19) arithmetic
 *     is restated from its published 3.3 semantics: summation order inside 3-vectors
 *     follows Eigen's unrolled fixed-size reduction x0 + (x1 + x2); per-point matrix products use
 *     ((a0 + a1) + a2).  These choices matter at the 1-ulp level only.
 *   - JacobiSVD (PlaneEstimationMEstimator.cpp:49) is restated as a one-sided Jacobi SVD (same
 *     singular subspace, not the same rounding).
 *
 * Build: g++ -O2 -fopenmp -ffp-contract=off -shared -fPIC (see oracle/Makefile).  FMA contraction is
 * off so that pixel truncation (NeighborFinderPixel.cpp:41-42) is reproducible bit-for-bit.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/mld.h"

namespace {

struct V3 {
    double x, y, z;
};

// ---- Eigen fixed-size Vector3d semantics -------------------------------------
inline V3 sub(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 scale(const V3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 divs(const V3& a, double s) { return {a.x / s, a.y / s, a.z / s}; }
// redux over a fixed-size 3 expression: func(x0, func(x1, x2))
inline double dot(const V3& a, const V3& b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline double sqnorm(const V3& a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
inline double norm(const V3& a) { return std::sqrt(sqnorm(a)); }
// MatrixBase::normalize()/normalized(): z = squaredNorm(); if (z > 0) v /= sqrt(z)
inline V3 normalized(const V3& a) {
    double z = sqnorm(a);
    if (z > 0.0) return divs(a, std::sqrt(z));
    return a;
}
inline V3 cross(const V3& a, const V3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct Plane {  // Eigen::Hyperplane<double,3>: normal, offset
    V3 n;
    double offset;
};

// 3x3 inverse by cofactors, as Eigen's compute_inverse<Matrix3d> (Inverse_SSE not used for 3x3):
// result(i,j) = cofactor<j,i>(m) * (1/det), det = c00*m00 + (c10*m10 + c20*m20).
inline double cof(const double m[9], int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
void inverse3(const double m[9], double out[9]) {  // row-major
    double c0 = cof(m, 0, 0), c1 = cof(m, 1, 0), c2 = cof(m, 2, 0);
    double det = c0 * m[0] + (c1 * m[3] + c2 * m[6]);
    double invdet = 1.0 / det;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[i * 3 + j] = cof(m, j, i) * invdet;
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi; used for the degenerate branch of
// Hyperplane::Through and for PCA (SelfAdjointEigenSolver, PCA.cpp:52).  Eigenvalues ascending.
void jacobi_eig3(const double a_in[9], double eval[3], double evec[9] /* columns */) {
    double a[9];
    std::memcpy(a, a_in, sizeof(a));
    double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        double diag = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
        if (!(off > 1e-300) || off <= 1e-32 * diag) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = a[p * 3 + q];
                if (apq == 0.0) continue;
                double app = a[p * 3 + p], aqq = a[q * 3 + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) {  // A <- A J
                    double akp = a[k * 3 + p], akq = a[k * 3 + q];
                    a[k * 3 + p] = c * akp - s * akq;
                    a[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {  // A <- J^T A
                    double apk = a[p * 3 + k], aqk = a[q * 3 + k];
                    a[p * 3 + k] = c * apk - s * aqk;
                    a[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = v[k * 3 + p], vkq = v[k * 3 + q];
                    v[k * 3 + p] = c * vkp - s * vkq;
                    v[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    int idx[3] = {0, 1, 2};
    double d[3] = {a[0], a[4], a[8]};
    std::sort(idx, idx + 3, [&](int i, int j) { return d[i] < d[j]; });
    for (int c = 0; c < 3; c++) {
        eval[c] = d[idx[c]];
        for (int r = 0; r < 3; r++) evec[r * 3 + c] = v[r * 3 + idx[c]];
    }
}

// Eigen::Hyperplane<double,3>::Through(p0,p1,p2) (call site LinePlaneIntersectionBase.cpp:41).
Plane plane_through(const V3& p0, const V3& p1, const V3& p2) {
    V3 v0 = sub(p2, p0), v1 = sub(p1, p0);
    V3 n = cross(v0, v1);
    double nn = norm(n);
    Plane r;
    if (nn <= norm(v0) * norm(v1) * std::numeric_limits<double>::epsilon()) {
        // Degenerate: Eigen takes the last right-singular vector of [v0^T; v1^T] (JacobiSVD, FullV).
        // Restated as the eigenvector of the smallest eigenvalue of m^T m.  Unreachable in C0
        // (CheckPlanar rejects such triangles first); parity unpinned for this branch.
        double m[9] = {v0.x * v0.x + v1.x * v1.x, v0.x * v0.y + v1.x * v1.y, v0.x * v0.z + v1.x * v1.z,
                       0, v0.y * v0.y + v1.y * v1.y, v0.y * v0.z + v1.y * v1.z,
                       0, 0, v0.z * v0.z + v1.z * v1.z};
        m[3] = m[1];
        m[6] = m[2];
        m[7] = m[5];
        double ev[3], evec[9];
        jacobi_eig3(m, ev, evec);
        r.n = {evec[0], evec[3], evec[6]};
    } else {
        r.n = divs(n, nn);
    }
    r.offset = -dot(p0, r.n);
    return r;
}

// ParametrizedLine::Through(a,b).intersectionPoint(plane).
V3 line_through_intersection(const V3& a, const V3& b, const Plane& pl) {
    V3 dir = normalized(sub(b, a));
    double t = -(pl.offset + dot(pl.n, a)) / dot(pl.n, dir);
    return add(a, scale(dir, t));
}

struct Frame {
    mld_params P;
    mld_camera cam;
    double T[12];     // lidar->camera, row-major 3x4
    double Tinv[12];  // camera->lidar (DepthEstimator.cpp:44)
    double Kinv[9];   // intrinsics.inverse() (camera_pinhole.h:65)

    // PointcloudData (PointcloudData.h:13-68)
    int64_t n = 0;
    std::vector<double> pts_cam;        // 3 x N col-major   (_points_cs_camera)
    std::vector<double> pts_img;        // 2 x N col-major   (_points_cs_image)
    std::vector<uint8_t> in_range;      // N                 (_pointsInImgRange)
    std::vector<double> pts_img_vis;    // 2 x Nvis          (_points_cs_image_visible)
    std::vector<int32_t> point_index;   // Nvis              (_pointIndex)
    std::vector<int32_t> pixel_map;     // W*H, x + y*W      (NeighborFinderPixel::_img_points_lidar)
    bool cloud_set = false;

    // GroundPlane (RansacPlane.h:84-122)
    bool has_plane = false;
    float coeffs[4] = {0, 0, 0, 0};
    std::vector<uint8_t> inlier;  // N, _pointIsInPlane.count(i)
    Plane prior;                  // M-estimator prior (DepthEstimator.cpp:286-292)
};

void make_calibration(Frame& fr) {
    const double* T = fr.T;
    // Transform::inverse(Affine): linear().inverse(); translation = -linear_inv * t
    double L[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    double Li[9];
    inverse3(L, Li);
    double t[3] = {T[3], T[7], T[11]};
    for (int i = 0; i < 3; i++) {
        fr.Tinv[i * 4 + 0] = Li[i * 3 + 0];
        fr.Tinv[i * 4 + 1] = Li[i * 3 + 1];
        fr.Tinv[i * 4 + 2] = Li[i * 3 + 2];
        fr.Tinv[i * 4 + 3] = (-Li[i * 3 + 0]) * t[0] + ((-Li[i * 3 + 1]) * t[1] + (-Li[i * 3 + 2]) * t[2]);
    }
    // makeIntrinsics (camera_pinhole.h:100-106)
    double K[9] = {fr.cam.focal_length, 0, fr.cam.principal_point_x, 0, fr.cam.focal_length,
                   fr.cam.principal_point_y, 0, 0, 1};
    inverse3(K, fr.Kinv);
}

// Transform_Cloud_LidarToCamera (DepthEstimator.cpp:156-217) + CameraPinhole::getImagePoints
// (camera_pinhole.h:84-97).
void transform_cloud(Frame& fr, const uint8_t* pts, int64_t n, int stride) {
    fr.n = n;
    fr.pts_cam.assign(3 * n, 0.0);
    fr.pts_img.assign(2 * n, 0.0);
    fr.in_range.assign(n, 0);
    const double* T = fr.T;
    const double f = fr.cam.focal_length, cu = fr.cam.principal_point_x, cv = fr.cam.principal_point_y;
    const double W = static_cast<double>(fr.cam.width), H = static_cast<double>(fr.cam.height);
    for (int64_t i = 0; i < n; i++) {
        const float* p = reinterpret_cast<const float*>(pts + i * stride);
        // :169 float -> double
        double x = static_cast<double>(p[0]), y = static_cast<double>(p[1]), z = static_cast<double>(p[2]);
        // :173 P_cam = translation + linear * p   (product summed k = 0,1,2, then added to t)
        double xc = T[3] + ((T[0] * x + T[1] * y) + T[2] * z);
        double yc = T[7] + ((T[4] * x + T[5] * y) + T[6] * z);
        double zc = T[11] + ((T[8] * x + T[9] * y) + T[10] * z);
        fr.pts_cam[3 * i + 0] = xc;
        fr.pts_cam[3 * i + 1] = yc;
        fr.pts_cam[3 * i + 2] = zc;
        // camera_pinhole.h:88-90  q = K p (all nine terms), hnormalized
        double q0 = (f * xc + 0.0 * yc) + cu * zc;
        double q1 = (0.0 * xc + f * yc) + cv * zc;
        double q2 = (0.0 * xc + 0.0 * yc) + 1.0 * zc;
        double u = q0 / q2, v = q1 / q2;
        fr.pts_img[2 * i + 0] = u;
        fr.pts_img[2 * i + 1] = v;
        // :93-95 inclusive bounds
        fr.in_range[i] = (u >= 0.) && (u <= W) && (v >= 0.) && (v <= H);
    }
    // :184-207 two passes, strict bounds, order preserved
    fr.pts_img_vis.clear();
    fr.point_index.clear();
    for (int64_t i = 0; i < n; i++) {
        if (fr.in_range[i]) {
            double u = fr.pts_img[2 * i], v = fr.pts_img[2 * i + 1];
            if ((u > 0) && (u < fr.cam.width) && (v > 0) && (v < fr.cam.height)) {
                fr.pts_img_vis.push_back(u);
                fr.pts_img_vis.push_back(v);
                fr.point_index.push_back(static_cast<int32_t>(i));
            }
        }
    }
}

// NeighborFinderPixel::InitializeLidarProjection (NeighborFinderPixel.cpp:29-58)
void init_lidar_projection(Frame& fr) {
    const int W = fr.cam.width, H = fr.cam.height;
    fr.pixel_map.assign(static_cast<size_t>(W) * H, -1);  // :38 setConstant(POINT_NOT_DEFINED)
    const int64_t nvis = static_cast<int64_t>(fr.point_index.size());
    for (int64_t i = 0; i < nvis; i++) {
        int x_img = static_cast<int>(fr.pts_img_vis[2 * i]);  // :41 truncation
        int y_img = static_cast<int>(fr.pts_img_vis[2 * i + 1]);
        int indexRaw = fr.point_index[i];
        double zc = fr.pts_cam[3 * static_cast<int64_t>(indexRaw) + 2];
        int32_t& cell = fr.pixel_map[static_cast<size_t>(x_img) + static_cast<size_t>(y_img) * W];
        if (cell == -1 && zc > 0) cell = static_cast<int32_t>(i);  // :51-54 first wins, z > 0
    }
}

// NeighborFinderPixel::getNeighbors (NeighborFinderPixel.cpp:60-95): visible indices, row-major window scan.
void get_neighbor_indices(const Frame& fr, double u, double v, float scaleW, float scaleH,
                          std::vector<int32_t>& out) {
    const int W = fr.cam.width, H = fr.cam.height;
    // NaN / inf coordinates: the reference casts them to int and indexes the map with the result (undefined behaviour);
    // defined here, as in the HIP path, as an empty window
    if (!std::isfinite(u) || !std::isfinite(v)) return;
    double halfX = static_cast<double>(fr.P.pixelarea_search_witdh) * 0.5 * static_cast<double>(scaleW);
    double halfY = static_cast<double>(fr.P.pixelarea_search_height) * 0.5 * static_cast<double>(scaleH);
    double left = std::max(u - halfX, 0.);
    double right = std::min(u + halfX, static_cast<double>(W - 1));
    double top = std::max(v - halfY, 0.);
    double bottom = std::min(v + halfY, static_cast<double>(H - 1));
    for (int i = static_cast<int>(top); i <= static_cast<int>(bottom); i++)
        for (int j = static_cast<int>(left); j <= static_cast<int>(right); j++) {
            int32_t idx = fr.pixel_map[static_cast<size_t>(j) + static_cast<size_t>(i) * W];
            if (idx != -1) out.push_back(idx);
        }
}

inline V3 cam_point_of_visible(const Frame& fr, int32_t visIdx) {  // NeighborFinderBase.cpp:15-27
    int64_t raw = fr.point_index[visIdx];
    return {fr.pts_cam[3 * raw], fr.pts_cam[3 * raw + 1], fr.pts_cam[3 * raw + 2]};
}

// DepthEstimator::CalculateNeighbors (DepthEstimator.cpp:636-684)
bool calculate_neighbors(const Frame& fr, double u, double v, float sx, float sy, std::vector<int32_t>& idx,
                         std::vector<V3>& nb) {
    get_neighbor_indices(fr, u, v, sx, sy, idx);
    for (int32_t i : idx) nb.push_back(cam_point_of_visible(fr, i));
    if (nb.size() < static_cast<unsigned int>(fr.P.radiusSearch_count_min)) return false;  // :680 (uint) cast
    return true;
}

// Histogram::AddElement (Histogram.cpp:19-33): the bin a value falls into
inline int histogram_bin_index(double value, double binW, int binCount) {
    value = std::min(value, 1e10);  // :29
    return static_cast<int>(std::min(std::abs(value / binW), static_cast<double>(binCount) - 1.));  // :30
}

// PointHistogram::FilterPointsMinDistBlob (HistogramPointDepth.cpp:15-123) + Histogram (Histogram.cpp:14-49).
// Returns false / true; out = positions (into the input list) of the kept points, in order.
bool filter_points_min_dist_blob(const double* depths, int n, double binW, int minimalMaximumSize,
                                 std::vector<int>& out, double& lowerBorder, double& higherBorder) {
    lowerBorder = -1;
    higherBorder = -1;
    int maxDist = 0;
    for (int i = 0; i < n; i++)
        if (depths[i] > maxDist) maxDist = static_cast<int>(std::ceil(depths[i]));  // :36-41
    int binCount = static_cast<int>((maxDist) / binW + 1);                           // :43
    if (binCount <= 1) return false;                                                // :53
    std::vector<int> bins(binCount, 0);
    for (int i = 0; i < n; i++) bins[histogram_bin_index(depths[i], binW, binCount)]++;
    int binMaxId = -1, binMaxVal = -1, binValue = 0;
    for (int i = 0; i < binCount; i++) {  // HistogramPointDepth.cpp:70-85
        float lastBinValue = static_cast<float>(binValue);
        binValue = bins[i];
        if ((binValue > binMaxVal) && (binValue >= minimalMaximumSize)) {
            binMaxVal = binValue;
            binMaxId = i;
        } else if (binValue < binMaxVal)
            break;
        if ((lastBinValue > 0) && (binValue == 0)) return false;
    }
    if (binMaxId < 0) return false;                    // :95
    lowerBorder = binMaxId * binW - 0.0f * binW;       // :99
    higherBorder = (binMaxId)*binW + 1.0f * binW;      // :100
    out.clear();
    for (int i = 0; i < n; i++)
        if ((depths[i] >= lowerBorder) && (depths[i] < higherBorder)) out.push_back(i);  // :115-120
    return true;
}

// PlaneEstimationCalcMaxSpanningTriangle::CalculatePlaneCorners (PlaneEstimationCalcMaxSpanningTriangle.cpp:37-100)
bool max_spanning_triangle(const std::vector<V3>& pts, double distThr, int& ci, int& cj, int& ck) {
    int n = static_cast<int>(pts.size());
    if (n < 3) return false;
    int mi = -1, mj = -1;
    double maxdist = -1;
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) {
            double d = sqnorm(sub(pts[i], pts[j]));
            if (d > maxdist) {
                maxdist = d;
                mi = i;
                mj = j;
            }
        }
    if (maxdist <= distThr) return false;
    double maxdist2 = -1;
    double mk = -1;
    for (int k = 0; k < n - 1; k++) {  // :71 last point never considered
        if (k == mi || k == mj) continue;
        double d1 = sqnorm(sub(pts[k], pts[mi]));
        if (d1 <= distThr) continue;
        double d2 = sqnorm(sub(pts[k], pts[mj]));
        if (d2 <= distThr) continue;
        double d = d1 + d2;
        if (d > maxdist2) {
            maxdist2 = d;
            mk = k;
        }
    }
    if (mi == -1 || mj == -1 || mk == -1) return false;
    ci = mi;
    cj = mj;
    ck = static_cast<int>(mk);
    return true;
}

// PlaneEstimationCheckPlanar::CheckPlanar (PlaneEstimationCheckPlanar.cpp:18-44)
bool check_planar(const V3& c1, const V3& c2, const V3& c3, double thr) {
    V3 e1 = normalized(sub(c2, c1)), e2 = normalized(sub(c3, c1)), e3 = normalized(sub(c3, c2));
    double l12 = norm(cross(e1, e2)), l13 = norm(cross(e1, e3)), l23 = norm(cross(e2, e3));
    return (l12 >= thr) && (l13 >= thr) && (l23 >= thr);
}

// CameraPinhole::getViewingRays<1> (camera_pinhole.h:52-69) + the flip at DepthEstimator.cpp:938-939.
V3 viewing_ray(const Frame& fr, double u, double v) {
    const double* Ki = fr.Kinv;
    // lhs * homogeneous(rhs) = lhs.leftCols(2) * rhs + lhs.col(2)
    V3 d = {(Ki[0] * u + Ki[1] * v) + Ki[2], (Ki[3] * u + Ki[4] * v) + Ki[5], (Ki[6] * u + Ki[7] * v) + Ki[8]};
    d = normalized(d);
    if (d.z < 0) d = scale(d, -1.0);
    return d;
}

// TresholdDepthGlobal::CheckInDepth (TresholdDepthGlobal.cpp:16-36): 0 InBounds, 1 SmallerMin, 2 GreaterMax
int threshold_global(const mld_params& P, double& depth) {
    double mn = P.treshold_depth_min, mx = P.treshold_depth_max;
    if (depth < mn) {
        if (P.treshold_depth_mode == 0) {
            depth = -1;
            return 1;
        }
        depth = mn;
    } else if (depth > mx) {
        if (P.treshold_depth_mode == 0) {
            depth = -1;
            return 2;
        }
        depth = mx;
    }
    return 0;
}

// TresholdDepthLocal::CheckInBounds (TresholdDepthLocal.cpp:18-66)
int threshold_local(const mld_params& P, const std::vector<V3>& pts, double& depth) {
    double minZ = std::numeric_limits<double>::max(), maxZ = std::numeric_limits<double>::lowest();
    for (const V3& p : pts) {
        if (p.z < minZ) minZ = p.z;
        if (p.z > maxZ) maxZ = p.z;
    }
    double interval = maxZ - minZ, lo, hi;
    if (P.treshold_depth_local_valuetype == 1) {
        double r = interval * P.treshold_depth_local_value;
        lo = minZ - r;
        hi = maxZ + r;
    } else {
        lo = minZ - P.treshold_depth_local_value;
        hi = maxZ + P.treshold_depth_local_value;
    }
    if (depth < lo) {
        if (P.treshold_depth_local_mode == 0) {
            depth = -1;
            return 1;
        }
        depth = lo;
    } else if (depth > hi) {
        if (P.treshold_depth_local_mode == 0) {
            depth = -1;
            return 2;
        }
        depth = hi;
    }
    return 0;
}

// Mono_LidarPipeline::PCA (PCA.cpp:11-62).  result: 0 Point, 1 Linear, 2 Plane, 3 Cubic
int pca_plane(const mld_params& P, const std::vector<V3>& pts, V3& normal, V3& mean) {
    size_t n = pts.size();
    // rowwise().mean(): sum in index order / n
    double sx = 0, sy = 0, sz = 0;
    for (const V3& p : pts) {
        sx += p.x;
        sy += p.y;
        sz += p.z;
    }
    mean = {sx / static_cast<double>(n), sy / static_cast<double>(n), sz / static_cast<double>(n)};
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (const V3& p : pts) {
        V3 d = sub(p, mean);
        c[0] += d.x * d.x;
        c[1] += d.x * d.y;
        c[2] += d.x * d.z;
        c[4] += d.y * d.y;
        c[5] += d.y * d.z;
        c[8] += d.z * d.z;
    }
    c[3] = c[1];
    c[6] = c[2];
    c[7] = c[5];
    double ev[3], evec[9];
    jacobi_eig3(c, ev, evec);
    V3 e0 = {evec[0], evec[3], evec[6]};
    normal = divs(e0, norm(e0));  // :56 eigenvector / norm
    // getResult (:21-40): classification in float
    double ev1 = ev[0], ev2 = ev[1], ev3 = ev[2];
    float planarity = static_cast<float>((ev2 - ev1) / ev3);
    float linearity = static_cast<float>((ev3 - ev2) / ev3);
    if (planarity < P.pca_treshold_2_1_rel_min) return 3;
    if (linearity > P.pca_treshold_3_2_rel_max) return 1;
    if (ev3 < P.pca_treshold_3_abs_min) return 0;
    return 2;
}

// LinePlaneIntersectionOrthogonalTreshold / ...Normal ::GetIntersection
// (LinePlaneIntersectionOrthogonalTreshold.cpp:16-48, LinePlaneIntersectionNormal.cpp:11-31)
bool intersect(const mld_params& P, bool useOrthThreshold, const Plane& pl, const V3& n0, const V3& n1, V3& pt,
               double& depth) {
    if (useOrthThreshold) {
        V3 lineNormal = normalized(n1);
        V3 planeNormal = normalized(pl.n);
        if (!(std::fabs(dot(planeNormal, lineNormal)) >= P.viewray_plane_orthoganality_treshold)) return false;
    }
    pt = line_through_intersection(n0, n1, pl);
    depth = pt.z;
    return true;
}

struct FeatureDebug {
    std::vector<int32_t> nb_idx;   // visible indices of the neighbours
    std::vector<int32_t> seg_pos;  // positions (into nb_idx) surviving the histogram
    int corner_pos[3] = {-1, -1, -1};  // positions into the segmented list
    std::vector<int32_t> road_idx;     // visible indices of the wide-window neighbours
    std::vector<int32_t> road_pos;     // positions (into road_idx) of the plane inliers
    double plane_n[3] = {0, 0, 0};
    double plane_offset = 0;
    int reached_road = 0;
};

// DepthEstimator::CalculateDepthSegmented (DepthEstimator.cpp:903-1037)
std::pair<int, double> calculate_depth_segmented(const Frame& fr, double u, double v, const std::vector<V3>& seg,
                                                 FeatureDebug* dbg) {
    const mld_params& P = fr.P;
    V3 c1{}, c2{}, c3{};
    if (!P.do_use_PCA && P.do_use_triangle_size_maximation) {
        int i, j, k;
        // _planeCalcMaxSpanning is built with the (bool) ctor -> _distTreshold = 0 (DepthEstimator.cpp:111-113)
        if (!max_spanning_triangle(seg, 0.0, i, j, k)) return {MLD_TriangleNotPlanarInsufficientPoints, -1.0};
        c1 = seg[i];
        c2 = seg[j];
        c3 = seg[k];
        if (dbg) {
            dbg->corner_pos[0] = i;
            dbg->corner_pos[1] = j;
            dbg->corner_pos[2] = k;
        }
    } else {
        if (seg.size() < 3) return {MLD_HistogramNoLocalMax, -1.0};  // :920-921
        c1 = seg[0];
        c2 = seg[1];
        c3 = seg[2];
        if (dbg) {
            dbg->corner_pos[0] = 0;
            dbg->corner_pos[1] = 1;
            dbg->corner_pos[2] = 2;
        }
    }
    if (!P.do_use_PCA && P.do_check_triangleplanar_condition) {
        if (!check_planar(c1, c2, c3, P.triangleplanar_crossnorm_treshold)) return {MLD_TriangleNotPlanar, -1.0};
    }
    V3 dir = viewing_ray(fr, u, v);
    V3 support = {0, 0, 0};
    bool orth = P.viewray_plane_orthoganality_treshold > 0;  // DepthEstimator.cpp:77-81
    V3 ip;
    double depth;
    if (P.do_use_PCA) {
        V3 normal, mean;
        int r = pca_plane(P, seg, normal, mean);
        if (r == 0) return {MLD_PcaIsPoint, -1.0};
        if (r == 1) return {MLD_PcaIsLine, -1.0};
        if (r == 3) return {MLD_PcaIsCubic, -1.0};
        Plane pl = {normal, -dot(normal, mean)};  // Hyperplane(normal, point)
        if (!intersect(P, orth, pl, support, dir, ip, depth)) return {MLD_PlaneViewrayNotOrthogonal, -1.0};
    } else {
        Plane pl = plane_through(c1, c2, c3);
        if (dbg) {
            dbg->plane_n[0] = pl.n.x;
            dbg->plane_n[1] = pl.n.y;
            dbg->plane_n[2] = pl.n.z;
            dbg->plane_offset = pl.offset;
        }
        if (!intersect(P, orth, pl, support, dir, ip, depth)) return {MLD_PlaneViewrayNotOrthogonal, -1.0};
    }
    if (P.treshold_depth_enabled) {
        int r = threshold_global(P, depth);
        if (r == 1) return {MLD_TresholdDepthGlobalSmallerMin, -1.0};
        if (r == 2) return {MLD_TresholdDepthGlobalGreaterMax, -1.0};
    }
    if (P.treshold_depth_local_enabled) {
        int r = threshold_local(P, seg, depth);
        if (r == 1) return {MLD_TresholdDepthLocalSmallerMin, -1.0};
        if (r == 2) return {MLD_TresholdDepthLocalGreaterMax, -1.0};
    }
    if (depth < 0 && P.do_use_cut_behind_camera) return {MLD_CornerBehindCamera, -1.0};
    return {MLD_Success, depth};
}

// PlaneEstimationMEstimator::EstimatePlane (PlaneEstimationMEstimator.cpp:18-55)
Plane mestimator_plane(const std::vector<V3>& pts, const Plane& prior) {
    size_t n = pts.size();
    std::vector<double> w(n);
    V3 center = {0, 0, 0};
    double wsum = 0;
    for (size_t i = 0; i < n; i++) {
        w[i] = 1 / std::fabs(dot(prior.n, pts[i]) + prior.offset);  // absDistance
        center = add(center, scale(pts[i], w[i]));
        wsum += w[i];
    }
    center = divs(center, wsum);
    // mat.col(i) = sqrt(w_i) * (p_i - center); JacobiSVD; normal = last column of U.
    // Restated as a one-sided (Hestenes) Jacobi SVD on the 3 rows of mat: rotate row pairs until
    // orthogonal; the accumulated rotations are U, the row norms the singular values.
    std::vector<double> r0(n), r1(n), r2(n);
    for (size_t i = 0; i < n; i++) {
        double ws = std::sqrt(w[i]);
        V3 d = sub(pts[i], center);
        r0[i] = ws * d.x;
        r1[i] = ws * d.y;
        r2[i] = ws * d.z;
    }
    double* rows[3] = {r0.data(), r1.data(), r2.data()};
    double U[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double app = 0, aqq = 0, apq = 0;
                for (size_t i = 0; i < n; i++) {
                    app += rows[p][i] * rows[p][i];
                    aqq += rows[q][i] * rows[q][i];
                    apq += rows[p][i] * rows[q][i];
                }
                if (!(std::fabs(apq) > 1e-15 * std::sqrt(app * aqq)) || apq == 0.0) continue;
                rotated = true;
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (size_t i = 0; i < n; i++) {
                    double a = rows[p][i], b = rows[q][i];
                    rows[p][i] = c * a - s * b;
                    rows[q][i] = s * a + c * b;
                }
                for (int k = 0; k < 3; k++) {
                    double a = U[k * 3 + p], b = U[k * 3 + q];
                    U[k * 3 + p] = c * a - s * b;
                    U[k * 3 + q] = s * a + c * b;
                }
            }
        if (!rotated) break;
    }
    double nrm[3];
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (size_t i = 0; i < n; i++) s += rows[r][i] * rows[r][i];
        nrm[r] = s;
    }
    int m = 0;
    // NaN-propagating pick of the smallest (all-NaN input keeps column 0 — result is NaN anyway)
    if (nrm[1] < nrm[m]) m = 1;
    if (nrm[2] < nrm[m]) m = 2;
    V3 normal = {U[0 * 3 + m], U[1 * 3 + m], U[2 * 3 + m]};
    bool any_nan = std::isnan(center.x) || std::isnan(center.y) || std::isnan(center.z);
    if (any_nan) normal = {std::nan(""), std::nan(""), std::nan("")};
    normal = normalized(normal);
    Plane r;
    r.n = normal;
    r.offset = -dot(normal, center);  // Hyperplane(n, e)
    return r;
}

// LinePlaneIntersectionCeckXZTreshold::Check (LinePlaneIntersectionCeckXZTreshold.cpp:15-45)
bool check_xz(const std::vector<V3>& pts, double thr) {
    double minX = std::numeric_limits<double>::max(), maxX = std::numeric_limits<double>::lowest();
    double minZ = minX, maxZ = maxX;
    for (const V3& p : pts) {
        if (p.x < minX) minX = p.x;
        if (p.x > maxX) maxX = p.x;
        if (p.z < minZ) minZ = p.z;
        if (p.z > maxZ) maxZ = p.z;
    }
    double relation = (maxZ - minZ) / (maxX - minX);
    return relation >= thr;
}

// DepthEstimator::CalculateDepthSegmentationPlane (DepthEstimator.cpp:782-900)
bool segmentation_plane(const Frame& fr, const std::vector<V3>& nb, const std::vector<int32_t>& idx,
                        std::vector<V3>& seg, FeatureDebug* dbg) {
    seg.clear();
    const double thr = fr.P.ransac_plane_point_distance_treshold;
    const double* Ti = fr.Tinv;
    for (size_t i = 0; i < nb.size(); i++) {
        int raw = fr.point_index[idx[i]];  // :808
        // :810 Affine3d * Vector3d : translation + (row . v) with the fixed-size reduction order
        const V3& p = nb[i];
        double xl = Ti[3] + (Ti[0] * p.x + (Ti[1] * p.y + Ti[2] * p.z));
        double yl = Ti[7] + (Ti[4] * p.x + (Ti[5] * p.y + Ti[6] * p.z));
        double zl = Ti[11] + (Ti[8] * p.x + (Ti[9] * p.y + Ti[10] * p.z));
        // :811 pcl::PointXYZ(float), :812 pcl::pointToPlaneDistance in float, un-normalised
        float xf = static_cast<float>(xl), yf = static_cast<float>(yl), zf = static_cast<float>(zl);
        float d = std::fabs(fr.coeffs[0] * xf + fr.coeffs[1] * yf + fr.coeffs[2] * zf + fr.coeffs[3]);
        double distance = d;
        if (distance > thr) {              // :814-815
            if (dbg) dbg->road_pos.clear();
            return false;
        }
        if (fr.inlier[raw]) {              // :817 CheckPointInPlane
            seg.push_back(nb[i]);
            if (dbg) dbg->road_pos.push_back(static_cast<int32_t>(i));
        }
    }
    if (seg.size() < 3) return false;  // :827-829
    return true;                       // :831-899 have no effect
}

// RoadDepthEstimatorMEstimator::CalculateDepth (RoadDepthEstimatorMEstimator.cpp:28-74) and
// RoadDepthEstimatorMaxSpanningTriangle::CalculateDepth (RoadDepthEstimatorMaxSpanningTriangle.cpp:24-75)
std::pair<int, double> road_depth(const Frame& fr, double u, double v, const std::vector<V3>& planePts) {
    const mld_params& P = fr.P;
    V3 dir = viewing_ray(fr, u, v);
    V3 support = {0, 0, 0};
    V3 ip;
    double depth;
    if (P.plane_estimator_use_triangle_maximation) {
        int i, j, k;
        if (!max_spanning_triangle(planePts, 0.0, i, j, k)) return {MLD_RadiusSearchInsufficientPoints, -1.0};
        if (!check_xz(planePts, P.plane_estimator_z_x_min_relation)) return {MLD_InsufficientRoadPoints, -1.0};
        Plane pl = plane_through(planePts[i], planePts[j], planePts[k]);
        intersect(P, false, pl, dir, support, ip, depth);  // arguments swapped as in the reference (:52-53)
    } else {
        Plane pl = mestimator_plane(planePts, fr.prior);
        intersect(P, false, pl, dir, support, ip, depth);  // RoadDepthEstimatorMEstimator.cpp:52-53
    }
    if (P.treshold_depth_enabled) {
        int r = threshold_global(P, depth);
        if (r == 1) return {MLD_TresholdDepthGlobalSmallerMin, -1.0};
        if (r == 2) return {MLD_TresholdDepthGlobalGreaterMax, -1.0};
    }
    if (P.treshold_depth_local_enabled) {
        int r = threshold_local(P, planePts, depth);
        if (r == 1) return {MLD_TresholdDepthLocalSmallerMin, -1.0};
        if (r == 2) return {MLD_TresholdDepthLocalGreaterMax, -1.0};
    }
    return {MLD_SuccessRoad, depth};
}

// DepthEstimator::CalculateDepth(Vector2d, gp, stats) (DepthEstimator.cpp:491-600)
std::pair<int, double> calculate_depth_feature(const Frame& fr, double u, double v, FeatureDebug* dbg) {
    const mld_params& P = fr.P;
    std::vector<int32_t> idx;
    std::vector<V3> nb;
    std::pair<int, double> result = {MLD_Unspecified, -1.0};
    bool ok = calculate_neighbors(fr, u, v, 1.0f, 1.0f, idx, nb);
    if (dbg) dbg->nb_idx = idx;
    if (!ok) return {MLD_RadiusSearchInsufficientPoints, -1.0};
    // :513-558 region growing: _lidarRowSegmenter is null unless do_use_depth_segmentation (rejected at create)

    // CalculateDepthSegmentation (:726-780)
    std::vector<V3> seg;
    bool histOk = true;
    if (P.do_use_histogram_segmentation) {
        std::vector<double> depths(nb.size());
        for (size_t i = 0; i < nb.size(); i++) depths[i] = std::min(nb[i].z, 999.);
        std::vector<int> keep;
        double lo, hi;
        histOk = filter_points_min_dist_blob(depths.data(), static_cast<int>(nb.size()),
                                             P.histogram_segmentation_bin_witdh,
                                             P.histogram_segmentation_min_pointcount, keep, lo, hi);
        if (histOk) {
            for (int k : keep) seg.push_back(nb[k]);
            if (dbg) dbg->seg_pos.assign(keep.begin(), keep.end());
        }
    } else {
        seg = nb;
        if (dbg)
            for (size_t i = 0; i < nb.size(); i++) dbg->seg_pos.push_back(static_cast<int32_t>(i));
    }
    if (!histOk) result = {MLD_HistogramNoLocalMax, -1.0};
    if (result.first != MLD_HistogramNoLocalMax) {
        result = calculate_depth_segmented(fr, u, v, seg, dbg);
        if (result.first == MLD_Success) return result;
    }
    // road fallback (:578-597)
    int resultOld = result.first;
    if (fr.has_plane && P.do_use_ransac_plane) {
        if (dbg) dbg->reached_road = 1;
        idx.clear();
        nb.clear();
        bool ok2 = calculate_neighbors(fr, u, v, 2.0f, 1.5f, idx, nb);
        if (dbg) dbg->road_idx = idx;
        if (!ok2) return {MLD_RadiusSearchInsufficientPoints, -1.0};
        if (!segmentation_plane(fr, nb, idx, seg, dbg)) return {resultOld, -1.0};
        result = road_depth(fr, u, v, seg);
    }
    return result;
}

int validate_params(const mld_params& P) {
    if (P.neighbor_search_mode != 0) return MLD_ERR_UNSUPPORTED_MODE;
    if (P.do_use_depth_segmentation) return MLD_ERR_UNSUPPORTED_MODE;
    if (P.do_use_ransac_plane) {
        if (P.plane_estimator_use_triangle_maximation) {
        } else if (P.plane_estimator_use_leastsquares)
            return MLD_ERR_UNSUPPORTED_MODE;
        else if (P.plane_estimator_use_mestimator) {
        } else
            return MLD_ERR_NO_ROAD_ESTIMATOR;
    }
    return MLD_OK;
}

}  // namespace

extern "C" {

struct orc_frame {
    Frame fr;
};

orc_frame* orc_create(const mld_params* params, const mld_camera* cam, const double T_cam_lidar[12], int* status) {
    int st = validate_params(*params);
    if (status) *status = st;
    if (st != MLD_OK) return nullptr;
    orc_frame* h = new orc_frame();
    h->fr.P = *params;
    h->fr.cam = *cam;
    std::memcpy(h->fr.T, T_cam_lidar, sizeof(double) * 12);
    make_calibration(h->fr);
    return h;
}

void orc_destroy(orc_frame* h) { delete h; }

void orc_get_calibration(const orc_frame* h, double Tinv[12], double Kinv[9]) {
    std::memcpy(Tinv, h->fr.Tinv, sizeof(double) * 12);
    std::memcpy(Kinv, h->fr.Kinv, sizeof(double) * 9);
}

// setInputCloud minus the ground-plane hook (DepthEstimator.cpp:220-271)
int orc_set_cloud(orc_frame* h, const void* pts, int64_t n, int stride_bytes) {
    if (!h || (!pts && n > 0) || n < 0 || (stride_bytes != 16 && stride_bytes != 32)) return MLD_ERR_INVALID_ARG;
    transform_cloud(h->fr, static_cast<const uint8_t*>(pts), n, stride_bytes);
    init_lidar_projection(h->fr);
    h->fr.cloud_set = true;
    h->fr.has_plane = false;
    return MLD_OK;
}

// Ground-plane hook (DepthEstimator.cpp:273-292) with the plane as an input object.
int orc_set_ground_plane(orc_frame* h, const float coeffs[4], const int32_t* inliers, int64_t n_inl) {
    if (!h || !h->fr.cloud_set) return MLD_ERR_NOT_INITIALIZED;
    Frame& fr = h->fr;
    if (!coeffs) {
        fr.has_plane = false;
        return MLD_OK;
    }
    std::memcpy(fr.coeffs, coeffs, sizeof(float) * 4);
    fr.inlier.assign(fr.n, 0);
    for (int64_t i = 0; i < n_inl; i++)
        if (inliers[i] >= 0 && inliers[i] < fr.n) fr.inlier[inliers[i]] = 1;
    // :289-291 prior = Hyperplane(normalize(a,b,c), d)
    V3 nrm = {static_cast<double>(coeffs[0]), static_cast<double>(coeffs[1]), static_cast<double>(coeffs[2])};
    fr.prior.n = normalized(nrm);
    fr.prior.offset = static_cast<double>(coeffs[3]);
    fr.has_plane = true;
    return MLD_OK;
}

// CalculateDepth(Matrix2Xd, VectorXd&, VectorXi&, gp) (DepthEstimator.cpp:429-488)
int orc_calculate_depth(orc_frame* h, const double* uv, int64_t F, double* depth, int32_t* type, int n_threads) {
    if (!h || !h->fr.cloud_set) return MLD_ERR_NOT_INITIALIZED;
    const Frame& fr = h->fr;
    if (fr.P.set_all_depths_to_zero) {  // :448-453
        for (int64_t i = 0; i < F; i++) {
            depth[i] = -1;
            if (type) type[i] = 1;
        }
        return MLD_OK;
    }
    (void)n_threads;
#pragma omp parallel for num_threads(n_threads > 0 ? n_threads : 1) schedule(static)
    for (int64_t i = 0; i < F; i++) {
        auto r = calculate_depth_feature(fr, uv[2 * i], uv[2 * i + 1], nullptr);
        depth[i] = r.second;
        if (type) type[i] = r.first;
    }
    return MLD_OK;
}

int64_t orc_num_points(const orc_frame* h) { return h->fr.n; }
int64_t orc_visible_count(const orc_frame* h) { return static_cast<int64_t>(h->fr.point_index.size()); }
void orc_get_visible_image_points(const orc_frame* h, double* out) {
    std::memcpy(out, h->fr.pts_img_vis.data(), h->fr.pts_img_vis.size() * sizeof(double));
}
void orc_get_point_index(const orc_frame* h, int32_t* out) {
    std::memcpy(out, h->fr.point_index.data(), h->fr.point_index.size() * sizeof(int32_t));
}
void orc_get_cloud_camera_cs(const orc_frame* h, double* out) {
    std::memcpy(out, h->fr.pts_cam.data(), h->fr.pts_cam.size() * sizeof(double));
}
void orc_get_cloud_image_cs(const orc_frame* h, double* out) {
    std::memcpy(out, h->fr.pts_img.data(), h->fr.pts_img.size() * sizeof(double));
}
void orc_get_in_range(const orc_frame* h, uint8_t* out) {
    std::memcpy(out, h->fr.in_range.data(), h->fr.in_range.size());
}
void orc_get_pixel_map(const orc_frame* h, int32_t* out) {
    std::memcpy(out, h->fr.pixel_map.data(), h->fr.pixel_map.size() * sizeof(int32_t));
}

// Per-feature trace for golden fixtures.  Arrays are caller-allocated with capacity `cap` each.
typedef struct orc_feature_trace {
    int32_t type;
    int32_t reached_road;
    double depth;
    int32_t n_nb, n_seg, n_road, n_road_inl;
    int32_t corner_pos[3];
    int32_t pad_;
    double plane_n[3];
    double plane_offset;
} orc_feature_trace;

int orc_trace_feature(orc_frame* h, double u, double v, orc_feature_trace* tr, int32_t* nb_idx, int32_t* seg_pos,
                      int32_t* road_idx, int32_t* road_pos, int32_t cap) {
    if (!h || !h->fr.cloud_set) return MLD_ERR_NOT_INITIALIZED;
    FeatureDebug dbg;
    auto r = calculate_depth_feature(h->fr, u, v, &dbg);
    tr->type = r.first;
    tr->depth = r.second;
    tr->reached_road = dbg.reached_road;
    tr->n_nb = static_cast<int32_t>(dbg.nb_idx.size());
    tr->n_seg = static_cast<int32_t>(dbg.seg_pos.size());
    tr->n_road = static_cast<int32_t>(dbg.road_idx.size());
    tr->n_road_inl = static_cast<int32_t>(dbg.road_pos.size());
    for (int i = 0; i < 3; i++) {
        tr->corner_pos[i] = dbg.corner_pos[i];
        tr->plane_n[i] = dbg.plane_n[i];
    }
    tr->plane_offset = dbg.plane_offset;
    if (tr->n_nb > cap || tr->n_seg > cap || tr->n_road > cap || tr->n_road_inl > cap) return MLD_ERR_CAPACITY;
    std::copy(dbg.nb_idx.begin(), dbg.nb_idx.end(), nb_idx);
    std::copy(dbg.seg_pos.begin(), dbg.seg_pos.end(), seg_pos);
    std::copy(dbg.road_idx.begin(), dbg.road_idx.end(), road_idx);
    std::copy(dbg.road_pos.begin(), dbg.road_pos.end(), road_pos);
    return MLD_OK;
}

// TrackletDepthModule::process, the feature marshalling either side of the path
// (tracklets_depth/src/tracklet_depth_module.cpp): ExractNewTrackletFrames :23-61 (a new track contributes its
// newest AND its previous feature; std::pair<int,int> truncates the float coordinates), CalculateFeatureDepths
// CurFrame :63-82 / LastFrame :84-117 (no previous cloud -> -1), SaveFeatureDepths :119-169 (float32 depths).
int orc_tracklets_depth(orc_frame* cur, orc_frame* last, const float* u_new, const float* v_new, const float* u_old,
                        const float* v_old, const uint8_t* is_new, int64_t n, float* d_cur, float* d_last,
                        int32_t* t_cur, int32_t* t_last, int n_threads) {
    if (!cur || !cur->fr.cloud_set) return MLD_ERR_NOT_INITIALIZED;
    std::vector<double> uv_cur(2 * n), uv_last;
    std::vector<int64_t> owner;
    for (int64_t i = 0; i < n; i++) {
        uv_cur[2 * i] = static_cast<double>(static_cast<int>(u_new[i]));
        uv_cur[2 * i + 1] = static_cast<double>(static_cast<int>(v_new[i]));
        if (is_new[i]) {
            uv_last.push_back(static_cast<double>(static_cast<int>(u_old[i])));
            uv_last.push_back(static_cast<double>(static_cast<int>(v_old[i])));
            owner.push_back(i);
        }
    }
    std::vector<double> dc(n), dl(owner.size(), -1.0);
    std::vector<int32_t> tc(n), tl(owner.size(), 0);
    int rc = orc_calculate_depth(cur, uv_cur.data(), n, dc.data(), tc.data(), n_threads);
    if (rc) return rc;
    if (last && last->fr.cloud_set && !owner.empty()) {
        rc = orc_calculate_depth(last, uv_last.data(), static_cast<int64_t>(owner.size()), dl.data(), tl.data(), n_threads);
        if (rc) return rc;
    }
    for (int64_t i = 0; i < n; i++) {
        d_cur[i] = static_cast<float>(dc[i]);
        if (t_cur) t_cur[i] = tc[i];
    }
    for (size_t r = 0; r < owner.size(); r++) {
        d_last[owner[r]] = static_cast<float>(dl[r]);
        if (t_last) t_last[owner[r]] = tl[r];
    }
    return MLD_OK;
}

// ------------------------------------------------------------------------------------------------
// RansacPlane::CalculateInliersPlane (monolidar_fusion/src/RansacPlane.cpp:41-140), restated.
//
// PARITY UNPINNED: the arithmetic lives in un-vendored PCL (RandomSample is time-seeded upstream, so the
// reference is not even run-to-run reproducible).  What is kept: the pipeline (z pass-through only if
// min_z > -1001 :57-64, sub-sample to 6000 :66-74, perpendicular-plane RANSAC with axis z / 10 degrees
// :94-108 incl. the adaptive iteration bound k = log(1-p)/log(1-w^3), optional least-squares refinement and the
// re-selection within `ransac_plane_refinement_treshold` of the UNREFINED model :117-126), float arithmetic as in
// PCL's plane model.  What is ours (shared bit-for-bit with the HIP implementation): the random draws come from a
// counter-based hash, the 6000-point sub-sample is stratified instead of Knuth's algorithm S, the refinement's
// float sums use 256 interleaved partial sums, the smallest eigenvector comes from a double Jacobi solve.
// Checked against the reference's own test RansacPlane.CalculateInlersPlane (coefficients within 0.2).
// ------------------------------------------------------------------------------------------------
namespace ransac {
constexpr int kSample = 6000;  // RansacPlane.cpp:32 _numberRandomSamplePoints
constexpr int kPartials = 256;

inline uint32_t mix(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u;
    h ^= b + 0x85EBCA6Bu + (h << 6) + (h >> 2);
    h ^= c * 0xC2B2AE35u + (h << 6) + (h >> 2);
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

struct Model {
    float c[4];
    bool degenerate;  // collinear sample: computeModelCoefficients fails, the draw is skipped
    bool valid;       // normal within eps_angle of the z axis (isModelValid)
};

inline Model plane_from(const float* p0, const float* p1, const float* p2) {
    Model m{};
    float a0 = p1[0] - p0[0], a1 = p1[1] - p0[1], a2 = p1[2] - p0[2];
    float b0 = p2[0] - p0[0], b1 = p2[1] - p0[1], b2 = p2[2] - p0[2];
    float r0 = a0 / b0, r1 = a1 / b1, r2 = a2 / b2;  // sac_model_plane: collinearity test on the ratios
    m.degenerate = (r0 == r1) && (r2 == r1);
    float n0 = a1 * b2 - a2 * b1, n1 = a2 * b0 - a0 * b2, n2 = a0 * b1 - a1 * b0;
    float nn = std::sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    n0 /= nn;
    n1 /= nn;
    n2 /= nn;
    m.c[0] = n0;
    m.c[1] = n1;
    m.c[2] = n2;
    m.c[3] = -1.0f * (n0 * p0[0] + n1 * p0[1] + n2 * p0[2]);
    if (!(nn > 0.0f) || !std::isfinite(nn)) m.degenerate = true;
    m.valid = !m.degenerate && (std::fabs(static_cast<double>(n2)) >= 0.984807753012208);  // cos(pi/18)
    return m;
}
inline float plane_dist(const float c[4], const float* p) { return std::fabs(c[0] * p[0] + c[1] * p[1] + c[2] * p[2] + c[3]); }
}  // namespace ransac

int orc_estimate_ground_plane(orc_frame* h, const void* pts_v, int64_t n, int stride, uint32_t seed, float coeffs_out[4],
                              int64_t* n_inliers_out) {
    using namespace ransac;
    if (!h || !h->fr.cloud_set) return MLD_ERR_NOT_INITIALIZED;
    const mld_params& P = h->fr.P;
    const uint8_t* pts = static_cast<const uint8_t*>(pts_v);
    if (n < 3) return MLD_ERR_CLOUD_TOO_SMALL;  // RansacPlane.cpp:44-50
    auto pt = [&](int64_t i) { return reinterpret_cast<const float*>(pts + i * stride); };
    // :57-64 PassThrough on z (float limits), only if min_z > -1001
    std::vector<int32_t> cand;
    const bool pass = P.ransac_plane_min_z > -1001.;
    if (pass) {
        const float lo = static_cast<float>(P.ransac_plane_min_z), hi = static_cast<float>(P.ransac_plane_max_z);
        for (int64_t i = 0; i < n; i++) {
            const float* p = pt(i);
            if (std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]) && !(p[2] < lo) && !(p[2] > hi))
                cand.push_back(static_cast<int32_t>(i));
        }
    }
    const int64_t M = pass ? static_cast<int64_t>(cand.size()) : n;
    // :66-74 RandomSample to 6000 (stratified restatement; ascending like Algorithm S)
    std::vector<int32_t> sample;
    if (M > kSample) {
        sample.resize(kSample);
        for (int j = 0; j < kSample; j++) {
            double u = static_cast<double>(mix(seed, static_cast<uint32_t>(j), 0x5A17u)) * (1.0 / 4294967296.0);
            int64_t pos = static_cast<int64_t>((static_cast<double>(j) + u) * static_cast<double>(M) / static_cast<double>(kSample));
            if (pos > M - 1) pos = M - 1;
            sample[j] = pass ? cand[pos] : static_cast<int32_t>(pos);
        }
    } else {
        sample.resize(M);
        for (int64_t j = 0; j < M; j++) sample[j] = pass ? cand[j] : static_cast<int32_t>(j);
    }
    const int S = static_cast<int>(sample.size());
    if (S < 3) return MLD_ERR_CLOUD_TOO_SMALL;
    std::vector<float> sp(3 * static_cast<size_t>(S));
    for (int j = 0; j < S; j++) {
        const float* p = pt(sample[j]);
        sp[3 * j] = p[0];
        sp[3 * j + 1] = p[1];
        sp[3 * j + 2] = p[2];
    }
    // :102-108 RandomSampleConsensus
    const double thr = P.ransac_plane_distance_treshold;
    const int max_it = P.ransac_plane_max_iterations;
    auto draw = [&](int d) {
        uint32_t a = mix(seed, static_cast<uint32_t>(d), 1u) % static_cast<uint32_t>(S);
        uint32_t b = mix(seed, static_cast<uint32_t>(d), 2u) % static_cast<uint32_t>(S);
        uint32_t c = mix(seed, static_cast<uint32_t>(d), 3u) % static_cast<uint32_t>(S);
        if (b == a) b = (b + 1) % static_cast<uint32_t>(S);
        while (c == a || c == b) c = (c + 1) % static_cast<uint32_t>(S);
        return plane_from(&sp[3 * a], &sp[3 * b], &sp[3 * c]);
    };
    auto count_inliers = [&](const Model& m) {
        if (!m.valid) return 0;
        int cnt = 0;
        for (int j = 0; j < S; j++)
            if (static_cast<double>(plane_dist(m.c, &sp[3 * j])) < thr) cnt++;
        return cnt;
    };
    int iterations = 0, best = -2147483647, best_draw = -1;
    double k = 1.0;
    const double log_probability = std::log(1.0 - P.ransac_plane_probability);
    const double one_over = 1.0 / static_cast<double>(S);
    const int max_draws = max_it + 1;  // one hypothesis per draw; skipped (degenerate) draws are not re-drawn
    for (int d = 0; d < max_draws && iterations < k; d++) {
        Model m = draw(d);
        if (m.degenerate) continue;  // ++skipped_count
        int cnt = count_inliers(m);
        if (cnt > best) {
            best = cnt;
            best_draw = d;
            double w = static_cast<double>(best) * one_over;
            double p_no = 1.0 - w * w * w;
            p_no = std::max(std::numeric_limits<double>::epsilon(), p_no);
            p_no = std::min(1.0 - std::numeric_limits<double>::epsilon(), p_no);
            k = log_probability / std::log(p_no);
        }
        ++iterations;
        if (iterations > max_it) break;
    }
    if (best_draw < 0) return MLD_ERR_CLOUD_TOO_SMALL;
    Model bm = draw(best_draw);
    float coeffs[4] = {bm.c[0], bm.c[1], bm.c[2], bm.c[3]};
    std::vector<int32_t> inl;  // positions in the sample
    if (bm.valid)
        for (int j = 0; j < S; j++)
            if (static_cast<double>(plane_dist(bm.c, &sp[3 * j])) < thr) inl.push_back(j);
    // :117-126 refinement
    if (P.ransac_plane_use_refinement) {
        if (inl.size() > 3) {
            // computeMeanAndCovarianceMatrix, float accumulators; partial p takes the inliers q = p, p+256, ...
            float acc[kPartials][9];
            for (int p = 0; p < kPartials; p++) {
                for (int t = 0; t < 9; t++) acc[p][t] = 0.0f;
                for (size_t q = p; q < inl.size(); q += kPartials) {
                    const float* v = &sp[3 * inl[q]];
                    acc[p][0] += v[0] * v[0];
                    acc[p][1] += v[0] * v[1];
                    acc[p][2] += v[0] * v[2];
                    acc[p][3] += v[1] * v[1];
                    acc[p][4] += v[1] * v[2];
                    acc[p][5] += v[2] * v[2];
                    acc[p][6] += v[0];
                    acc[p][7] += v[1];
                    acc[p][8] += v[2];
                }
            }
            float a[9];
            for (int t = 0; t < 9; t++) {
                a[t] = 0.0f;
                for (int p = 0; p < kPartials; p++) a[t] += acc[p][t];
                a[t] /= static_cast<float>(inl.size());
            }
            float cov[6] = {a[0] - a[6] * a[6], a[1] - a[6] * a[7], a[2] - a[6] * a[8],
                            a[3] - a[7] * a[7], a[4] - a[7] * a[8], a[5] - a[8] * a[8]};
            double m9[9] = {cov[0], cov[1], cov[2], cov[1], cov[3], cov[4], cov[2], cov[4], cov[5]};
            double ev[3], evec[9];
            jacobi_eig3(m9, ev, evec);
            float e0 = static_cast<float>(evec[0]), e1 = static_cast<float>(evec[3]), e2 = static_cast<float>(evec[6]);
            coeffs[0] = e0;
            coeffs[1] = e1;
            coeffs[2] = e2;
            coeffs[3] = -1.0f * (e0 * a[6] + e1 * a[7] + e2 * a[8]);
        }
        // selectWithinDistance(UNREFINED model, refinement threshold)
        inl.clear();
        if (bm.valid)
            for (int j = 0; j < S; j++)
                if (static_cast<double>(plane_dist(bm.c, &sp[3 * j])) < P.ransac_plane_refinement_treshold) inl.push_back(j);
    }
    std::vector<int32_t> inlier_idx(inl.size());
    for (size_t q = 0; q < inl.size(); q++) inlier_idx[q] = sample[inl[q]];
    for (int t = 0; t < 4; t++) coeffs_out[t] = coeffs[t];
    if (n_inliers_out) *n_inliers_out = static_cast<int64_t>(inlier_idx.size());
    return orc_set_ground_plane(h, coeffs, inlier_idx.data(), static_cast<int64_t>(inlier_idx.size()));
}

// SampleConsensusModelPlane::optimizeModelCoefficients (PCL sac_model_plane.hpp, restated in SURVEY.md §9) over the
// cloud points whose `member` flag is set: computeMeanAndCovarianceMatrix with float accumulators, smallest eigenvector of
// the covariance, d = -n.centroid.  Fewer than 4 members: `fallback` is returned.  PCL adds the members one after the
// other; the association here is the one the HIP kernels use (mld_ransac.hip, k_sem_candidates / k_sem_select /
// k_sem_fit; parity with PCL's order is unpinned either way): the nine moment terms of the members of every GROUP of 64
// consecutive cloud points are summed by a fixed binary tree over the 64 positions (non-members and positions beyond the
// cloud contribute +0.0f; strides 32, 16, ..., 1), the group sums are added, in group order, into 256 interleaved
// partials (group g -> partial g % 256), and the partials are combined in index order.
static void ls_plane_fit(const uint8_t* pts, int stride, int64_t n, const std::vector<uint8_t>& member, const float fallback[4],
                         float out[4], size_t* m_out = nullptr) {
    using namespace ransac;
    for (int t = 0; t < 4; t++) out[t] = fallback[t];
    std::vector<float> acc(static_cast<size_t>(kPartials) * 9, 0.0f);
    size_t m = 0;
    const int64_t G = (n + 63) / 64;
    for (int64_t g = 0; g < G; g++) {
        float tree[64][9];
        for (int l = 0; l < 64; l++) {
            const int64_t i = g * 64 + l;
            const bool in = i < n && member[static_cast<size_t>(i)];
            float x = 0.f, y = 0.f, z = 0.f;
            if (in) {
                const float* v = reinterpret_cast<const float*>(pts + i * stride);
                x = v[0];
                y = v[1];
                z = v[2];
                m++;
            }
            float* a = tree[l];
            a[0] = in ? x * x : 0.0f;
            a[1] = in ? x * y : 0.0f;
            a[2] = in ? x * z : 0.0f;
            a[3] = in ? y * y : 0.0f;
            a[4] = in ? y * z : 0.0f;
            a[5] = in ? z * z : 0.0f;
            a[6] = in ? x : 0.0f;
            a[7] = in ? y : 0.0f;
            a[8] = in ? z : 0.0f;
        }
        for (int o = 32; o > 0; o >>= 1)
            for (int l = 0; l < o; l++)
                for (int t = 0; t < 9; t++) tree[l][t] = tree[l][t] + tree[l + o][t];
        float* a = &acc[static_cast<size_t>(g % kPartials) * 9];
        for (int t = 0; t < 9; t++) a[t] += tree[0][t];
    }
    if (m_out) *m_out = m;
    if (m < 4) return;
    float a[9];
    for (int t = 0; t < 9; t++) {
        a[t] = 0.0f;
        for (int p = 0; p < kPartials; p++) a[t] += acc[static_cast<size_t>(p) * 9 + t];
        a[t] /= static_cast<float>(m);
    }
    float cov[6] = {a[0] - a[6] * a[6], a[1] - a[6] * a[7], a[2] - a[6] * a[8],
                    a[3] - a[7] * a[7], a[4] - a[7] * a[8], a[5] - a[8] * a[8]};
    double m9[9] = {cov[0], cov[1], cov[2], cov[1], cov[3], cov[4], cov[2], cov[4], cov[5]};
    double ev[3], evec[9];
    jacobi_eig3(m9, ev, evec);
    float e0 = static_cast<float>(evec[0]), e1 = static_cast<float>(evec[3]), e2 = static_cast<float>(evec[6]);
    out[0] = e0;
    out[1] = e1;
    out[2] = e2;
    out[3] = -1.0f * (e0 * a[6] + e1 * a[7] + e2 * a[8]);
}

// SemanticPlane::CalculateInliersPlane (monolidar_fusion/src/RansacPlane.cpp:195-274) with the estimator's own
// calibration as SemanticPlane::Camera (tracklets_depth/src/tracklet_depth_module.cpp:273-284 builds it from the same
// camera info and transform).  `img`: rows x cols uint8 labels, row_stride bytes per row.
//   :198-199  pcl::transformPointCloud (double arithmetic, float result) and project(): K * p (Eigen 3x3 * 3x1 in
//             double), p /= p[2], cv::Point(int) truncation
//   :202-221  keep the points whose pixel is inside [0,cols] x [0,rows] and carries a ground label.  The reference
//             reads image(x == cols) / image(y == rows) out of bounds; those pixels count as unlabeled here, and
//             non-finite or int-overflowing projections (UB in the reference) as invalid.
//   :224-227  fewer than 3 candidates: ExceptionPclInvalid
//   :238-246  least-squares fit to the candidates (dummy prior 0,0,1,0 returned for < 4 points)
//   :250-254  selectWithinDistance(all points, inlier_threshold) and the fit to them
//   :259-268  coefficients = refined fit, inliers = the selected points
int orc_estimate_semantic_plane(orc_frame* h, const void* pts_v, int64_t n, int stride, const uint8_t* img, int rows,
                                int cols, int row_stride, const int32_t* labels, int n_labels, double inlier_threshold,
                                float coeffs_out[4], int64_t* n_inliers_out) {
    if (!h || !h->fr.cloud_set) return MLD_ERR_NOT_INITIALIZED;
    const Frame& fr = h->fr;
    const uint8_t* pts = static_cast<const uint8_t*>(pts_v);
    const double* T = fr.T;
    const double f = fr.cam.focal_length, cu = fr.cam.principal_point_x, cv = fr.cam.principal_point_y;
    bool is_ground[256] = {false};
    for (int i = 0; i < n_labels; i++)
        if (labels[i] >= 0 && labels[i] < 256) is_ground[labels[i]] = true;
    std::vector<uint8_t> cand(static_cast<size_t>(n), 0);
    size_t n_cand = 0;
    for (int64_t i = 0; i < n; i++) {
        const float* q = reinterpret_cast<const float*>(pts + i * stride);
        const double x = q[0], y = q[1], z = q[2];
        const float xc = static_cast<float>(((T[0] * x + T[1] * y) + T[2] * z) + T[3]);
        const float yc = static_cast<float>(((T[4] * x + T[5] * y) + T[6] * z) + T[7]);
        const float zc = static_cast<float>(((T[8] * x + T[9] * y) + T[10] * z) + T[11]);
        const double px = static_cast<double>(xc), py = static_cast<double>(yc), pz = static_cast<double>(zc);
        const double p0 = f * px + (0.0 * py + cu * pz);
        const double p1 = 0.0 * px + (f * py + cv * pz);
        const double p2 = 0.0 * px + (0.0 * py + 1.0 * pz);
        const double u = p0 / p2, v = p1 / p2;
        if (!std::isfinite(u) || !std::isfinite(v) || std::fabs(u) >= 2147483648.0 || std::fabs(v) >= 2147483648.0) continue;
        const int ix = static_cast<int>(u), iy = static_cast<int>(v);
        if (ix < 0 || ix > cols || iy < 0 || iy > rows) continue;  // :206-207
        if (ix == cols || iy == rows) continue;                    // out-of-bounds read in the reference
        if (is_ground[img[static_cast<size_t>(iy) * row_stride + ix]]) {
            cand[static_cast<size_t>(i)] = 1;
            n_cand++;
        }
    }
    if (n_cand < 3) return MLD_ERR_CLOUD_TOO_SMALL;  // :224-227
    const float dummy[4] = {0.f, 0.f, 1.f, 0.f};
    float c1[4], c2[4];
    ls_plane_fit(pts, stride, n, cand, dummy, c1);
    std::vector<int32_t> inl;
    std::vector<uint8_t> sel(static_cast<size_t>(n), 0);
    for (int64_t i = 0; i < n; i++) {
        const float* q = reinterpret_cast<const float*>(pts + i * stride);
        if (static_cast<double>(ransac::plane_dist(c1, q)) < inlier_threshold) {
            inl.push_back(static_cast<int32_t>(i));
            sel[static_cast<size_t>(i)] = 1;
        }
    }
    ls_plane_fit(pts, stride, n, sel, c1, c2);
    for (int t = 0; t < 4; t++) coeffs_out[t] = c2[t];
    if (n_inliers_out) *n_inliers_out = static_cast<int64_t>(inl.size());
    return orc_set_ground_plane(h, c2, inl.data(), static_cast<int64_t>(inl.size()));
}

int64_t orc_get_plane_inliers(const orc_frame* h, int32_t* out, int64_t cap) {
    int64_t k = 0;
    for (int64_t i = 0; i < h->fr.n; i++)
        if (h->fr.has_plane && h->fr.inlier[i]) {
            if (k < cap) out[k] = static_cast<int32_t>(i);
            k++;
        }
    return k;
}

// ---- component entry points (known-answer tests, micro-vectors) --------------------------------

// PointHistogram::FilterPointsMinDistBlob: returns 1/0; keep[] = positions kept.
int orc_filter_points_min_dist_blob(const double* depths, int n, double binW, int minCount, int32_t* keep,
                                    int32_t* n_keep, double* lower, double* higher) {
    std::vector<int> k;
    bool ok = filter_points_min_dist_blob(depths, n, binW, minCount, k, *lower, *higher);
    *n_keep = ok ? static_cast<int32_t>(k.size()) : 0;
    if (ok) std::copy(k.begin(), k.end(), keep);
    return ok ? 1 : 0;
}

// PointHistogram::GetNearestPoint (HistogramPointDepth.cpp:125-150): returns the neighbour index stored at
// the minimum depth (float-typed running minimum as in the reference), -1 if none.
int orc_get_nearest_point(const double* depths, const int32_t* neighborsIndex, int n) {
    float minDepth = static_cast<float>(std::numeric_limits<double>::max());  // float minDepth = DBL_MAX -> +inf
    int minIndex = -1;
    for (int i = 0; i < n; i++)
        if (depths[i] < minDepth) {
            minDepth = static_cast<float>(depths[i]);
            minIndex = neighborsIndex[i];
        }
    return minIndex;
}

int orc_max_spanning_triangle(const double* pts /*3 x n col-major*/, int n, double thr, int32_t out[3]) {
    std::vector<V3> p(n);
    for (int i = 0; i < n; i++) p[i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    int i, j, k;
    if (!max_spanning_triangle(p, thr, i, j, k)) return 0;
    out[0] = i;
    out[1] = j;
    out[2] = k;
    return 1;
}

int orc_check_planar(const double c1[3], const double c2[3], const double c3[3], double thr) {
    return check_planar({c1[0], c1[1], c1[2]}, {c2[0], c2[1], c2[2]}, {c3[0], c3[1], c3[2]}, thr) ? 1 : 0;
}

void orc_viewing_ray(const orc_frame* h, double u, double v, double dir[3]) {
    V3 d = viewing_ray(h->fr, u, v);
    dir[0] = d.x;
    dir[1] = d.y;
    dir[2] = d.z;
}

// GetIntersectionPoint(p1,p2,p3,n0,n1,...) (LinePlaneIntersectionBase.cpp:33-44); orth_thr <= 0 -> Normal variant.
int orc_intersect_triangle(const double p1[3], const double p2[3], const double p3[3], const double n0[3],
                           const double n1[3], double orth_thr, double pt[3], double* depth) {
    mld_params P{};
    P.viewray_plane_orthoganality_treshold = orth_thr;
    Plane pl = plane_through({p1[0], p1[1], p1[2]}, {p2[0], p2[1], p2[2]}, {p3[0], p3[1], p3[2]});
    V3 ip{};
    bool ok = intersect(P, orth_thr > 0, pl, {n0[0], n0[1], n0[2]}, {n1[0], n1[1], n1[2]}, ip, *depth);
    pt[0] = ip.x;
    pt[1] = ip.y;
    pt[2] = ip.z;
    return ok ? 1 : 0;
}

int orc_threshold_global(const mld_params* P, double* depth) { return threshold_global(*P, *depth); }

// Histogram::AddElement for every value, then the bin sizes (Histogram.cpp:14-49) — the binning expression
// filter_points_min_dist_blob uses, exposed for the check against the reference's own Histogram class
void orc_histogram_counts(const double* values, int n, double binW, int binCount, int32_t* counts) {
    for (int b = 0; b < binCount; b++) counts[b] = 0;
    for (int i = 0; i < n; i++) counts[histogram_bin_index(values[i], binW, binCount)]++;
}

int orc_threshold_local(const mld_params* P, const double* pts, int n, double* depth) {
    std::vector<V3> p(n);
    for (int i = 0; i < n; i++) p[i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    return threshold_local(*P, p, *depth);
}

void orc_mestimator_plane(const double* pts, int n, const double prior_n[3], double prior_offset, double out_n[3],
                          double* out_offset) {
    std::vector<V3> p(n);
    for (int i = 0; i < n; i++) p[i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    Plane prior = {{prior_n[0], prior_n[1], prior_n[2]}, prior_offset};
    Plane r = mestimator_plane(p, prior);
    out_n[0] = r.n.x;
    out_n[1] = r.n.y;
    out_n[2] = r.n.z;
    *out_offset = r.offset;
}

int orc_pca(const mld_params* P, const double* pts, int n, double normal[3], double mean[3]) {
    std::vector<V3> p(n);
    for (int i = 0; i < n; i++) p[i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    V3 nn, mm;
    int r = pca_plane(*P, p, nn, mm);
    normal[0] = nn.x;
    normal[1] = nn.y;
    normal[2] = nn.z;
    mean[0] = mm.x;
    mean[1] = mm.y;
    mean[2] = mm.z;
    return r;
}

}  // extern "C"

// C++ shim: the reference's DepthEstimator interface on top of the C-ABI (include/mld.h).
//
// Mirrors Mono_Lidar::DepthEstimator (monolidar_fusion/include/monolidar_fusion/DepthEstimator.h:39-359) and
// the companion types a caller touches — CameraPinhole (camera_pinhole.h:20-114), GroundPlane
// (RansacPlane.h:38-123), DepthEstimatorParameters (DepthEstimatorParameters.h:7-173), DepthResultType
// (eDepthResultType.h:8-30) — with the same method names, argument meaning and error behaviour (usage errors
// throw, per-feature failures are (resultType, -1)).  Eigen/PCL types are replaced by std::vector / plain
// structs with the same memory layout; when <Eigen/Core> is available the Eigen overloads are compiled too,
// so tracklets_depth (tracklets_depth/src/tracklet_depth_module.cpp:80,115) links against this header unchanged.
// Header-only; link with -lmld_hip.
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <functional>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "mld.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#include <Eigen/Geometry>
#define MLD_HAVE_EIGEN 1
#endif
// The reference's own cloud and image types (DepthEstimator.h:62-63: Cloud = pcl::PointCloud<pcl::PointXYZI>;
// RansacPlane.h:189: SemanticPlane(const cv::Mat&, Camera, ...)): used as they are when their headers are on the include
// path, so that tracklets_depth (tracklet_depth_module.cpp:63-117, 269-284) compiles against this header unchanged.
#if __has_include(<pcl/point_cloud.h>) && __has_include(<pcl/point_types.h>)
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#define MLD_HAVE_PCL 1
#endif
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define MLD_HAVE_OPENCV 1
#endif
#endif

class CameraPinhole final {
public:
    using Ptr = std::shared_ptr<CameraPinhole>;
    using ConstPtr = std::shared_ptr<const CameraPinhole>;
    explicit CameraPinhole(int width, int height, double focal_length, double principal_point_x,
                           double principal_point_y)
            : width_(width), height_(height), focal_length_(focal_length), principal_point_x_(principal_point_x),
              principal_point_y_(principal_point_y) {}
    void getImageSize(int& width, int& height) const {
        width = width_;
        height = height_;
    }
    mld_camera asStruct() const { return mld_camera{focal_length_, principal_point_x_, principal_point_y_, width_, height_}; }

private:
    int width_, height_;
    double focal_length_, principal_point_x_, principal_point_y_;
};

namespace Mono_Lidar {

enum DepthResultType {
    Unspecified = 0,
    Success = 1,
    RadiusSearchInsufficientPoints = 2,
    HistogramNoLocalMax = 3,
    TresholdDepthGlobalGreaterMax = 4,
    TresholdDepthGlobalSmallerMin = 5,
    TresholdDepthLocalGreaterMax = 6,
    TresholdDepthLocalSmallerMin = 7,
    TriangleNotPlanar = 8,
    TriangleNotPlanarInsufficientPoints = 9,
    CornerBehindCamera = 10,
    PlaneViewrayNotOrthogonal = 11,
    PcaIsPoint = 12,
    PcaIsLine = 13,
    PcaIsCubic = 14,
    InsufficientRoadPoints = 15,
    SuccessRoad = 16,
    RegionGrowingNearestSeedNotAvailable = 17,
    RegionGrowingSeedsOutOfRange = 18,
    RegionGrowingInsufficientPoints = 19,
    SuccessRegionGrowing = 20
};

#ifdef MLD_HAVE_PCL
// the caller's own types: 32-byte records read in place by mld_set_cloud (stride 32), `points` a contiguous vector
using PointXYZI = pcl::PointXYZI;
using PointCloud = pcl::PointCloud<pcl::PointXYZI>;
#else
// pcl::PointXYZI memory layout (8 floats, 32 bytes): x,y,z,pad, intensity,pad,pad,pad.
struct alignas(16) PointXYZI {
    float x, y, z, pad0_;
    float intensity, pad1_, pad2_, pad3_;
};

struct PointCloud {
    using Ptr = std::shared_ptr<PointCloud>;
    using ConstPtr = std::shared_ptr<const PointCloud>;
    std::vector<PointXYZI> points;
};
#endif
static_assert(sizeof(PointXYZI) == 32, "PointXYZI must match pcl::PointXYZI");

class DepthEstimatorParameters : public mld_params {
public:
    DepthEstimatorParameters() { mld_params_default(this); }
    void fromFile(const std::string& filePath) {
        char err[2048];
        if (mld_params_from_file(this, filePath.c_str(), err, sizeof(err)) != MLD_OK) throw std::string(err);
        absentKeys = err;  // "absent (read as 0): ..." or empty
        // The reference's own parameters.yaml has no ransac_plane_min_z / ransac_plane_max_z: cv::FileStorage reads 0 / 0,
        // the z pass-through keeps only points with z == 0 and every RANSAC plane estimate fails
        // (GroundPlane::ExceptionPclInvalid).  Same behaviour here - but said out loud.
        if (do_use_ransac_plane && (absentKeys.find("ransac_plane_min_z") != std::string::npos ||
                                    absentKeys.find("ransac_plane_max_z") != std::string::npos))
            std::fprintf(stderr,
                         "DepthEstimatorParameters: %s holds no ransac_plane_min_z / ransac_plane_max_z; they read as 0 "
                         "(as in the reference), so RansacPlane keeps only points with z == 0 and its estimate fails. "
                         "Add them (the header's defaults are -10000 / 10000) unless every frame brings a SemanticPlane or "
                         "a segmented plane.\n",
                         filePath.c_str());
    }
    std::string absentKeys;
};

// DepthCalculationStatistics (DepthCalculationStatistics.h:11-260): per-call counters of the result types (cleared by
// every CalculateDepth, DepthEstimator.cpp:445-446, counted by LogDepthCalcStats, :1039-1090) — same getter names.
class DepthCalculationStatistics {
public:
    void Clear() { _c.fill(0); }
    void SetFromTypes(const int32_t* types, int64_t n) {
        _pointCount = (int)n;
        mld_result_histogram(types, n, _c.data());
    }
    int getPointCount() const { return _pointCount; }
    int getUnspecified() const { return (int)_c[Unspecified]; }
    int getSuccess() const { return (int)_c[Success]; }
    int getRadiusSearchInsufficientPoints() const { return (int)_c[RadiusSearchInsufficientPoints]; }
    int getHistogramNoLocalMax() const { return (int)_c[HistogramNoLocalMax]; }
    int getTresholdDepthGlobalGreaterMax() const { return (int)_c[TresholdDepthGlobalGreaterMax]; }
    int getTresholdDepthGlobalSmallerMin() const { return (int)_c[TresholdDepthGlobalSmallerMin]; }
    int getTresholdDepthLocalGreaterMax() const { return (int)_c[TresholdDepthLocalGreaterMax]; }
    int getTresholdDepthLocalSmallerMin() const { return (int)_c[TresholdDepthLocalSmallerMin]; }
    int getTriangleNotPlanar() const { return (int)_c[TriangleNotPlanar]; }
    int getTriangleNotPlanarInsufficientPoints() const { return (int)_c[TriangleNotPlanarInsufficientPoints]; }
    int getCornerBehindCamera() const { return (int)_c[CornerBehindCamera]; }
    int getPlaneViewrayNotOrthogonal() const { return (int)_c[PlaneViewrayNotOrthogonal]; }
    int getPCAIsPoint() const { return (int)_c[PcaIsPoint]; }
    int getPCAIsLine() const { return (int)_c[PcaIsLine]; }
    int getPCAIsCubic() const { return (int)_c[PcaIsCubic]; }
    int getInsufficientRoadPoints() const { return (int)_c[InsufficientRoadPoints]; }
    int getSuccessRoad() const { return (int)_c[SuccessRoad]; }
    int getRegionGrowingNearestSeedNotAvailable() const { return (int)_c[RegionGrowingNearestSeedNotAvailable]; }
    int getRegionGrowingSeedsOutOfRange() const { return (int)_c[RegionGrowingSeedsOutOfRange]; }
    int getRegionGrowingInsufficientPoints() const { return (int)_c[RegionGrowingInsufficientPoints]; }
    int getSuccessRegionGrowing() const { return (int)_c[SuccessRegionGrowing]; }

private:
    int _pointCount = 0;
    std::array<int64_t, MLD_RESULT_TYPE_COUNT> _c{};
};

class GroundPlane {
public:
    using Ptr = std::shared_ptr<GroundPlane>;
    using Cloud = PointCloud;
    struct ExceptionPclInvalid : public std::exception {
        const char* what() const throw() override { return "In GroundPlane: Input pointcloud is invalid"; }
    };
    GroundPlane() = default;
    // The plane as an input object: coefficients a,b,c,d (lidar frame) + inlier indices of the ORIGINAL cloud.
    GroundPlane(const std::array<float, 4>& coeffs, std::vector<int> inliers)
            : is_segmented_(true), _modelCoeffs(coeffs), _inliersIndex(std::move(inliers)) {}
    virtual ~GroundPlane() = default;
    // Hook for a per-frame estimator (RansacPlane::CalculateInliersPlane, RansacPlane.cpp:41-140).  The
    // RANSAC implementation is a "next" row; the base class requires a pre-segmented plane.
    virtual void CalculateInliersPlane(const Cloud::ConstPtr& pointCloud, double /*min_z*/, double /*max_z*/) {
        if (!pointCloud || pointCloud->points.size() < 3) throw ExceptionPclInvalid();
        throw std::runtime_error("GroundPlane: no estimator attached; supply a segmented plane");
    }
    bool isSegmented() const { return is_segmented_; }
    std::array<float, 4>& getModelCoeffs() { return _modelCoeffs; }
    const std::array<float, 4>& getModelCoeffs() const { return _modelCoeffs; }
    const std::vector<int>& getInlinersIndex() const {
        materialize();
        return _inliersIndex;
    }
    // RansacPlane.h:116-122: a std::map<int, bool> lookup in the reference; here a bitmask over the point indices, built on
    // the first call for the current inlier list (O(1) per call afterwards).  Safe to call concurrently, as the reference's
    // const map lookup is: the bitmask is an immutable object published through an atomic shared_ptr (two threads that
    // find it stale each build their own).  Staleness: the key holds the list's size, storage and a fingerprint of 18
    // sampled entries, so that a same-size rewrite in place (`_inliersIndex = other` in a subclass) is noticed too; the
    // base class and RansacPlane invalidate explicitly wherever they rewrite the list.
    bool CheckPointInPlane(const int index) const {
        materialize();
        std::shared_ptr<const Lookup> lk = std::atomic_load(&_lookup);
        const uint64_t fp = fingerprint(_inliersIndex);
        if (!lk || lk->size != _inliersIndex.size() || lk->data != _inliersIndex.data() || lk->fp != fp) {
            auto fresh = std::make_shared<Lookup>();
            int mx = -1;
            for (int i : _inliersIndex) mx = i > mx ? i : mx;
            fresh->bits.assign((size_t)(mx + 1 + 63) / 64, 0ull);
            for (int i : _inliersIndex)
                if (i >= 0) fresh->bits[(size_t)i >> 6] |= 1ull << (i & 63);
            fresh->size = _inliersIndex.size();
            fresh->data = _inliersIndex.data();
            fresh->fp = fp;
            lk = fresh;
            std::atomic_store(&_lookup, lk);
        }
        const std::vector<uint64_t>& b = lk->bits;
        return index >= 0 && ((size_t)index >> 6) < b.size() && ((b[(size_t)index >> 6] >> (index & 63)) & 1ull);
    }
    // A plane estimated on the GPU keeps its inlier set there (a bitmask the kernels read); the index list of
    // getInlinersIndex (RansacPlane.h:100-103) is fetched when somebody asks for it - or when the frame slot that holds
    // the mask is about to be reused.
    virtual void materialize() const {}

protected:
    // A subclass that rewrites _inliersIndex calls this (the fingerprint in CheckPointInPlane's key catches a rewrite
    // that forgets to, unless the 18 sampled entries all kept their values).
    void inliersChanged() const { std::atomic_store(&_lookup, std::shared_ptr<const Lookup>()); }
    bool is_segmented_ = false;
    std::array<float, 4> _modelCoeffs{{0, 0, 0, 0}};
    mutable std::vector<int> _inliersIndex;

private:
    struct Lookup {  // CheckPointInPlane's bitmask of _inliersIndex + the state of the list it was built from
        std::vector<uint64_t> bits;
        size_t size = 0;
        const int* data = nullptr;
        uint64_t fp = 0;
    };
    static uint64_t fingerprint(const std::vector<int>& v) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)v.size();
        const size_t n = v.size();
        if (!n) return h;
        const size_t step = n > 16 ? n / 16 : 1;
        for (size_t i = 0; i < n; i += step) h = (h ^ (uint64_t)(uint32_t)v[i]) * 0x100000001B3ull;
        return (h ^ (uint64_t)(uint32_t)v[n - 1]) * 0x100000001B3ull;
    }
    mutable std::shared_ptr<const Lookup> _lookup;
};

// RansacPlane (RansacPlane.h:129-170): a GroundPlane that DepthEstimator::setInputCloud estimates on the GPU while it
// is not segmented (RansacPlane::CalculateInliersPlane, RansacPlane.cpp:41-140 -> mld_estimate_ground_plane).
// `seed` fixes the random draws (the reference's pcl::RandomSample is time-seeded).
class RansacPlane : public GroundPlane {
public:
    using Ptr = std::shared_ptr<RansacPlane>;
    RansacPlane() = default;
    explicit RansacPlane(const std::shared_ptr<DepthEstimatorParameters>& /*parameters*/, uint32_t seed_ = 0) : seed(seed_) {}
    void assign(const std::array<float, 4>& coeffs, std::vector<int> inliers) {
        _modelCoeffs = coeffs;
        _inliersIndex = std::move(inliers);
        inliersChanged();
        _fetch = nullptr;
        is_segmented_ = true;
    }
    // estimated on the GPU: coefficients and inlier count now, the index list on demand (`fetch` fills it)
    void assignLazy(const std::array<float, 4>& coeffs, int64_t n_inliers, std::function<void(std::vector<int>&)> fetch) {
        _modelCoeffs = coeffs;
        _inliersIndex.clear();
        inliersChanged();
        _numInliers = n_inliers;
        _fetch = std::move(fetch);
        is_segmented_ = true;
    }
    void materialize() const override {
        if (!_fetch) return;
        auto f = std::move(_fetch);
        _fetch = nullptr;
        f(_inliersIndex);
        inliersChanged();
    }
    bool inliersPending() const { return (bool)_fetch; }
    int64_t numInliers() const { return _fetch ? _numInliers : (int64_t)_inliersIndex.size(); }
    uint32_t seed = 0;

private:
    mutable std::function<void(std::vector<int>&)> _fetch;
    int64_t _numInliers = 0;
};

// SemanticPlane (RansacPlane.h:175-218): ground plane from a semantic label image, estimated on the GPU by
// DepthEstimator::setInputCloud while not segmented (SemanticPlane::CalculateInliersPlane, RansacPlane.cpp:195-274 ->
// mld_estimate_semantic_plane).  The image replaces cv::Mat: rows x cols uint8, copied like the reference's
// std::make_unique<cv::Mat>(img).  SemanticPlane::Camera is the estimator's own calibration.
class SemanticPlane : public RansacPlane {
public:
    using Ptr = std::shared_ptr<SemanticPlane>;
    // SemanticPlane::Camera (RansacPlane.h:177-188).  The GPU estimator projects with the DepthEstimator's own
    // calibration - which is what the reference's caller puts here (tracklet_depth_module.cpp:273-277: the same camera
    // info and the same _camLidarTransform it initialises the estimator with); a Camera that differs from it is refused
    // when the plane is estimated (DepthEstimator::checkSemanticCamera).
    struct Camera {
        double f = 0, cu = 0, cv = 0;
#ifdef MLD_HAVE_EIGEN
        Eigen::Affine3d transform_cam_lidar;
#else
        std::array<double, 12> transform_cam_lidar{};  // row-major 3x4
#endif
    };
    SemanticPlane(const uint8_t* img, int rows, int cols, int row_stride_bytes, std::set<int> groundplane_label,
                  double inlier_threshold)
            : rows_(rows), cols_(cols), groundplane_label_(groundplane_label.begin(), groundplane_label.end()),
              inlier_threshold_(inlier_threshold) {
        copyImage(img, rows, cols, row_stride_bytes);
    }
    SemanticPlane(const uint8_t* img, int rows, int cols, int row_stride_bytes, Camera cam, std::set<int> groundplane_label,
                  double inlier_threshold)
            : SemanticPlane(img, rows, cols, row_stride_bytes, std::move(groundplane_label), inlier_threshold) {
        cam_ = cam;
        has_cam_ = true;
    }
#ifdef MLD_HAVE_OPENCV
    // the reference's constructor (RansacPlane.h:189): an 8-bit single-channel label image, copied like its
    // std::make_unique<cv::Mat>(img)
    explicit SemanticPlane(const cv::Mat& img, Camera cam, std::set<int> groundplane_label, double inlier_threshold)
            : rows_(img.rows), cols_(img.cols), groundplane_label_(groundplane_label.begin(), groundplane_label.end()),
              inlier_threshold_(inlier_threshold), cam_(cam), has_cam_(true) {
        if (img.type() != CV_8UC1) throw std::runtime_error("SemanticPlane: the label image must be CV_8UC1 (MONO8)");
        semantic_image_.resize((size_t)rows_ * (size_t)cols_);
        for (int r = 0; r < rows_; r++)
            std::copy(img.ptr<uint8_t>(r), img.ptr<uint8_t>(r) + cols_, semantic_image_.begin() + (size_t)r * cols_);
    }
#endif
    bool hasCamera() const { return has_cam_; }
    const Camera& camera() const { return cam_; }
    const std::vector<uint8_t>& image() const { return semantic_image_; }
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    const std::vector<int>& labels() const { return groundplane_label_; }
    double inlierThreshold() const { return inlier_threshold_; }

private:
    void copyImage(const uint8_t* img, int rows, int cols, int row_stride_bytes) {
        semantic_image_.resize((size_t)rows * (size_t)cols);
        for (int r = 0; r < rows; r++)
            std::copy(img + (size_t)r * row_stride_bytes, img + (size_t)r * row_stride_bytes + cols,
                      semantic_image_.begin() + (size_t)r * cols);
    }
    std::vector<uint8_t> semantic_image_;
    int rows_, cols_;
    std::vector<int> groundplane_label_{6, 7, 8, 9};
    double inlier_threshold_{0.1};
    Camera cam_;
    bool has_cam_ = false;
};

class DepthEstimator {
public:
    using Point = PointXYZI;
    using Cloud = PointCloud;
    using UniquePtr = std::unique_ptr<DepthEstimator>;
    using SharedPtr = std::shared_ptr<DepthEstimator>;

    // max_frames > 1 exposes the C-ABI's frame slots (tracklets_depth keeps the previous frame resident in a second slot)
    explicit DepthEstimator(int device = 0, int max_frames = 1) : _device(device), _maxFrames(max_frames) {
        _lazyPlane.resize((size_t)(max_frames > 0 ? max_frames : 1));
    }
    mld_ctx* ctx() const { return _ctx; }
    ~DepthEstimator() {
        // planes that still count on this estimator for their inlier lists get them now
        for (size_t s = 0; s < _lazyPlane.size(); s++) {
            try {
                flushLazyPlane((int)s);
            } catch (...) {
            }
        }
        *_alive = false;
        if (_ctx) mld_destroy(_ctx);
    }
    DepthEstimator(const DepthEstimator&) = delete;
    DepthEstimator& operator=(const DepthEstimator&) = delete;

    bool InitConfig(const std::string& filePath, const bool printparams = true) {
        (void)printparams;
        _parameters = std::make_shared<DepthEstimatorParameters>();
        _parameters->fromFile(filePath);
        _isInitializedConfig = true;
        return true;
    }
    bool InitConfig(std::shared_ptr<DepthEstimatorParameters> parameters = nullptr, const bool printparams = false) {
        (void)printparams;
        _parameters = parameters ? parameters : std::make_shared<DepthEstimatorParameters>();
        _isInitializedConfig = true;
        return true;
    }

    // transform_lidar_to_cam: row-major 3x4 [R|t]
    bool Initialize(const std::shared_ptr<CameraPinhole>& camera, const std::array<double, 12>& transform_lidar_to_cam) {
        if (!_isInitializedConfig) throw "Call 'InitConfig' before calling 'Initialize'.";
        _camera = camera;
        _transform = transform_lidar_to_cam;
        if (_ctx) {
            for (size_t s = 0; s < _lazyPlane.size(); s++) flushLazyPlane((int)s);
            mld_destroy(_ctx);
            _ctx = nullptr;
        }
        int status = 0;
        mld_camera cam = camera->asStruct();
        _ctx = mld_create(_parameters.get(), &cam, _transform.data(), _device, _maxFrames, 0, 0, &status);
        if (!_ctx) {
            if (status == MLD_ERR_NO_ROAD_ESTIMATOR) throw "No road depth estimator selected.";
            throw std::string(mld_create_error());
        }
        _isInitialized = true;
        return true;
    }
#ifdef MLD_HAVE_EIGEN
    bool Initialize(const std::shared_ptr<CameraPinhole>& camera, const Eigen::Affine3d& transform_lidar_to_cam) {
        std::array<double, 12> T;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) T[r * 4 + c] = transform_lidar_to_cam.matrix()(r, c);
        return Initialize(camera, T);
    }
#endif

    void setInputCloud(const Cloud::ConstPtr& cloud, GroundPlane::Ptr& groundPlane, int slot = 0) {
        if (!_isInitialized) throw "call of 'setInputCloud' without 'initialize'";
        flushLazyPlane(slot);
        if (_parameters->do_use_ransac_plane) {
            if (groundPlane == nullptr) groundPlane = std::make_shared<RansacPlane>(_parameters);  // DepthEstimator.cpp:275-278
            if (!groundPlane->isSegmented()) {                                               // :281-283
                if (auto* rp = dynamic_cast<RansacPlane*>(groundPlane.get())) {
                    // RansacPlane / SemanticPlane::CalculateInliersPlane on the GPU, in ONE asynchronous chain with the
                    // upload and the projection (the plane is in place before the points are projected)
                    estimateFrame(cloud, *rp, groundPlane, slot, nullptr, 0, nullptr, nullptr);
                    return;
                }
                check(mld_set_cloud(_ctx, slot, cloud->points.data(), (int64_t)cloud->points.size(), (int)sizeof(PointXYZI)));
                noteCloud(cloud);
                groundPlane->CalculateInliersPlane(cloud, _parameters->ransac_plane_min_z, _parameters->ransac_plane_max_z);
                installPlane(*groundPlane, slot);
                return;
            }
            check(mld_set_cloud(_ctx, slot, cloud->points.data(), (int64_t)cloud->points.size(), (int)sizeof(PointXYZI)));
            noteCloud(cloud);
            installPlane(*groundPlane, slot);
        } else {
            check(mld_set_cloud(_ctx, slot, cloud->points.data(), (int64_t)cloud->points.size(), (int)sizeof(PointXYZI)));
            noteCloud(cloud);
            check(mld_set_ground_plane(_ctx, slot, nullptr, nullptr, 0));
        }
    }

    // TrackletDepthModule::process's GPU side as ONE C call (mld_tracklets_frame): setInputCloud(cloud, groundPlane) on
    // slot_cur - the plane estimated inside the call when it is not segmented yet, as the reference does every frame
    // (tracklet_depth_module.cpp:269-284) -, the feature marshalling, both CalculateDepth calls (previous frame from its
    // resident slot_last, -1: none) and the float32 scatter.  Throws GroundPlane::ExceptionPclInvalid when the estimation
    // fails; d_last is answered even then (the reference's two try blocks, :318-347), d_cur is -1.
    void trackletsFrame(const Cloud::ConstPtr& cloud, GroundPlane::Ptr& groundPlane, int slot_cur, int slot_last,
                        const float* u_new, const float* v_new, const float* u_old, const float* v_old, const uint8_t* is_new,
                        int64_t n_tracks, float* d_cur, float* d_last) {
        if (!_isInitialized) throw "call of 'setInputCloud' without 'initialize'";
        flushLazyPlane(slot_cur);
        const bool road = _parameters->do_use_ransac_plane;
        if (road && groundPlane == nullptr) groundPlane = std::make_shared<RansacPlane>(_parameters);
        mld_plane_request rq{};
        const mld_plane_request* prq = nullptr;
        const float* coeffs = nullptr;
        const int32_t* inl = nullptr;
        int64_t n_inl = 0;
        RansacPlane* rp = road ? dynamic_cast<RansacPlane*>(groundPlane.get()) : nullptr;
        static_assert(sizeof(int) == sizeof(int32_t), "int must be 32 bit");
        if (road && !groundPlane->isSegmented()) {
            if (!rp) {  // a foreign GroundPlane subclass estimates itself on the CPU
                try {
                    groundPlane->CalculateInliersPlane(cloud, _parameters->ransac_plane_min_z, _parameters->ransac_plane_max_z);
                } catch (const GroundPlane::ExceptionPclInvalid&) {
                    // the reference's two try blocks (tracklet_depth_module.cpp:318-347): the previous frame's features
                    // are answered all the same - the frame runs without a plane -, the current ones are invalid
                    const int rc0 = mld_tracklets_frame(_ctx, slot_cur, slot_last, cloud->points.data(), (int64_t)cloud->points.size(),
                                                        (int)sizeof(PointXYZI), nullptr, nullptr, nullptr, 0, u_new, v_new, u_old,
                                                        v_old, is_new, n_tracks, d_cur, d_last, nullptr, nullptr, nullptr, nullptr);
                    noteCloud(cloud);
                    for (int64_t i = 0; i < n_tracks; i++) d_cur[i] = -1.f;
                    if (rc0 != MLD_OK && rc0 != MLD_ERR_CLOUD_TOO_SMALL) check(rc0);
                    throw;
                }
            } else {
                rq.kind = MLD_PLANE_RANSAC;
                rq.seed = rp->seed;
                if (auto* sp = dynamic_cast<SemanticPlane*>(rp)) {
                    checkSemanticCamera(*sp);
                    rq.kind = MLD_PLANE_SEMANTIC;
                    rq.label_image = sp->image().data();
                    rq.rows = sp->rows();
                    rq.cols = sp->cols();
                    rq.row_stride_bytes = sp->cols();
                    rq.ground_labels = reinterpret_cast<const int32_t*>(sp->labels().data());
                    rq.n_labels = (int)sp->labels().size();
                    rq.inlier_threshold = sp->inlierThreshold();
                }
                prq = &rq;
            }
        }
        if (road && !prq) {
            coeffs = groundPlane->getModelCoeffs().data();
            inl = reinterpret_cast<const int32_t*>(groundPlane->getInlinersIndex().data());
            n_inl = (int64_t)groundPlane->getInlinersIndex().size();
        }
        mld_plane_result pr{};
        const int rc = mld_tracklets_frame(_ctx, slot_cur, slot_last, cloud->points.data(), (int64_t)cloud->points.size(),
                                           (int)sizeof(PointXYZI), prq, coeffs, inl, n_inl, u_new, v_new, u_old, v_old, is_new,
                                           n_tracks, d_cur, d_last, nullptr, nullptr, nullptr, &pr);
        noteCloud(cloud);
        check(rc);
        if (prq) {
            mld_ctx* ctx = _ctx;
            std::shared_ptr<bool> alive = _alive;
            const int64_t n_in = pr.n_inliers;
            rp->assignLazy({pr.coeffs[0], pr.coeffs[1], pr.coeffs[2], pr.coeffs[3]}, n_in, [ctx, slot_cur, alive, n_in](std::vector<int>& out) {
                if (!*alive) throw std::runtime_error("GroundPlane: the DepthEstimator that holds this plane's inliers is gone");
                out.resize((size_t)n_in);
                int64_t k = 0;
                if (mld_get_ground_plane_inliers(ctx, slot_cur, reinterpret_cast<int32_t*>(out.data()), n_in, &k) != MLD_OK)
                    throw std::runtime_error(std::string("GroundPlane: ") + mld_last_error(ctx));
                out.resize((size_t)k);
            });
            if ((size_t)slot_cur < _lazyPlane.size()) _lazyPlane[(size_t)slot_cur] = groundPlane;
        }
        _installedPlane = road ? groundPlane.get() : nullptr;
    }

    std::shared_ptr<DepthEstimatorParameters> getParameters() { return _parameters; }
    std::shared_ptr<CameraPinhole> getCamera() { return _camera; }
    std::array<double, 12> getTransformLidarToCam() { return _transform; }
    double getPointDepthCamVisible(int index) {
        double d = 0;
        check(mld_get_point_depth_cam_visible(_ctx, 0, index, &d));
        return d;
    }

    // getCloudCameraCs: 3 x N column-major (x0,y0,z0,x1,...)
    void getCloudCameraCs(std::vector<double>& xyz) {
        xyz.resize((size_t)_numPoints * 3);
        check(mld_get_cloud_camera_cs(_ctx, 0, xyz.data(), _numPoints));
    }
    // getPointsCloudImageCs: 2 x Nvis column-major
    void getPointsCloudImageCs(std::vector<double>& uv) {
        int64_t n = 0;
        check(mld_get_visible_count(_ctx, 0, &n));
        uv.resize((size_t)n * 2);
        check(mld_get_visible_image_points(_ctx, 0, uv.data(), n));
    }

    // ---- debug mode (DepthEstimator.h:85-87) and the debug clouds (DepthEstimator.cpp:335-398) ----
    void ActivateDebugMode() { _debugMode = true; }
    // triangle corners published by CalculatePlaneCorners (PlaneEstimationCalcMaxSpanningTriangle.cpp:27-32) in the
    // last debug-mode CalculateDepth, feature order
    void getCloudTriangleCorners(Cloud::Ptr& pointCloud_triangle_corner) {
        pointCloud_triangle_corner->points.clear();
        for (size_t i = 0; i + 8 < _dbgCorners.size(); i += 9) {
            if (_dbgCorners[i] != _dbgCorners[i]) continue;  // NaN: no triangle for this feature
            for (int c = 0; c < 3; c++) pushPoint(*pointCloud_triangle_corner, &_dbgCorners[i + 3 * c]);
        }
    }
    // `_points_groundplane` (DepthEstimator.cpp:294-308)
    void getCloudRansacPlane(Cloud::Ptr& pointCloud_plane_ransac) {
        int64_t n = 0;
        check(mld_get_ground_plane_cloud(_ctx, 0, nullptr, 0, &n));
        std::vector<double> xyz((size_t)n * 3);
        if (n) check(mld_get_ground_plane_cloud(_ctx, 0, xyz.data(), n, &n));
        pointCloud_plane_ransac->points.clear();
        for (int64_t i = 0; i < n; i++) pushPoint(*pointCloud_plane_ransac, &xyz[(size_t)i * 3]);
    }
    // Intersection points ray x plane of the features with a valid depth (debug mode).  The reference's push is
    // commented out (DepthEstimator.cpp:1032), its cloud is always empty; this returns what the member is documented
    // to hold (DepthEstimator.h:328).
    void getCloudInterpolated(Cloud::Ptr& pointCloud_interpolated) {
        pointCloud_interpolated->points.clear();
        const mld_camera cam = _camera->asStruct();
        for (size_t i = 0; i < _dbgDepth.size(); i++) {
            if (!(_dbgDepth[i] >= 0)) continue;
            const double p[3] = {(_dbgUv[2 * i] - cam.principal_point_x) / cam.focal_length * _dbgDepth[i],
                                 (_dbgUv[2 * i + 1] - cam.principal_point_y) / cam.focal_length * _dbgDepth[i], _dbgDepth[i]};
            pushPoint(*pointCloud_interpolated, p);
        }
    }
    void getCloudInterpolatedPlane(Cloud::Ptr& pointCloud_interpolated_plane) {
        getCloudInterpolated(pointCloud_interpolated_plane);  // same source vector in the reference (:343-345)
    }
    // never filled by the reference (push commented out, DepthEstimator.cpp:663): always empty
    void getCloudNeighbors(Cloud::Ptr& pointCloud_neighbors) { pointCloud_neighbors->points.clear(); }
    // getCloudCameraCs as a cloud (DepthEstimator.cpp:314-334)
    void getCloudCameraCs(Cloud::Ptr& pointCloud_cam_cs) {
        std::vector<double> xyz;
        getCloudCameraCs(xyz);
        pointCloud_cam_cs->points.clear();
        for (int64_t i = 0; i < _numPoints; i++) pushPoint(*pointCloud_cam_cs, &xyz[(size_t)i * 3]);
    }

    // CalculateDepth overloads (DepthEstimator.cpp:404-488).  uv: 2 x F column-major.
    void CalculateDepth(const Cloud::ConstPtr& pointCloud, const std::vector<double>& points_image_cs,
                        std::vector<double>& points_depths, GroundPlane::Ptr& ransacPlane) {
        setInputCloud(pointCloud, ransacPlane);
        CalculateDepth(points_image_cs, points_depths, ransacPlane);
    }
    void CalculateDepth(const Cloud::ConstPtr& pointCloud, const std::vector<double>& points_image_cs,
                        std::vector<double>& points_depths, std::vector<int>& resultType, GroundPlane::Ptr& ransacPlane) {
        const int64_t F = (int64_t)(points_image_cs.size() / 2);
        if (estimateCall(pointCloud, ransacPlane)) {
            // the reference's production call (tracklet_depth_module.cpp:269-284, DepthEstimator.cpp:275-283): the
            // plane is not segmented yet - estimated on the GPU ahead of the projection, one C-ABI call for the frame
            points_depths.resize((size_t)F);
            resultType.resize((size_t)F);
            static_assert(sizeof(int) == sizeof(int32_t), "int must be 32 bit");
            if (ransacPlane == nullptr) ransacPlane = std::make_shared<RansacPlane>(_parameters);
            flushLazyPlane(0);
            estimateFrame(pointCloud, *static_cast<RansacPlane*>(ransacPlane.get()), ransacPlane, 0, points_image_cs.data(), F,
                          points_depths.data(), reinterpret_cast<int32_t*>(resultType.data()));
            _depthCalcStats.SetFromTypes(reinterpret_cast<const int32_t*>(resultType.data()), F);
            return;
        }
        if (frameCall(pointCloud, points_image_cs.data(), F, ransacPlane)) {  // one C-ABI call for the whole frame
            points_depths.resize((size_t)F);
            resultType.resize((size_t)F);
            static_assert(sizeof(int) == sizeof(int32_t), "int must be 32 bit");
            flushLazyPlane(0);
            const bool road = _parameters->do_use_ransac_plane;
            check(mld_calculate_depth_frame(_ctx, 0, pointCloud->points.data(), (int64_t)pointCloud->points.size(),
                                            (int)sizeof(PointXYZI), road ? ransacPlane->getModelCoeffs().data() : nullptr,
                                            road ? reinterpret_cast<const int32_t*>(ransacPlane->getInlinersIndex().data()) : nullptr,
                                            road ? (int64_t)ransacPlane->getInlinersIndex().size() : 0,
                                            points_image_cs.data(), F, points_depths.data(),
                                            reinterpret_cast<int32_t*>(resultType.data())));
            _numPoints = (int64_t)pointCloud->points.size();
            _isInitializedPointCloud = true;
            _installedPlane = road ? ransacPlane.get() : nullptr;
            _depthCalcStats.SetFromTypes(reinterpret_cast<const int32_t*>(resultType.data()), F);
            return;
        }
        setInputCloud(pointCloud, ransacPlane);
        CalculateDepth(points_image_cs, points_depths, resultType, ransacPlane);
    }
    void CalculateDepth(const std::vector<double>& points_image_cs, std::vector<double>& points_depths,
                        const GroundPlane::Ptr& ransacPlane) {
        std::vector<int> types;
        CalculateDepth(points_image_cs, points_depths, types, ransacPlane);
    }
    void CalculateDepth(const std::vector<double>& points_image_cs, std::vector<double>& points_depths,
                        std::vector<int>& resultType, const GroundPlane::Ptr& ransacPlane) {
        if (!_isInitializedPointCloud) throw "call of 'CalculateDepth' without 'SetInputCloud'";
        // The reference decides per call (DepthEstimator.cpp:580): a null plane skips the road fallback for THIS call
        // only, a plane other than the one installed with the cloud is installed first.
        const uint32_t flags = planeForCall(ransacPlane);
        const int64_t F = (int64_t)(points_image_cs.size() / 2);
        points_depths.resize((size_t)F);  // callee-resized outputs, DepthEstimator.cpp:442-443
        resultType.resize((size_t)F);
        static_assert(sizeof(int) == sizeof(int32_t), "int must be 32 bit");
        if (_debugMode) {
            _dbgCorners.assign((size_t)F * 9, 0.0);
            check(mld_calculate_depth_debug(_ctx, 0, points_image_cs.data(), F, points_depths.data(),
                                            reinterpret_cast<int32_t*>(resultType.data()), _dbgCorners.data()));
            _dbgUv = points_image_cs;
            _dbgDepth = points_depths;
            _depthCalcStats.SetFromTypes(reinterpret_cast<const int32_t*>(resultType.data()), F);
            return;
        }
        check(mld_calculate_depth_opts(_ctx, 0, points_image_cs.data(), F, points_depths.data(),
                                       reinterpret_cast<int32_t*>(resultType.data()), flags));
        _depthCalcStats.SetFromTypes(reinterpret_cast<const int32_t*>(resultType.data()), F);
    }
    std::pair<DepthResultType, double> CalculateDepth(const std::array<double, 2>& point_image_cs,
                                                      const GroundPlane::Ptr& ransacPlane) {
        std::vector<double> uv{point_image_cs[0], point_image_cs[1]}, d;
        std::vector<int> t;
        CalculateDepth(uv, d, t, ransacPlane);
        return {static_cast<DepthResultType>(t[0]), d[0]};
    }
#ifdef MLD_HAVE_EIGEN
    void CalculateDepth(const Cloud::ConstPtr& pointCloud, const Eigen::Matrix2Xd& points_image_cs,
                        Eigen::VectorXd& points_depths, GroundPlane::Ptr& ransacPlane) {
        Eigen::VectorXi types;
        CalculateDepth(pointCloud, points_image_cs, points_depths, types, ransacPlane);
    }
    void CalculateDepth(const Cloud::ConstPtr& pointCloud, const Eigen::Matrix2Xd& points_image_cs,
                        Eigen::VectorXd& points_depths, Eigen::VectorXi& resultType, GroundPlane::Ptr& ransacPlane) {
        const int64_t F = points_image_cs.cols();
        if (estimateCall(pointCloud, ransacPlane)) {  // the production call: plane estimated inside the one frame call
            points_depths.resize(F);
            resultType.resize(F);
            if (ransacPlane == nullptr) ransacPlane = std::make_shared<RansacPlane>(_parameters);
            flushLazyPlane(0);
            estimateFrame(pointCloud, *static_cast<RansacPlane*>(ransacPlane.get()), ransacPlane, 0, points_image_cs.data(), F,
                          points_depths.data(), resultType.data());
            _depthCalcStats.SetFromTypes(resultType.data(), F);
            return;
        }
        setInputCloud(pointCloud, ransacPlane);
        points_depths.resize(F);
        resultType.resize(F);
        check(mld_calculate_depth(_ctx, 0, points_image_cs.data(), F, points_depths.data(), resultType.data()));
        _depthCalcStats.SetFromTypes(resultType.data(), F);
    }
    // feature-only overloads (DepthEstimator.cpp:421-488)
    void CalculateDepth(const Eigen::Matrix2Xd& featurePoints_image_cs, Eigen::VectorXd& points_depths,
                        const GroundPlane::Ptr& ransacPlane) {
        Eigen::VectorXi types;
        CalculateDepth(featurePoints_image_cs, points_depths, types, ransacPlane);
    }
    void CalculateDepth(const Eigen::Matrix2Xd& featurePoints_image_cs, Eigen::VectorXd& points_depths,
                        Eigen::VectorXi& resultType, const GroundPlane::Ptr& ransacPlane) {
        if (!_isInitializedPointCloud) throw "call of 'CalculateDepth' without 'SetInputCloud'";
        const uint32_t flags = planeForCall(ransacPlane);
        const int64_t F = featurePoints_image_cs.cols();
        points_depths.resize(F);
        resultType.resize(F);
        check(mld_calculate_depth_opts(_ctx, 0, featurePoints_image_cs.data(), F, points_depths.data(), resultType.data(), flags));
        _depthCalcStats.SetFromTypes(resultType.data(), F);
    }
    std::pair<DepthResultType, double> CalculateDepth(const Eigen::Vector2d& point_image_cs, const GroundPlane::Ptr& ransacPlane) {
        return CalculateDepth(std::array<double, 2>{point_image_cs.x(), point_image_cs.y()}, ransacPlane);
    }
#endif

    // getDepthCalcStats (DepthEstimator.cpp:400-402): the counters of the last CalculateDepth call
    const DepthCalculationStatistics& getDepthCalcStats() { return _depthCalcStats; }
    // histogram of an arbitrary result-type array
    static std::array<int64_t, MLD_RESULT_TYPE_COUNT> getDepthCalcStats(const std::vector<int>& resultType) {
        std::array<int64_t, MLD_RESULT_TYPE_COUNT> c{};
        mld_result_histogram(reinterpret_cast<const int32_t*>(resultType.data()), (int64_t)resultType.size(), c.data());
        return c;
    }

private:
    // installs a segmented plane (coefficients + inlier set) as the slot's ground plane
    void installPlane(const GroundPlane& gp, int slot = 0) {
        const auto& c = gp.getModelCoeffs();
        const auto& inl = gp.getInlinersIndex();
        static_assert(sizeof(int) == sizeof(int32_t), "int must be 32 bit");
        check(mld_set_ground_plane(_ctx, slot, c.data(), reinterpret_cast<const int32_t*>(inl.data()), (int64_t)inl.size()));
        _installedPlane = &gp;
    }
    // per-call plane of the feature-only overloads: returns the flags for mld_calculate_depth_opts
    uint32_t planeForCall(const GroundPlane::Ptr& ransacPlane) {
        if (!_parameters->do_use_ransac_plane) return 0u;
        if (ransacPlane == nullptr) return MLD_CALC_SKIP_ROAD;
        if (ransacPlane.get() != _installedPlane && ransacPlane->isSegmented()) installPlane(*ransacPlane);
        return 0u;
    }
    // CalculateDepth(cloud, ..., plane) whose plane still has to be estimated on the GPU (a null pointer, or a RansacPlane /
    // SemanticPlane that is not segmented): mld_calculate_depth_frame_estimate
    bool estimateCall(const Cloud::ConstPtr& cloud, const GroundPlane::Ptr& gp) const {
        if (!_isInitialized || _debugMode || !cloud || !_parameters->do_use_ransac_plane) return false;
        if (gp == nullptr) return true;
        return !gp->isSegmented() && dynamic_cast<RansacPlane*>(gp.get()) != nullptr;
    }
    // setInputCloud(cloud, unsegmented plane) [+ the feature loop when F > 0] as ONE call; the plane object receives the
    // coefficients and the inlier count, its index list stays on the GPU until it is asked for
    void estimateFrame(const Cloud::ConstPtr& cloud, RansacPlane& rp, const GroundPlane::Ptr& owner, int slot, const double* uv,
                       int64_t F, double* depths, int32_t* types) {
        mld_plane_request rq{};
        rq.kind = MLD_PLANE_RANSAC;
        rq.seed = rp.seed;
        static_assert(sizeof(int) == sizeof(int32_t), "int must be 32 bit");
        if (auto* sp = dynamic_cast<SemanticPlane*>(&rp)) {
            checkSemanticCamera(*sp);
            rq.kind = MLD_PLANE_SEMANTIC;
            rq.label_image = sp->image().data();
            rq.rows = sp->rows();
            rq.cols = sp->cols();
            rq.row_stride_bytes = sp->cols();
            rq.ground_labels = reinterpret_cast<const int32_t*>(sp->labels().data());
            rq.n_labels = (int)sp->labels().size();
            rq.inlier_threshold = sp->inlierThreshold();
        }
        mld_plane_result pr{};
        check(mld_calculate_depth_frame_estimate(_ctx, slot, cloud->points.data(), (int64_t)cloud->points.size(),
                                                 (int)sizeof(PointXYZI), &rq, uv, F, depths, types, &pr));
        noteCloud(cloud);
        mld_ctx* ctx = _ctx;
        std::shared_ptr<bool> alive = _alive;
        const int64_t n_inl = pr.n_inliers;
        rp.assignLazy({pr.coeffs[0], pr.coeffs[1], pr.coeffs[2], pr.coeffs[3]}, n_inl, [ctx, slot, alive, n_inl](std::vector<int>& out) {
            if (!*alive) throw std::runtime_error("GroundPlane: the DepthEstimator that holds this plane's inliers is gone");
            out.resize((size_t)n_inl);
            int64_t n = 0;
            if (mld_get_ground_plane_inliers(ctx, slot, reinterpret_cast<int32_t*>(out.data()), n_inl, &n) != MLD_OK)
                throw std::runtime_error(std::string("GroundPlane: ") + mld_last_error(ctx));
            out.resize((size_t)n);
        });
        if ((size_t)slot < _lazyPlane.size()) _lazyPlane[(size_t)slot] = owner;
        _installedPlane = &rp;
    }
    // SemanticPlane::Camera must be the estimator's calibration: the GPU estimator projects with that one
    void checkSemanticCamera(const SemanticPlane& sp) const {
        if (!sp.hasCamera()) return;
        const mld_camera cam = _camera->asStruct();
        const SemanticPlane::Camera& c = sp.camera();
        bool same = c.f == cam.focal_length && c.cu == cam.principal_point_x && c.cv == cam.principal_point_y;
        for (int r = 0; r < 3 && same; r++)
            for (int k = 0; k < 4; k++) {
#ifdef MLD_HAVE_EIGEN
                const double v = c.transform_cam_lidar.matrix()(r, k);
#else
                const double v = c.transform_cam_lidar[(size_t)(r * 4 + k)];
#endif
                same = same && v == _transform[(size_t)(r * 4 + k)];
            }
        if (!same)
            throw std::runtime_error("SemanticPlane::Camera differs from the calibration the DepthEstimator was initialised "
                                     "with: the GPU plane estimator projects with the estimator's own (unsupported)");
    }
    // a plane whose inlier list still lives in `slot` gets it now: the slot's mask is about to be overwritten
    void flushLazyPlane(int slot) {
        if ((size_t)slot >= _lazyPlane.size()) return;
        if (auto gp = _lazyPlane[(size_t)slot].lock()) gp->materialize();
        _lazyPlane[(size_t)slot].reset();
    }
    void noteCloud(const Cloud::ConstPtr& cloud) {
        _numPoints = (int64_t)cloud->points.size();
        _isInitializedPointCloud = true;
        _installedPlane = nullptr;
    }
    // the whole frame can go through mld_calculate_depth_frame: no debug vectors wanted and nothing to estimate
    bool frameCall(const Cloud::ConstPtr& cloud, const double* uv, int64_t F, const GroundPlane::Ptr& gp) const {
        (void)uv;
        (void)F;
        if (!_isInitialized || _debugMode || !cloud) return false;
        if (!_parameters->do_use_ransac_plane) return true;
        return gp != nullptr && gp->isSegmented();
    }
    const GroundPlane* _installedPlane = nullptr;
    std::vector<std::weak_ptr<GroundPlane>> _lazyPlane;  // per slot: the plane whose inlier list is still on the GPU
    std::shared_ptr<bool> _alive = std::make_shared<bool>(true);

    void check(int rc) {
        if (rc == MLD_OK) return;
        if (rc == MLD_ERR_CLOUD_TOO_SMALL) throw GroundPlane::ExceptionPclInvalid();
        throw std::runtime_error(std::string("DepthEstimator: ") + mld_last_error(_ctx));
    }
    // FillCloud (DepthEstimator.cpp:351-371): xyz as float, intensity 1
    static void pushPoint(Cloud& cloud, const double* xyz) {
        PointXYZI q{};
        q.x = (float)xyz[0];
        q.y = (float)xyz[1];
        q.z = (float)xyz[2];
        q.intensity = 1;
        cloud.points.push_back(q);
    }
    DepthCalculationStatistics _depthCalcStats;
    bool _debugMode = false;
    std::vector<double> _dbgCorners, _dbgUv, _dbgDepth;
    int _device;
    int _maxFrames = 1;
    mld_ctx* _ctx = nullptr;
    bool _isInitialized = false, _isInitializedConfig = false, _isInitializedPointCloud = false;
    std::shared_ptr<DepthEstimatorParameters> _parameters;
    std::shared_ptr<CameraPinhole> _camera;
    std::array<double, 12> _transform{};
    int64_t _numPoints = 0;
};

}  // namespace Mono_Lidar

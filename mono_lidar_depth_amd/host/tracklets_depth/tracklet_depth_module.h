// C++ shim: tracklets_depth::TrackletDepthModule on top of the C-ABI (include/mld.h) — the caller of the hot path
// (tracklets_depth/include/tracklets_depth/tracklet_depth_module.h:20-160, src/tracklet_depth_module.cpp).
//
// Same method names, tracklet bookkeeping and error behaviour as the reference; ROS / feature_tracking message types
// are replaced by the plain structs below (same fields).  What changes underneath: the previous frame is not
// re-projected for the features of new tracklets (CalculateFeatureDepthsLastFrame re-runs setInputCloud on the old
// cloud in the reference, :115) — its slot (cloud, pixel map, ground plane) stays resident on the GPU and the two
// slots ping-pong; upload, ground plane, projection, both depth calls and the float32 scatter of a frame run as ONE
// mld_tracklets_frame call.
// Header-only; link with -lmld_hip.
#pragma once

#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <set>
#include <utility>
#include <vector>

#include "../monolidar_fusion/DepthEstimator.h"

namespace tracklets_depth {

// matches_msg_ros/FeaturePoint.msg, Tracklet.msg, MatchesMsg.msg (input) — feature_points[0] is the newest
struct FeaturePointIn {
    float u, v;
};
struct TrackletIn {
    uint64_t id = 0;
    int age = 0;
    std::vector<FeaturePointIn> feature_points;
};
struct MatchesMsgIn {
    std::vector<TrackletIn> tracks;
    std::vector<double> stamps;
};
// matches_msg_depth_ros/FeaturePoint.msg:3-5, Tracklet.msg, MatchesMsg.msg (output)
struct FeaturePoint {
    float u, v, d;
};
struct Tracklet {
    uint64_t id = 0;
    int age = 0;
    std::vector<FeaturePoint> feature_points;
};
struct MatchesMsg {
    std::vector<Tracklet> tracks;
    std::vector<double> stamps;
};

// TempTrackletFrame.h:5-33: one incoming track; `has_last` = TempTrackletFrameExp (a new tracklet brings two features)
struct TempTrackletFrame {
    int _keyFrameId = 0;
    std::pair<int, int> _feature{0, 0};
    bool has_last = false;
    std::pair<int, int> _featureLast{0, 0};
};

class TrackletDepthModule {
public:
    using Cloud = Mono_Lidar::PointCloud;
    using TypeTrackletKey = uint64_t;

    explicit TrackletDepthModule(const Mono_Lidar::DepthEstimatorParameters& depth_estimator_parameters, int device = 0)
            : depth_estimator_parameters_(depth_estimator_parameters), _depthEstimator(device, 2) {}

    void SetCameraLidarTransform(const std::array<double, 12>& camera_T_lidar) { _camLidarTransform = camera_T_lidar; }
    // stands in for the sensor_msgs::CameraInfo argument of process() (CreateCameraPinholeFromCameraInfo, :8-20)
    void SetCamera(const std::shared_ptr<CameraPinhole>& camera) { _camera = camera; }

    bool InitDepthEstimatorPre() {  // :388-396
        if (!_depthEstimator.InitConfig(std::make_shared<Mono_Lidar::DepthEstimatorParameters>(depth_estimator_parameters_)))
            throw "Error in 'initConfig' of DepthEstimator.";
        return true;
    }
    bool InitDepthEstimatorPost() {  // :374-386
        if (!_depthEstimator.Initialize(_camera, _camLidarTransform)) throw "Error in 'Initialize' of DepthEstimator.";
        _isDepthEstimatorInitialized = true;
        return true;
    }

    // TrackletDepthModule::process (:261-372).  `gp`: the frame's ground plane — nullptr (a RansacPlane is created
    // and estimated on the GPU, DepthEstimator.cpp:275-283), a SemanticPlane built from the label image (:270-284), or
    // a pre-segmented plane.  `msg_out` receives what convert_tracklets_to_matches_msg produces (:351-353).
    void process(const Cloud::ConstPtr& cloud_in, const MatchesMsgIn& tracklets_in, Mono_Lidar::GroundPlane::Ptr gp,
                 MatchesMsg* msg_out = nullptr) {
        if (!_isDepthEstimatorInitialized) {
            InitDepthEstimatorPre();
            InitDepthEstimatorPost();
        }
        std::vector<TempTrackletFrame> tempFrames;
        const auto frameCount = ExractNewTrackletFrames(tracklets_in, tempFrames);
        const int n = frameCount.first;
        std::vector<float> u_new((size_t)n), v_new((size_t)n), u_old((size_t)n), v_old((size_t)n), d_cur((size_t)n, -1.f),
                d_last((size_t)n, -1.f);
        std::vector<uint8_t> is_new((size_t)n);
        for (int i = 0; i < n; i++) {
            const TrackletIn& t = tracklets_in.tracks[(size_t)i];
            u_new[i] = t.feature_points.at(0).u;
            v_new[i] = t.feature_points.at(0).v;
            is_new[i] = tempFrames[i].has_last ? 1 : 0;
            if (is_new[i]) {
                u_old[i] = t.feature_points.at(1).u;
                v_old[i] = t.feature_points.at(1).v;
            }
        }
        const int slot_cur = _slotCur, slot_last = _haveLast ? 1 - _slotCur : -1;
        bool cur_ok = true;
        try {
            // the frame's whole GPU side is ONE C call: upload + ground plane (estimated inside the call when it is not
            // segmented yet) + the only projection of this frame + both depth calls + scatter
            _depthEstimator.trackletsFrame(cloud_in, gp, slot_cur, slot_last, u_new.data(), v_new.data(), u_old.data(),
                                           v_old.data(), is_new.data(), n, d_cur.data(), d_last.data());
        } catch (const Mono_Lidar::GroundPlane::ExceptionPclInvalid&) {
            // :337-347  current frame continues with invalid depths (trackletsFrame has left -1 there and answered the
            // previous frame's features - whether the GPU estimator or a foreign plane's CPU estimator threw), plane and
            // cloud are forgotten
            cur_ok = false;
            gp = nullptr;
        }
        groundPlaneLast_ = gp;
        // SaveFeatureDepths consumes the last-frame depths compacted over the new tracklets (index j, :143-147)
        std::vector<float> dl;
        dl.reserve((size_t)frameCount.second);
        for (int i = 0; i < n; i++)
            if (is_new[i]) dl.push_back(slot_last >= 0 ? d_last[i] : -1.f);  // no previous cloud: -1 (:93-96)
        std::vector<TypeTrackletKey> updatedIds;
        SaveFeatureDepths(tempFrames, dl, d_cur, updatedIds);
        if (msg_out) convert_tracklets_to_matches_msg(tracklets_in, updatedIds, *msg_out);
        TidyUpTracklets(updatedIds);
        // remember cloud / plane: the current slot becomes the previous one (:333, :348)
        _haveLast = cur_ok;
        if (cur_ok) _slotCur = 1 - _slotCur;
    }

    // :23-61
    std::pair<int, int> ExractNewTrackletFrames(const MatchesMsgIn& tracklets_in, std::vector<TempTrackletFrame>& newFrames) {
        int frameCountNew = 0, frameCountOld = 0;
        for (const auto& tracklet : tracklets_in.tracks) {
            const int id = (int)tracklet.id;
            TempTrackletFrame f;
            f._keyFrameId = id;
            const auto& matchNew = tracklet.feature_points.at(0);
            f._feature = std::make_pair((int)matchNew.u, (int)matchNew.v);
            if (!_trackletMap.count((TypeTrackletKey)id)) {
                const auto& matchOld = tracklet.feature_points.at(1);
                f.has_last = true;
                f._featureLast = std::make_pair((int)matchOld.u, (int)matchOld.v);
                frameCountOld++;
            }
            frameCountNew++;
            newFrames.push_back(f);
        }
        return std::make_pair(frameCountNew, frameCountOld);
    }

    // :119-169
    std::pair<int, int> SaveFeatureDepths(const std::vector<TempTrackletFrame>& newFrames, const std::vector<float>& depthsLastFrame,
                                          const std::vector<float>& depthsCurFrame, std::vector<TypeTrackletKey>& updatedIds) {
        int i = 0, j = 0, newCount = 0, oldCount = 0;
        for (const auto& featureNew : newFrames) {
            const TypeTrackletKey id = (TypeTrackletKey)featureNew._keyFrameId;
            if (featureNew.has_last) {
                StoredTracklet& t = _trackletMap[id];
                t.id = id;
                t.age = 0;
                t.points.push_front(FeaturePoint{(float)featureNew._featureLast.first, (float)featureNew._featureLast.second,
                                                 depthsLastFrame[(size_t)j]});
                j++;
                newCount++;
            } else {
                oldCount++;
            }
            _trackletMap[id].points.push_front(FeaturePoint{(float)featureNew._feature.first, (float)featureNew._feature.second,
                                                            depthsCurFrame[(size_t)i]});
            i++;
            updatedIds.push_back(id);
        }
        return std::make_pair(oldCount, newCount);
    }

    // :171-193
    void TidyUpTracklets(const std::vector<TypeTrackletKey>& updatedIds) {
        const std::set<TypeTrackletKey> keep(updatedIds.begin(), updatedIds.end());
        for (auto it = _trackletMap.begin(); it != _trackletMap.end();)
            it = keep.count(it->first) ? std::next(it) : _trackletMap.erase(it);
    }

    // :209-259
    std::pair<int, int> convert_tracklets_to_matches_msg(const MatchesMsgIn& tracklets_in,
                                                         const std::vector<TypeTrackletKey>& trackletIds, MatchesMsg& Out) {
        Out.tracks.clear();
        Out.tracks.reserve(trackletIds.size());
        int success = 0, failed = 0;
        for (const auto trackletId : trackletIds) {
            const StoredTracklet& cur = _trackletMap[trackletId];
            Tracklet t;
            t.id = cur.id;
            t.age = cur.age;
            t.feature_points.assign(cur.points.begin(), cur.points.end());
            for (const auto& p : t.feature_points) (p.d >= 0 ? success : failed)++;
            Out.tracks.push_back(std::move(t));
        }
        Out.stamps = tracklets_in.stamps;
        return std::make_pair(success, failed);
    }

    // tracklet_depth_module.h:109-123
    const Mono_Lidar::DepthCalculationStatistics& getDepthCalcStats() { return _depthEstimator.getDepthCalcStats(); }
    void getCloudCameraCs(Cloud::Ptr& pointCloud_cam_cs) { _depthEstimator.getCloudCameraCs(pointCloud_cam_cs); }
    void getCloudInterpolated(Cloud::Ptr& pointCloud_interpolated) { _depthEstimator.getCloudInterpolated(pointCloud_interpolated); }
    void getPointsCloudImageCs(std::vector<double>& visiblePointsImageCs) { _depthEstimator.getPointsCloudImageCs(visiblePointsImageCs); }
    size_t trackletCount() const { return _trackletMap.size(); }
    Mono_Lidar::DepthEstimator& depthEstimator() { return _depthEstimator; }

private:
    struct StoredTracklet {  // feature_tracking::Tracklet: id, age, matches newest first (push_front)
        TypeTrackletKey id = 0;
        int age = 0;
        std::deque<FeaturePoint> points;
    };
    void check(int rc) {
        if (rc == MLD_OK) return;
        if (rc == MLD_ERR_CLOUD_TOO_SMALL) throw Mono_Lidar::GroundPlane::ExceptionPclInvalid();
        throw std::runtime_error(std::string("TrackletDepthModule: ") + mld_last_error(_depthEstimator.ctx()));
    }
    Mono_Lidar::DepthEstimatorParameters depth_estimator_parameters_;
    std::map<TypeTrackletKey, StoredTracklet> _trackletMap;
    Mono_Lidar::GroundPlane::Ptr groundPlaneLast_;
    std::shared_ptr<CameraPinhole> _camera;
    Mono_Lidar::DepthEstimator _depthEstimator;
    bool _isDepthEstimatorInitialized = false;
    std::array<double, 12> _camLidarTransform{};
    int _slotCur = 0;
    bool _haveLast = false;
};

}  // namespace tracklets_depth

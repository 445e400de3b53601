// Drives the C++ TrackletDepthModule shim the way tracklets_depth_ros_tool drives the reference's module
// (tracklet_depth_interface.cpp callbacks -> TrackletDepthModule::process), frame by frame from files:
//   <dir>/cloud_<k>.bin   pcl::PointXYZI records (32 bytes)
//   <dir>/tracks_<k>.bin  n x { uint64 id; float u0, v0, u1, v1 }   (newest, previous feature)
//   <dir>/inl_<k>.bin     int32 ground-plane inliers (plane 0,0,1,1.73 as in the synthetic scenes), "-" = estimate
// and writes <dir>/out_<k>.bin: n x { float d_newest; float d_previous (NaN unless the track is new); int32 length }.
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "tracklets_depth/tracklet_depth_module.h"

using namespace Mono_Lidar;

template <typename T>
static std::vector<T> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<T> v((size_t)n / sizeof(T));
    f.read(reinterpret_cast<char*>(v.data()), n);
    return v;
}

struct TrackRec {
    uint64_t id;
    float u0, v0, u1, v1;
    float pad_;
};
static_assert(sizeof(TrackRec) == 32, "record layout");

// A GroundPlane subclass of the CALLER's (not one of the shim's GPU-estimated planes): it segments itself on the CPU
// when setInputCloud finds it un-segmented (DepthEstimator.cpp:281-283) - here from a prepared inlier list - or throws
// ExceptionPclInvalid, as RansacPlane does for an unusable cloud (RansacPlane.cpp:44-50).
class ForeignPlane : public GroundPlane {
public:
    ForeignPlane(std::vector<int> inliers, bool fail) : inl_(std::move(inliers)), fail_(fail) {}
    void CalculateInliersPlane(const Cloud::ConstPtr&, double, double) override {
        if (fail_) throw ExceptionPclInvalid();
        _modelCoeffs = {0.f, 0.f, 1.f, 1.73f};
        _inliersIndex = inl_;
        inliersChanged();  // (CheckPointInPlane's bitmask belongs to the previous list)
        is_segmented_ = true;
    }

private:
    std::vector<int> inl_;
    bool fail_;
};

int main(int argc, char** argv) {
    if (argc < 4) {
        std::cerr << "usage: mld_tracklet_demo <dir> <n_frames> <plane: 0 supplied | 1 estimated on the GPU | 2 a foreign "
                     "CPU-estimated plane that fails on frame 2>\n";
        return 2;
    }
    try {
        const std::string dir = argv[1];
        const int n_frames = std::atoi(argv[2]);
        const int plane_mode = std::atoi(argv[3]);
        const bool estimate = plane_mode == 1;
        DepthEstimatorParameters P;
        mld_params_c0(&P);
        tracklets_depth::TrackletDepthModule mod(P);
        mod.SetCamera(std::make_shared<CameraPinhole>(1242, 375, 721.5377, 609.5593, 172.854));
        mod.SetCameraLidarTransform({0, -1, 0, 0.0, 0, 0, -1, -0.08, 1, 0, 0, -0.27});
        for (int k = 0; k < n_frames; k++) {
            const std::string sk = std::to_string(k);
            auto cloud = std::make_shared<PointCloud>();
            cloud->points = slurp<PointXYZI>(dir + "/cloud_" + sk + ".bin");
            const std::vector<TrackRec> recs = slurp<TrackRec>(dir + "/tracks_" + sk + ".bin");
            tracklets_depth::MatchesMsgIn in;
            in.tracks.resize(recs.size());
            for (size_t i = 0; i < recs.size(); i++) {
                in.tracks[i].id = recs[i].id;
                in.tracks[i].feature_points = {{recs[i].u0, recs[i].v0}, {recs[i].u1, recs[i].v1}};
            }
            GroundPlane::Ptr gp;
            if (plane_mode == 2) {
                gp = std::make_shared<ForeignPlane>(slurp<int>(dir + "/inl_" + sk + ".bin"), k == 2);
            } else if (!estimate) {
                std::vector<int> inl = slurp<int>(dir + "/inl_" + sk + ".bin");
                gp = std::make_shared<GroundPlane>(std::array<float, 4>{0.f, 0.f, 1.f, 1.73f}, inl);
            }
            tracklets_depth::MatchesMsg out;
            PointCloud::ConstPtr ccloud = cloud;
            mod.process(ccloud, in, gp, &out);
            std::ofstream f(dir + "/out_" + sk + ".bin", std::ios::binary);
            for (size_t i = 0; i < out.tracks.size(); i++) {
                const auto& t = out.tracks[i];
                float rec[2] = {t.feature_points.at(0).d, t.feature_points.size() > 1 ? t.feature_points.at(1).d : NAN};
                int32_t len = (int32_t)t.feature_points.size();
                f.write(reinterpret_cast<const char*>(rec), sizeof(rec));
                f.write(reinterpret_cast<const char*>(&len), sizeof(len));
            }
            std::cout << "frame " << k << " tracks " << out.tracks.size() << " stored " << mod.trackletCount() << "\n";
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    } catch (const char* msg) {
        std::cerr << "error: " << msg << "\n";
        return 1;
    } catch (const std::string& msg) {
        std::cerr << "error: " << msg << "\n";
        return 1;
    }
    return 0;
}

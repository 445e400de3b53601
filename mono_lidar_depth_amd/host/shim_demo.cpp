// Drives the C++ shim exactly the way tracklets_depth does (tracklet_depth_module.cpp:401,413,80):
// InitConfig -> Initialize -> CalculateDepth(cloud, uv, depths, groundPlane).
//
// usage: mld_shim_demo <params.yaml|-> <cloud.bin> <uv.bin> <inliers.bin|-> <out.bin>
//   cloud.bin  : N x 8 float32 (pcl::PointXYZI layout)     uv.bin : F x 2 float64 (= 2 x F column-major)
//   inliers.bin: int32 indices; plane coefficients are fixed to (0,0,1,1.73)
//   out.bin    : F float64 depths followed by F int32 result types
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>

#include "monolidar_fusion/DepthEstimator.h"

template <typename T>
static std::vector<T> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<T> out(raw.size() / sizeof(T));
    std::memcpy(out.data(), raw.data(), out.size() * sizeof(T));
    return out;
}

int main(int argc, char** argv) {
    if (argc != 6) {
        std::cerr << "usage: mld_shim_demo <params.yaml|-> <cloud.bin> <uv.bin> <inliers.bin|-> <out.bin>\n";
        return 2;
    }
    try {
        using namespace Mono_Lidar;
        DepthEstimator est(0);
        if (std::strcmp(argv[1], "-") == 0) {
            auto p = std::make_shared<DepthEstimatorParameters>();
            mld_params_c0(p.get());
            est.InitConfig(p);
        } else {
            est.InitConfig(std::string(argv[1]), false);
        }
        auto cam = std::make_shared<CameraPinhole>(1242, 375, 721.5377, 609.5593, 172.854);
        std::array<double, 12> T = {0, -1, 0, 0.0, 0, 0, -1, -0.08, 1, 0, 0, -0.27};
        est.Initialize(cam, T);

        auto cloud = std::make_shared<PointCloud>();
        cloud->points = slurp<PointXYZI>(argv[2]);
        std::vector<double> uv = slurp<double>(argv[3]);
        GroundPlane::Ptr gp;
        if (std::strcmp(argv[4], "-") != 0) {
            std::vector<int> inl = slurp<int>(argv[4]);
            gp = std::make_shared<GroundPlane>(std::array<float, 4>{0.f, 0.f, 1.f, 1.73f}, inl);
        } else {
            est.getParameters()->do_use_ransac_plane = 0;
            est.Initialize(cam, T);
        }
        std::vector<double> depths;
        std::vector<int> types;
        PointCloud::ConstPtr ccloud = cloud;
        est.CalculateDepth(ccloud, uv, depths, types, gp);

        std::ofstream out(argv[5], std::ios::binary);
        out.write(reinterpret_cast<const char*>(depths.data()), (std::streamsize)(depths.size() * sizeof(double)));
        out.write(reinterpret_cast<const char*>(types.data()), (std::streamsize)(types.size() * sizeof(int)));
        auto stats = DepthEstimator::getDepthCalcStats(types);
        const DepthCalculationStatistics& st = est.getDepthCalcStats();
        if (st.getPointCount() != (int)types.size() || st.getSuccess() != (int)stats[Success] ||
            st.getSuccessRoad() != (int)stats[SuccessRoad] ||
            st.getRadiusSearchInsufficientPoints() != (int)stats[RadiusSearchInsufficientPoints])
            throw std::runtime_error("getDepthCalcStats() disagrees with the result types");
        std::cout << "features " << types.size() << " success " << stats[Success] << " road " << stats[SuccessRoad]
                  << " insufficient " << stats[RadiusSearchInsufficientPoints] << "\n";
        // debug mode: same results, plus the debug clouds of the reference's getters
        est.ActivateDebugMode();
        std::vector<double> depths_dbg;
        std::vector<int> types_dbg;
        est.CalculateDepth(uv, depths_dbg, types_dbg, gp);
        auto corners = std::make_shared<PointCloud>(), plane = std::make_shared<PointCloud>(),
             interp = std::make_shared<PointCloud>(), nb = std::make_shared<PointCloud>(),
             camcs = std::make_shared<PointCloud>();
        est.getCloudTriangleCorners(corners);
        est.getCloudRansacPlane(plane);
        est.getCloudInterpolated(interp);
        est.getCloudNeighbors(nb);
        est.getCloudCameraCs(camcs);
        size_t valid = 0;
        for (double d : depths_dbg) valid += d >= 0 ? 1 : 0;
        std::cout << "debug same_types " << (types_dbg == types ? 1 : 0) << " corners " << corners->points.size()
                  << " plane " << plane->points.size() << " interpolated " << interp->points.size() << " valid " << valid
                  << " neighbors " << nb->points.size() << " camcs " << camcs->points.size() << "\n";
        // semantic ground plane (the 4-argument TrackletDepthModule::process, tracklet_depth_module.cpp:270-284): a
        // label image with every pixel "road" makes every projectable point a candidate
        if (est.getParameters()->do_use_ransac_plane) {
            std::vector<uint8_t> labels((size_t)375 * 1242, 7);
            GroundPlane::Ptr sem = std::make_shared<SemanticPlane>(labels.data(), 375, 1242, 1242, std::set<int>{6, 7, 8, 9},
                                                                   est.getParameters()->ransac_plane_refinement_treshold);
            std::vector<double> d3;
            std::vector<int> t3;
            est.CalculateDepth(ccloud, uv, d3, t3, sem);  // ONE C call: plane estimated on the GPU ahead of the projection
            auto* lazy = static_cast<RansacPlane*>(sem.get());
            const int pending = lazy->inliersPending() ? 1 : 0;  // the index list has not left the GPU yet
            const long long counted = (long long)lazy->numInliers();
            // a fresh RansacPlane on the next frame reuses the slot: the semantic plane's list is fetched just before
            GroundPlane::Ptr rp = std::make_shared<RansacPlane>(est.getParameters(), 5u);
            std::vector<double> d4;
            std::vector<int> t4;
            est.CalculateDepth(ccloud, uv, d4, t4, rp);
            const int flushed = lazy->inliersPending() ? 0 : 1;
            std::cout << "semantic segmented " << (sem->isSegmented() ? 1 : 0) << " inliers " << sem->getInlinersIndex().size()
                      << " nz " << sem->getModelCoeffs()[2] << " lazy " << pending << " counted " << counted << " flushed "
                      << flushed << " ransac_pending " << (static_cast<RansacPlane*>(rp.get())->inliersPending() ? 1 : 0)
                      << " ransac_inliers " << rp->getInlinersIndex().size() << " ransac_counted "
                      << static_cast<RansacPlane*>(rp.get())->numInliers() << "\n";
        }
        // The reference decides about the road fallback per call (DepthEstimator.cpp:580): a null plane switches it off
        // for that call only, the plane handed in afterwards is used again, and so is a DIFFERENT plane object.
        if (est.getParameters()->do_use_ransac_plane) {
            DepthEstimator e2(0);
            auto p2 = std::make_shared<DepthEstimatorParameters>();
            mld_params_c0(p2.get());
            e2.InitConfig(p2);
            e2.Initialize(cam, T);
            std::vector<double> da, db, dc, dd;
            std::vector<int> ta, tb, tc, td;
            e2.CalculateDepth(ccloud, uv, da, ta, gp);
            GroundPlane::Ptr none;
            e2.CalculateDepth(uv, db, tb, none);
            e2.CalculateDepth(uv, dc, tc, gp);
            std::vector<int> half;
            for (size_t i = 0; i < gp->getInlinersIndex().size(); i += 2) half.push_back(gp->getInlinersIndex()[i]);
            GroundPlane::Ptr other = std::make_shared<GroundPlane>(std::array<float, 4>{0.f, 0.f, 1.f, 1.73f}, half);
            e2.CalculateDepth(uv, dd, td, other);
            size_t road_b = 0, road_d = 0;
            for (int t : tb) road_b += t == SuccessRoad;
            for (int t : td) road_d += t == SuccessRoad;
            std::cout << "percall same_after_null " << ((ta == tc && da == dc) ? 1 : 0) << " road_with_null " << road_b
                      << " road_other " << road_d << " road_first " << stats[SuccessRoad] << "\n";
        }
        // usage error as in the reference: CalculateDepth before setInputCloud
        DepthEstimator fresh(0);
        fresh.InitConfig();
        bool threw = false;
        try {
            fresh.Initialize(cam, T);
            std::vector<double> d2;
            fresh.CalculateDepth(uv, d2, gp);
        } catch (const char* msg) {
            threw = std::string(msg) == "call of 'CalculateDepth' without 'SetInputCloud'";
        }
        std::cout << "usage_error_ok " << (threw ? 1 : 0) << "\n";
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

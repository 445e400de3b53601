// COMPILE EVIDENCE, not parity evidence: drives the Eigen-typed overloads of the C++ shim - the ones the reference's
// tracklets_depth binds (tracklet_depth_module.cpp:63-117: CalculateDepth(cloud, Eigen::Matrix2Xd, Eigen::VectorXd&,
// GroundPlane::Ptr&)) - against the tests-only stand-in tests/stubs/Eigen (this image has no Eigen).  Built and run by
// tests/test_eigen_overloads.py only.
//
// usage: eigen_overloads_demo <cloud.bin> <uv.bin> <inliers.bin> <out.bin>
//   cloud.bin : N x 8 float32 (pcl::PointXYZI layout)   uv.bin : F x 2 float64   inliers.bin : int32 indices
//   out.bin   : F float64 depths (4-argument overload) | F float64 depths + F int32 types (5-argument overload)
//               | F float64 depths + F int32 types (feature-only overload) | 1 float64 + 1 int32 (single feature)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>

#include "monolidar_fusion/DepthEstimator.h"

#ifndef MLD_HAVE_EIGEN
#error "the Eigen overloads are not compiled: <Eigen/Core> was not found on the include path"
#endif

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
    if (argc != 5) return 2;
    try {
        using namespace Mono_Lidar;
        DepthEstimator est(0);
        auto p = std::make_shared<DepthEstimatorParameters>();
        mld_params_c0(p.get());
        est.InitConfig(p);
        auto cam = std::make_shared<CameraPinhole>(1242, 375, 721.5377, 609.5593, 172.854);
        Eigen::Affine3d T;  // lidar -> camera, as tracklets_depth hands it over (tracklet_depth_module.cpp:401-413)
        const double Tm[12] = {0, -1, 0, 0.0, 0, 0, -1, -0.08, 1, 0, 0, -0.27};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) T.matrix()(r, c) = Tm[r * 4 + c];
        est.Initialize(cam, T);

        auto cloud = std::make_shared<PointCloud>();
        cloud->points = slurp<PointXYZI>(argv[1]);
        const std::vector<double> uv = slurp<double>(argv[2]);
        const int frameCount = (int)(uv.size() / 2);
        GroundPlane::Ptr ransacPlane =
            std::make_shared<GroundPlane>(std::array<float, 4>{0.f, 0.f, 1.f, 1.73f}, slurp<int>(argv[3]));
        PointCloud::ConstPtr cloud_in_cur = cloud;

        // tracklet_depth_module.cpp:63-82, literally
        Eigen::VectorXd depths;
        depths.resize(frameCount);
        Eigen::Matrix2Xd featureCoordinates(2, frameCount);
        for (int i = 0; i < frameCount; i++) {
            featureCoordinates(0, i) = uv[2 * i];
            featureCoordinates(1, i) = uv[2 * i + 1];
        }
        est.CalculateDepth(cloud_in_cur, featureCoordinates, depths, ransacPlane);

        // the overload with result types (DepthEstimator.h:190-196 of the reference)
        Eigen::VectorXd depths5;
        Eigen::VectorXi types5;
        est.CalculateDepth(cloud_in_cur, featureCoordinates, depths5, types5, ransacPlane);

        // feature-only overloads on the cloud already set (DepthEstimator.cpp:421-488)
        Eigen::VectorXd depthsF;
        Eigen::VectorXi typesF;
        est.CalculateDepth(featureCoordinates, depthsF, typesF, ransacPlane);
        Eigen::VectorXd depthsF3;
        est.CalculateDepth(featureCoordinates, depthsF3, ransacPlane);
        const auto one = est.CalculateDepth(Eigen::Vector2d(uv[0], uv[1]), ransacPlane);

        std::ofstream out(argv[4], std::ios::binary);
        out.write(reinterpret_cast<const char*>(depths.data()), sizeof(double) * frameCount);
        out.write(reinterpret_cast<const char*>(depths5.data()), sizeof(double) * frameCount);
        out.write(reinterpret_cast<const char*>(types5.data()), sizeof(int) * frameCount);
        out.write(reinterpret_cast<const char*>(depthsF.data()), sizeof(double) * frameCount);
        out.write(reinterpret_cast<const char*>(typesF.data()), sizeof(int) * frameCount);
        const double d1 = one.second;
        const int t1 = (int)one.first;
        out.write(reinterpret_cast<const char*>(&d1), sizeof(double));
        out.write(reinterpret_cast<const char*>(&t1), sizeof(int));
        bool same3 = depthsF3.size() == depthsF.size();
        for (int i = 0; same3 && i < frameCount; i++)
            same3 = (depthsF3[i] == depthsF[i]) || (depthsF3[i] != depthsF3[i] && depthsF[i] != depthsF[i]);
        std::cout << "eigen_overloads ok features " << frameCount << " same3 " << (same3 ? 1 : 0) << "\n";
        return 0;
    } catch (const char* e) {
        std::cerr << "error: " << e << "\n";
    } catch (const std::string& e) {
        std::cerr << "error: " << e << "\n";
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
    }
    return 1;
}

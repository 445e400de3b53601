// COMPILE EVIDENCE, not parity evidence: the shim's boundary in the reference's OWN types - Cloud =
// pcl::PointCloud<pcl::PointXYZI> (reference DepthEstimator.h:62-63), SemanticPlane(const cv::Mat&, Camera, std::set<int>,
// double) (RansacPlane.h:173-193), Eigen feature / depth containers - driven the way tracklets_depth drives it
// (tracklet_depth_module.cpp:63-117: the two CalculateDepth(cloud, Matrix2Xd, VectorXd&, GroundPlane::Ptr&) calls;
// :269-284: a fresh SemanticPlane per frame from the label image, the camera intrinsics and _camLidarTransform).
// Compiled against the tests-only stand-ins tests/stubs/{pcl,opencv2,Eigen} (this image has none of the three); built
// and run by tests/test_pcl_cv_boundary.py only.
//
// usage: pcl_cv_boundary_demo <cloud.bin> <cloud_last.bin> <uv.bin> <labels.bin> <rows> <cols> <out.bin>
//   cloud*.bin : N x 8 float32 (pcl::PointXYZI records)   uv.bin : F x 2 float64   labels.bin : rows x cols uint8
//   out.bin    : F float64 depths of the current frame | F float64 depths of the previous frame | 4 float32 plane
//                coefficients | int32 inlier count | int32 CheckPointInPlane hits over all points
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <set>

#include "monolidar_fusion/DepthEstimator.h"

#if !defined(MLD_HAVE_PCL) || !defined(MLD_HAVE_OPENCV) || !defined(MLD_HAVE_EIGEN)
#error "the pcl / cv::Mat / Eigen overloads are not compiled: the headers were not found on the include path"
#endif
static_assert(std::is_same<Mono_Lidar::DepthEstimator::Cloud, pcl::PointCloud<pcl::PointXYZI>>::value,
              "DepthEstimator::Cloud must be the caller's pcl cloud type");

template <typename T>
static std::vector<T> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<T> out(raw.size() / sizeof(T));
    std::memcpy(static_cast<void*>(out.data()), raw.data(), out.size() * sizeof(T));
    return out;
}

using Cloud = pcl::PointCloud<pcl::PointXYZI>;

static Cloud::ConstPtr load_cloud(const char* path) {
    Cloud::Ptr cloud(new Cloud());
    cloud->points = slurp<pcl::PointXYZI>(path);
    cloud->width = (std::uint32_t)cloud->points.size();
    cloud->height = 1;
    return cloud;
}

int main(int argc, char** argv) {
    if (argc != 8) return 2;
    try {
        Mono_Lidar::DepthEstimator _depthEstimator(0);
        auto parameters = std::make_shared<Mono_Lidar::DepthEstimatorParameters>();
        mld_params_c0(parameters.get());
        _depthEstimator.InitConfig(parameters);
        auto _camera = std::make_shared<CameraPinhole>(1242, 375, 721.5377, 609.5593, 172.854);
        Eigen::Affine3d _camLidarTransform;
        const double Tm[12] = {0, -1, 0, 0.0, 0, 0, -1, -0.08, 1, 0, 0, -0.27};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) _camLidarTransform.matrix()(r, c) = Tm[r * 4 + c];
        _depthEstimator.Initialize(_camera, _camLidarTransform);

        const Cloud::ConstPtr cloud_in = load_cloud(argv[1]);
        const Cloud::ConstPtr _cloud_last_frame = load_cloud(argv[2]);
        const std::vector<double> uv = slurp<double>(argv[3]);
        std::vector<std::uint8_t> labels = slurp<std::uint8_t>(argv[4]);
        const int rows = std::atoi(argv[5]), cols = std::atoi(argv[6]);
        const int frameCount = (int)(uv.size() / 2);

        // process(): the frame's ground plane object, built from the label image (tracklet_depth_module.cpp:269-284)
        Mono_Lidar::GroundPlane::Ptr gp;
        Mono_Lidar::SemanticPlane::Camera cam;
        cam.f = 721.5377;
        cam.cu = 609.5593;
        cam.cv = 172.854;
        cam.transform_cam_lidar = _camLidarTransform;
        const cv::Mat image(rows, cols, CV_8UC1, labels.data(), (size_t)cols);
        std::set<int> gp_labels{6, 7, 8, 9};
        double plane_inlier_threshold = parameters->ransac_plane_refinement_treshold;
        gp = std::make_shared<Mono_Lidar::SemanticPlane>(image, cam, gp_labels, plane_inlier_threshold);

        // CalculateFeatureDepthsCurFrame (:63-82): features into a 2 x n matrix, depths out of a vector the callee resizes
        Eigen::VectorXd depthsCurFrame;
        depthsCurFrame.resize(frameCount);
        Eigen::Matrix2Xd featureCoordinates(2, frameCount);
        for (int i = 0; i < frameCount; i++) {
            featureCoordinates(0, i) = uv[2 * i];
            featureCoordinates(1, i) = uv[2 * i + 1];
        }
        _depthEstimator.CalculateDepth(cloud_in, featureCoordinates, depthsCurFrame, gp);
        if (gp == nullptr || !gp->isSegmented()) throw std::runtime_error("plane not calculated");
        const std::array<float, 4> coeffs = gp->getModelCoeffs();
        const int n_inliers = (int)gp->getInlinersIndex().size();
        int hits = 0;
        for (int i = 0; i < (int)cloud_in->points.size(); i++) hits += gp->CheckPointInPlane(i) ? 1 : 0;

        // CalculateFeatureDepthsLastFrame (:84-117): the previous cloud with ITS plane object (a fresh one here)
        Mono_Lidar::GroundPlane::Ptr groundPlaneLast_;  // null: a RansacPlane is created and estimated (DepthEstimator.cpp:275-283)
        Eigen::VectorXd depthsLastFrame;
        depthsLastFrame.resize(frameCount);
        _depthEstimator.CalculateDepth(_cloud_last_frame, featureCoordinates, depthsLastFrame, groundPlaneLast_);
        if (groundPlaneLast_ == nullptr || !groundPlaneLast_->isSegmented()) throw std::runtime_error("old plane not calculated");

        // a SemanticPlane::Camera other than the estimator's calibration is refused, not silently replaced
        bool refused = false;
        try {
            Mono_Lidar::SemanticPlane::Camera other = cam;
            other.f = 700.0;
            Mono_Lidar::GroundPlane::Ptr gp2 = std::make_shared<Mono_Lidar::SemanticPlane>(image, other, gp_labels, plane_inlier_threshold);
            Eigen::VectorXd d2;
            _depthEstimator.CalculateDepth(cloud_in, featureCoordinates, d2, gp2);
        } catch (const std::runtime_error&) {
            refused = true;
        }
        // a label image that is not MONO8 is refused by the constructor
        bool refused_type = false;
        try {
            std::vector<std::uint8_t> rgb((size_t)rows * cols * 3);
            const cv::Mat bad(rows, cols, CV_8UC3, rgb.data(), (size_t)cols * 3);
            Mono_Lidar::SemanticPlane sp(bad, cam, gp_labels, plane_inlier_threshold);
        } catch (const std::runtime_error&) {
            refused_type = true;
        }

        std::ofstream out(argv[7], std::ios::binary);
        out.write(reinterpret_cast<const char*>(depthsCurFrame.data()), sizeof(double) * frameCount);
        out.write(reinterpret_cast<const char*>(depthsLastFrame.data()), sizeof(double) * frameCount);
        out.write(reinterpret_cast<const char*>(coeffs.data()), sizeof(float) * 4);
        out.write(reinterpret_cast<const char*>(&n_inliers), sizeof(int));
        out.write(reinterpret_cast<const char*>(&hits), sizeof(int));
        std::cout << "pcl_cv_boundary ok features " << frameCount << " points " << cloud_in->points.size() << " inliers "
                  << n_inliers << " hits " << hits << " refused " << (refused ? 1 : 0) << (refused_type ? 1 : 0) << std::endl;
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "exception: " << e.what() << std::endl;
    } catch (const char* e) {
        std::cerr << "exception: " << e << std::endl;
    } catch (const std::string& e) {
        std::cerr << "exception: " << e << std::endl;
    }
    return 1;
}

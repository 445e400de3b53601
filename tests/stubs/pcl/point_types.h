// TESTS-ONLY minimal stand-in for <pcl/point_types.h>.  NOT PCL, not a reference build, not parity evidence: it exists
// so that the pcl-typed boundary of the C++ shim (mono_lidar_depth_amd/host/monolidar_fusion/DepthEstimator.h,
// `#ifdef MLD_HAVE_PCL`: Cloud = pcl::PointCloud<pcl::PointXYZI>, reference DepthEstimator.h:62-63) is compiled and
// called at least once in this image, which has no PCL.  Only the members the shim and its call sites touch are
// provided; the record has PCL's documented memory layout (x,y,z,pad | intensity,pad,pad,pad: 32 bytes, 16-byte aligned).
#pragma once

namespace pcl {

struct alignas(16) PointXYZI {
    union {
        float data[4];
        struct {
            float x, y, z;
        };
    };
    union {
        struct {
            float intensity;
        };
        float data_c[4];
    };
    PointXYZI() : data{0.f, 0.f, 0.f, 1.f}, data_c{0.f, 0.f, 0.f, 0.f} {}
};

}  // namespace pcl

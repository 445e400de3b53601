// TESTS-ONLY minimal stand-in for <pcl/point_cloud.h> (see tests/stubs/pcl/point_types.h): pcl::PointCloud<PointT> with
// the members the shim and the reference's call sites touch - `points` (contiguous), size(), push_back, Ptr / ConstPtr
// (std::shared_ptr, as PCL >= 1.11; older PCL uses boost::shared_ptr - the shim only ever names Cloud::Ptr / ::ConstPtr).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

namespace pcl {

template <typename PointT>
class PointCloud {
public:
    using Ptr = std::shared_ptr<PointCloud<PointT>>;
    using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
    std::vector<PointT> points;
    std::uint32_t width = 0, height = 0;
    bool is_dense = true;
    std::size_t size() const { return points.size(); }
    void push_back(const PointT& p) {
        points.push_back(p);
        width = (std::uint32_t)points.size();
        height = 1;
    }
    void clear() {
        points.clear();
        width = height = 0;
    }
};

}  // namespace pcl

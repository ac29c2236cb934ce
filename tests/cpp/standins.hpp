// Minimal stand-ins for the third-party TYPES the reference's frameToFrame signature mentions (pcl::PointXYZ,
// pcl::PointCloud<>::Ptr, pcl::KdTreeFLANN, cv::Point2f, Eigen::Matrix4d).  They exist only so that OUR adaptor
// (include/velo_frame_to_frame.hpp) can be exercised in an image that has none of PCL / OpenCV / Eigen; no reference
// source is compiled against them.
#pragma once
#include <memory>
#include <vector>

namespace standin {
struct PointXYZ { float x, y, z, pad; PointXYZ() : x(0), y(0), z(0), pad(1) {} PointXYZ(float a, float b, float c) : x(a), y(b), z(c), pad(1) {} };
struct PointCloud {
    std::vector<PointXYZ> points;
    typedef std::shared_ptr<PointCloud> Ptr;
    size_t size() const { return points.size(); }
    const PointXYZ& at(size_t i) const { return points.at(i); }
    void push_back(const PointXYZ& p) { points.push_back(p); }
};
struct KdTree {};                      // the adaptor ignores the trees
struct Point2f { float x, y; };
struct Matrix4d {
    double m[16];
    double& operator()(int i, int j) { return m[i * 4 + j]; }
    double operator()(int i, int j) const { return m[i * 4 + j]; }
};
}  // namespace standin

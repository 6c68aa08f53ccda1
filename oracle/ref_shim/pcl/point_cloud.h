// Stand-in for the PCL types named by the reference's teaser_utils headers (PCL is not installed).  Test infrastructure.
#pragma once
#include <memory>
#include <vector>
#include <Eigen/Core>
namespace pcl {
template <typename PointT>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  std::vector<PointT> points;
  typename std::vector<PointT>::iterator begin() { return points.begin(); }
  typename std::vector<PointT>::iterator end() { return points.end(); }
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
  size_t size() const { return points.size(); }
  void push_back(const PointT& p) { points.push_back(p); }
};
}  // namespace pcl

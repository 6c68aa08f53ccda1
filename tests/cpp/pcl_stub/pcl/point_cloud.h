// Stand-in for <pcl/point_cloud.h> (PCL 1.10): aligned-allocator storage, boost::shared_ptr Ptr / ConstPtr, header fields.
#pragma once
#include <cstdint>
#include <vector>

#include <Eigen/Core>
#include <boost/make_shared.hpp>
#include <boost/shared_ptr.hpp>
namespace pcl {
template <typename PointT>
class PointCloud {
 public:
  using PointType = PointT;
  using VectorType = std::vector<PointT, Eigen::aligned_allocator<PointT>>;
  using Ptr = boost::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = boost::shared_ptr<const PointCloud<PointT>>;
  using iterator = typename VectorType::iterator;
  using const_iterator = typename VectorType::const_iterator;
  VectorType points;
  std::uint32_t width = 0, height = 0;
  bool is_dense = true;
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void reserve(std::size_t n) { points.reserve(n); }
  void resize(std::size_t n) {
    points.resize(n);
    width = static_cast<std::uint32_t>(n);
    height = 1;
  }
  void clear() {
    points.clear();
    width = height = 0;
  }
  void push_back(const PointT& p) {
    points.push_back(p);
    width = static_cast<std::uint32_t>(points.size());
    height = 1;
  }
  PointT& operator[](std::size_t i) { return points[i]; }
  const PointT& operator[](std::size_t i) const { return points[i]; }
  PointT& at(std::size_t i) { return points.at(i); }
  const PointT& at(std::size_t i) const { return points.at(i); }
  iterator begin() { return points.begin(); }
  iterator end() { return points.end(); }
  const_iterator begin() const { return points.begin(); }
  const_iterator end() const { return points.end(); }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};
}  // namespace pcl

// stand-in for the two PCL types the extracted functions name: a point with x, y, z and a cloud with `points`
#pragma once
#include <memory>
#include <vector>
namespace pcl {
struct PointXYZ {
  float x, y, z;
};
template <typename T>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<T>> Ptr;
  typedef std::shared_ptr<const PointCloud<T>> ConstPtr;
  std::vector<T> points;
};
}  // namespace pcl

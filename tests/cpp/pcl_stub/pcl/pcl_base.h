#pragma once
#include <pcl/point_cloud.h>
namespace pcl {
template <typename PointT>
class PCLBase {
 public:
  using PointCloud = pcl::PointCloud<PointT>;
  using PointCloudPtr = typename PointCloud::Ptr;
  using PointCloudConstPtr = typename PointCloud::ConstPtr;
  virtual ~PCLBase() = default;
  virtual void setInputCloud(const PointCloudConstPtr& cloud) { input_ = cloud; }
  const PointCloudConstPtr getInputCloud() const { return input_; }

 protected:
  PointCloudConstPtr input_;
};
}  // namespace pcl

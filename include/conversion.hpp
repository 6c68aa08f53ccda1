// conversion.hpp — drop-in for url-kaist/Quatro's include/conversion.hpp: the container conversions the path's callers
// use (pcl2teaser :13-18 — FPFHManager::setFeaturePair include/fpfh_manager.hpp:112,119; pcl2eigen :38-44 —
// Quatro::computeTransformation include/quatro.hpp:774-775; eigen2pcl :47-57 — include/fpfh_manager.hpp:151-152;
// xyzi2xyz :29-36).  cloud2msg (:21-26) builds a sensor_msgs::PointCloud2 for the RViz publishers of the demo: ROS is out
// of this back end's scope, so it is not provided here.
// Host code only; the functions use element access, so they compile against real Eigen / PCL and against the stand-ins
// of quatro.hpp alike.
#ifndef CONVERSION_HPP
#define CONVERSION_HPP

#include "quatro.hpp"  // pcl:: / Eigen:: types (the real ones when installed, otherwise the stand-ins)
#include "teaser/geometry.h"

template <typename T>
void pcl2teaser(const pcl::PointCloud<T>& pcl_raw, teaser::PointCloud& cloud) {  // :13-18
  cloud.clear();
  cloud.reserve(pcl_raw.points.size());
  for (const auto& pt : pcl_raw.points) cloud.push_back({pt.x, pt.y, pt.z});
}

inline void xyzi2xyz(QUATRO_SHARED_PTR<pcl::PointCloud<pcl::PointXYZI>> XYZI,
                     QUATRO_SHARED_PTR<pcl::PointCloud<pcl::PointXYZ>> XYZ) {  // :29-36
  XYZ->points.resize(XYZI->points.size());
  for (size_t i = 0; i < XYZI->points.size(); ++i) {
    XYZ->points[i].x = XYZI->points[i].x;
    XYZ->points[i].y = XYZI->points[i].y;
    XYZ->points[i].z = XYZI->points[i].z;
  }
}

template <typename T>
void pcl2eigen(const pcl::PointCloud<T>& pcl_raw, Eigen::Matrix<double, 3, Eigen::Dynamic>& cloud) {  // :38-44
  const int N = static_cast<int>(pcl_raw.points.size());
  cloud.resize(3, N);
  for (int i = 0; i < N; ++i) {
    cloud(0, i) = pcl_raw.points[static_cast<size_t>(i)].x;
    cloud(1, i) = pcl_raw.points[static_cast<size_t>(i)].y;
    cloud(2, i) = pcl_raw.points[static_cast<size_t>(i)].z;
  }
}

template <typename T>
void eigen2pcl(const Eigen::Matrix<double, 3, Eigen::Dynamic>& src, pcl::PointCloud<T>& cloud) {  // :47-57
  const int num_pc = static_cast<int>(src.cols());
  T pt_tmp;
  if (!cloud.empty()) cloud.clear();
  for (int i = 0; i < num_pc; ++i) {
    pt_tmp.x = static_cast<float>(src(0, i));
    pt_tmp.y = static_cast<float>(src(1, i));
    pt_tmp.z = static_cast<float>(src(2, i));
    cloud.points.emplace_back(pt_tmp);
  }
}

#endif  // CONVERSION_HPP

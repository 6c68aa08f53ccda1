// teaser_utils/fpfh.h — drop-in for url-kaist/Quatro's include/teaser_utils/fpfh.h: class
// teaser::FPFHEstimation (:25-87) whose computeFPFHFeatures (src/teaser_utils/fpfh.cc:17-75: pcl::NormalEstimation +
// pcl::FPFHEstimationOMP over one kd-tree) becomes one qtr_fpfh call (grid search, normals, SPFH and FPFH kernels).
#pragma once
#include <memory>
#include <vector>

#include "../quatro.hpp"
#include "../teaser/geometry.h"

#ifndef QUATRO_HAVE_PCL
namespace pcl {
struct Normal {
  float normal_x = 0, normal_y = 0, normal_z = 0, curvature = 0;
};
struct FPFHSignature33 {
  float histogram[33] = {0};
  static int descriptorSize() { return 33; }
};
}  // namespace pcl
#else
#include <pcl/point_types.h>
#endif

namespace teaser {

using FPFHCloud = pcl::PointCloud<pcl::FPFHSignature33>;
using FPFHCloudPtr = QUATRO_SHARED_PTR<pcl::PointCloud<pcl::FPFHSignature33>>;

class FPFHEstimation {
 public:
  FPFHEstimation() = default;

  // 3-argument form of the reference (:39-40): the normals are computed and dropped
  FPFHCloudPtr computeFPFHFeatures(const PointCloud& input_cloud, double normal_search_radius = 0.03,
                                   double fpfh_search_radius = 0.05) {
    pcl::PointCloud<pcl::Normal> normals;
    return computeFPFHFeatures(input_cloud, normals, normal_search_radius, fpfh_search_radius);
  }

  // 4-argument form (:44-46; the one FPFHManager uses): descriptors returned, normals written to `normals`
  FPFHCloudPtr computeFPFHFeatures(const PointCloud& input_cloud, pcl::PointCloud<pcl::Normal>& normals,
                                   double normal_search_radius = 0.03, double fpfh_search_radius = 0.05) {
    const int n = static_cast<int>(input_cloud.size());
    FPFHCloudPtr out(new FPFHCloud());
    normals.points.assign(static_cast<size_t>(n), pcl::Normal());
    out->points.assign(static_cast<size_t>(n), pcl::FPFHSignature33());
    if (n == 0) return out;
    std::vector<float> xyz4(static_cast<size_t>(4) * n, 0.f), nrm(static_cast<size_t>(4) * n), desc(static_cast<size_t>(33) * n);
    for (int i = 0; i < n; ++i) {
      xyz4[4 * static_cast<size_t>(i)] = input_cloud[static_cast<size_t>(i)].x;
      xyz4[4 * static_cast<size_t>(i) + 1] = input_cloud[static_cast<size_t>(i)].y;
      xyz4[4 * static_cast<size_t>(i) + 2] = input_cloud[static_cast<size_t>(i)].z;
    }
    qtr_handle* h = quatro_hip::default_handle();
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    quatro_hip::check(h, qtr_fpfh(h, slot_lease.slot, xyz4.data(), n, static_cast<float>(normal_search_radius),
                                  static_cast<float>(fpfh_search_radius), nrm.data(), desc.data(), QTR_MEM_HOST));
    for (int i = 0; i < n; ++i) {
      pcl::Normal& q = normals.points[static_cast<size_t>(i)];
      q.normal_x = nrm[4 * static_cast<size_t>(i)];
      q.normal_y = nrm[4 * static_cast<size_t>(i) + 1];
      q.normal_z = nrm[4 * static_cast<size_t>(i) + 2];
      q.curvature = nrm[4 * static_cast<size_t>(i) + 3];
      for (int k = 0; k < 33; ++k) out->points[static_cast<size_t>(i)].histogram[k] = desc[33 * static_cast<size_t>(i) + k];
    }
    return out;
  }
};

}  // namespace teaser

// imageProjection.hpp — drop-in for url-kaist/Quatro's include/imageProjection.hpp (class ImageProjection, :31-581,
// LeGO-LOAM lineage): range-image projection of a scan and rejection of small sub-clusters, the stage the demo runs
// right before voxelisation (examples/run_global_registration.cpp:124-160).  segmentCloud() is one call into
// libquatro_hip.so (qtr_segment_cloud: projection, union-find component labelling, the reference's validity rule,
// compaction — quatro_amd/csrc/segment.hip); the getters hand back what the reference's getters do.
//
// Differences a caller can observe:
//   * ground handling is the "Patchwork" mode of the reference (:286-290): the input is expected to hold the
//     non-ground points already; "LeGO-LOAM" (the class's own ground removal, :366-422) is not implemented and the
//     constructor throws std::invalid_argument for it.
//   * no ROS publishers, no cv::Mat: getrangeMat() is not provided; getLabelMat() returns the label image as a
//     std::vector<int> (row-major n_scan x horizon_scan: -1 no return, 999999 rejected, >= 1 segment label).
#ifndef IMAGE_PROJECTION_HPP
#define IMAGE_PROJECTION_HPP

#include <stdexcept>
#include <string>
#include <vector>

#include "quatro.hpp"

class ImageProjection {
 public:
  ImageProjection() : ImageProjection("Velodyne-64-HDE", "4CrossNeighbor", "Patchwork") {}
  ImageProjection(const std::string& lidarType, const std::string& neighborSelectionMode,
                  const std::string& groundSegmentationMode, int numSubclusteringCriteria = 30) {
    if (qtr_ip_default_params(lidarType.c_str(), neighborSelectionMode.c_str(), &ip_) != QTR_OK) {
      // the reference has one message per cause (:131, :140); which one applies is decided the same way
      qtr_ip_params probe;
      if (qtr_ip_default_params(lidarType.c_str(), "4Neighbor", &probe) != QTR_OK)
        throw std::invalid_argument("[ImageProjection]:Check your paramter. Lidar Type is wrong!");
      throw std::invalid_argument("[ImageProjection]:Check your paramter. Neighbor selection mode is wrong!");
    }
    if (groundSegmentationMode != "LeGO-LOAM" && groundSegmentationMode != "Patchwork")
      throw std::invalid_argument("[ImageProjection]: Check your paramter. Ground Segmentation mode is wrong!");  // :144-146
    if (groundSegmentationMode == "LeGO-LOAM")
      throw std::invalid_argument("[ImageProjection]: the LeGO-LOAM ground removal is not part of the device path; "
                                  "pass the non-ground points and \"Patchwork\"");
    ip_.num_min_pts = numSubclusteringCriteria;
  }

  // :273-294 — any 16-byte x,y,z,* point type (pcl::PointXYZ; the KITTI record)
  template <typename PointT>
  void segmentCloud(const QUATRO_SHARED_PTR<pcl::PointCloud<PointT>>& pcPtr) {
    static_assert(sizeof(PointT) == 16, "point type must be a 16-byte x,y,z,pad record");
    const int P = static_cast<int>(pcPtr->points.size());
    const size_t NP = static_cast<size_t>(ip_.n_scan) * ip_.horizon_scan;
    valid_.assign(4 * NP, 0.f);
    outl_.assign(4 * NP, 0.f);
    labelmat_.assign(NP, -1);
    qtr_handle* h = quatro_hip::default_handle();
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    quatro_hip::check(h, qtr_segment_cloud(h, slot_lease.slot, reinterpret_cast<const float*>(pcPtr->points.data()), P, &ip_,
                                           valid_.data(), static_cast<int>(NP), &n_valid_, outl_.data(),
                                           static_cast<int>(NP), &n_outl_, &n_segments_, labelmat_.data(), QTR_MEM_HOST));
  }

  void getValidSegments(pcl::PointCloud<pcl::PointXYZ>& output) const { fill_xyz(output, valid_, n_valid_); }    // :244
  void getValidSegments(pcl::PointCloud<pcl::PointXYZI>& output) const { fill_xyzi(output, valid_, n_valid_); }  // :248
  void getOutliers(pcl::PointCloud<pcl::PointXYZ>& output) const { fill_xyz(output, outl_, n_outl_); }           // :260
  void getOutliers(pcl::PointCloud<pcl::PointXYZI>& output) const { fill_xyzi(output, outl_, n_outl_); }
  void getGround(pcl::PointCloud<pcl::PointXYZ>& output) const { output.clear(); }   // "Patchwork" mode: none (:252)
  void getGround(pcl::PointCloud<pcl::PointXYZI>& output) const { output.clear(); }
  const std::vector<int>& getLabelMat() const { return labelmat_; }
  int numSegments() const { return n_segments_; }
  const qtr_ip_params& params() const { return ip_; }

 private:
  static void fill_xyz(pcl::PointCloud<pcl::PointXYZ>& out, const std::vector<float>& src, int n) {
    out.clear();
    out.reserve(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) out.push_back(pcl::PointXYZ(src[4 * static_cast<size_t>(i)], src[4 * static_cast<size_t>(i) + 1],
                                                            src[4 * static_cast<size_t>(i) + 2]));
  }
  static void fill_xyzi(pcl::PointCloud<pcl::PointXYZI>& out, const std::vector<float>& src, int n) {
    out.clear();
    out.reserve(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
      pcl::PointXYZI p;
      p.x = src[4 * static_cast<size_t>(i)];
      p.y = src[4 * static_cast<size_t>(i) + 1];
      p.z = src[4 * static_cast<size_t>(i) + 2];
      p.intensity = src[4 * static_cast<size_t>(i) + 3];
      out.push_back(p);
    }
  }
  qtr_ip_params ip_;
  std::vector<float> valid_, outl_;
  std::vector<int> labelmat_;
  int n_valid_ = 0, n_outl_ = 0, n_segments_ = 0;
};

#endif  // IMAGE_PROJECTION_HPP

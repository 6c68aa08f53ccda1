// fpfh_manager.hpp — drop-in for url-kaist/Quatro's include/fpfh_manager.hpp (class FPFHManager, :25-238):
// normals + FPFH for both clouds, reciprocal matching, matched key-point clouds — all on the GPU through
// the C ABI (qtr_fpfh, qtr_match).  The matched-pair PCD cache (setSaveDir / setLoadDir, saveFeaturePair,
// loadFeaturePair, :91-96,179-232) goes through qtr_write_pcd_xyz / qtr_read_pcd_xyz: the same file name pattern
// and the same layout (source key points, then target key points, in one ASCII PCD).
#ifndef FPFH_MANAGER_H
#define FPFH_MANAGER_H

#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "quatro.hpp"

#ifndef QUATRO_POINT_TYPE_DEFINED
#define QUATRO_POINT_TYPE_DEFINED
typedef pcl::PointXYZ PointType;  // reference include/utility.h:50
#endif

class FPFHManager {
 public:
  double normal_radius_ = 0.5;
  double fpfh_radius_ = 0.6;
  bool is_initial_ = true;
  bool is_odometry_test_ = false;
  int interval_ = 1;
  unsigned long long seed_ = 0;              // tuple-test RNG seed (the reference seeds with time(NULL))
  std::vector<std::pair<int, int>> corr;     // correspondences (src index, tgt index)
  pcl::PointCloud<PointType> src_matched_pcl, tgt_matched_pcl;

  FPFHManager(double normal_radius, double fpfh_radius, int interval = 1)
      : normal_radius_(normal_radius), fpfh_radius_(fpfh_radius), interval_(interval) {}
  FPFHManager() {}

  void flushAllFeatures() { is_initial_ = true; }
  void setParams(float normal_radius, float fpfh_radius, int interval) {
    normal_radius_ = normal_radius;
    fpfh_radius_ = fpfh_radius;
    interval_ = interval;
  }
  void clearInputs() {
    is_initial_ = true;
    src_cloud_.clear();
    tgt_cloud_.clear();
    obj_desc_.clear();
    scene_desc_.clear();
  }
  void swapTgt2Src() {
    src_cloud_ = tgt_cloud_;
    obj_desc_ = scene_desc_;
  }

  // reference :98-153
  void setFeaturePair(QUATRO_SHARED_PTR<pcl::PointCloud<PointType>> src,
                      QUATRO_SHARED_PTR<pcl::PointCloud<PointType>> target) {
    if (normal_radius_ > fpfh_radius_)
      throw std::invalid_argument("[FPFHManager]: Normal should be lower than fpfh_radius!!!!");  // :99-102
    qtr_handle* h = quatro_hip::default_handle();
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    if (is_initial_ && !is_odometry_test_) {
      src_cloud_.assign(src->points.begin(), src->points.end());  // PCL's storage has its own allocator type
      compute(h, src_cloud_, obj_desc_);
      is_initial_ = false;
    } else {
      swapTgt2Src();
    }
    tgt_cloud_.assign(target->points.begin(), target->points.end());
    compute(h, tgt_cloud_, scene_desc_);
    qtr_frontend_params fp;
    qtr_default_frontend_params(&fp);
    fp.normal_radius = static_cast<float>(normal_radius_);
    fp.fpfh_radius = static_cast<float>(fpfh_radius_);
    fp.tuple_scale = 0.95f;  // calculateCorrespondences(..., true, true, true, 0.95), reference :126-127
    fp.seed = seed_;
    const int ns = static_cast<int>(src_cloud_.size()), nt = static_cast<int>(tgt_cloud_.size());
    std::vector<int> c2(2 * static_cast<size_t>(ns < nt ? ns : nt) + 2);
    int L = 0;
    quatro_hip::check(h, qtr_match(h, slot_lease.slot, quatro_hip::xyz4(src_cloud_), ns, obj_desc_.data(), quatro_hip::xyz4(tgt_cloud_),
                                   nt, scene_desc_.data(), &fp, c2.data(), static_cast<int>(c2.size() / 2), &L,
                                   QTR_MEM_HOST));
    corr.clear();
    src_matched_pcl.clear();
    tgt_matched_pcl.clear();
    for (int i = 0; i < L; ++i) {
      corr.emplace_back(c2[2 * i], c2[2 * i + 1]);
      const PointType& a = src_cloud_[static_cast<size_t>(c2[2 * i])];
      const PointType& b = tgt_cloud_[static_cast<size_t>(c2[2 * i + 1])];
      src_matched_pcl.push_back(PointType(a.x, a.y, a.z));
      tgt_matched_pcl.push_back(PointType(b.x, b.y, b.z));
    }
  }

  void setLoadDir(std::string loaddir) { loaddir_ = loaddir; }  // :91-93
  void setSaveDir(std::string savedir) { savedir_ = savedir; }  // :94-96

  // :179-200 — "Source is the first": one cloud, source half then target half
  void saveFeaturePair(int src_idx, int tgt_idx, bool verbose = false) {
    if (savedir_.empty()) throw std::invalid_argument("Save dir. is not set");
    const std::string pcdname = pair_name(savedir_, src_idx, tgt_idx);
    if (verbose) std::printf("[SAVER]: %s\n%zu + %zu\n", pcdname.c_str(), src_matched_pcl.size(), tgt_matched_pcl.size());
    std::vector<PointType> merge(src_matched_pcl.points.begin(), src_matched_pcl.points.end());
    merge.insert(merge.end(), tgt_matched_pcl.points.begin(), tgt_matched_pcl.points.end());
    if (qtr_write_pcd_xyz(pcdname.c_str(), reinterpret_cast<const float*>(merge.data()), static_cast<int>(merge.size()), 0) !=
        QTR_OK)
      throw std::runtime_error("[FPFHManager]: Save feature set failed.");
  }
  // :202-232
  void loadFeaturePair(int src_idx, int tgt_idx, bool verbose = false) {
    if (loaddir_.empty()) throw std::invalid_argument("Load dir. is not set");
    const std::string pcdname = pair_name(loaddir_, src_idx, tgt_idx);
    int n = 0;
    int rc = qtr_read_pcd_xyz(pcdname.c_str(), nullptr, 0, &n);  // size query
    std::vector<PointType> merge(static_cast<size_t>(n));
    if (rc == QTR_ERR_CAPACITY) rc = qtr_read_pcd_xyz(pcdname.c_str(), reinterpret_cast<float*>(merge.data()), n, &n);
    if (rc != QTR_OK) throw std::invalid_argument("[FPFHManager]: Load feature set failed.");
    src_matched_pcl.clear();
    tgt_matched_pcl.clear();
    for (size_t i = 0; i < merge.size(); ++i) (i < merge.size() / 2 ? src_matched_pcl : tgt_matched_pcl).push_back(merge[i]);
    if (verbose)
      std::printf("[LOADER]: Loaded data from %s...\n=>%zu %zu\n", pcdname.c_str(), src_matched_pcl.size(), tgt_matched_pcl.size());
  }

  pcl::PointCloud<PointType> getSrcKps() { return src_matched_pcl; }          // :172-174
  pcl::PointCloud<PointType> getTgtKps() { return tgt_matched_pcl; }          // :175-177
  std::vector<std::pair<int, int>> getCorrespondences() { return corr; }      // :234
  const std::vector<float>& getObjDescriptor() const { return obj_desc_; }    // n x 33, row-major
  const std::vector<float>& getSceneDescriptor() const { return scene_desc_; }

 private:
  static std::string pair_name(const std::string& dir, int src_idx, int tgt_idx) {
    char name[64];
    std::snprintf(name, sizeof(name), "/%06d_to_%06d.pcd", src_idx, tgt_idx);
    return dir + name;
  }
  std::string savedir_, loaddir_;
  void compute(qtr_handle* h, const std::vector<PointType>& cloud, std::vector<float>& desc) {
    desc.assign(33 * cloud.size(), 0.f);
    quatro_hip::SlotLease slot_lease;  // (the caller's lease, when it holds one: setFeaturePair)
    quatro_hip::check(h, qtr_fpfh(h, slot_lease.slot, quatro_hip::xyz4(cloud), static_cast<int>(cloud.size()),
                                  static_cast<float>(normal_radius_), static_cast<float>(fpfh_radius_), nullptr,
                                  desc.data(), QTR_MEM_HOST));
  }
  std::vector<PointType> src_cloud_, tgt_cloud_;
  std::vector<float> obj_desc_, scene_desc_;
};

#endif  // FPFH_MANAGER_H

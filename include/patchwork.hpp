// patchwork.hpp — drop-in for url-kaist/Quatro's include/patchwork.hpp (class PatchWork<PointT>, :36-233, the
// ground segmentation the demo runs first on raw scans, examples/run_global_registration.cpp:136-146).
// estimate_ground() is one call into libquatro_hip.so (qtr_patchwork: concentric-zone binning by radix sort, one
// wavefront per patch for height sort, seed selection, iterated plane fits and the uprightness / elevation / flatness
// rule — quatro_amd/csrc/patchwork.hip).
//
// Differences a caller can observe:
//   * no ROS: the reference's constructor takes a ros::NodeHandle* and reads "/patchwork/..." parameters (:47-139).
//     Here the default constructor carries config/patchwork_params.yaml (the values the demo runs with), a second one
//     takes a qtr_pw_params, and a template constructor reads the same parameter names from any object with
//     ros::NodeHandle's param()/getParam() members — so `new PatchWork<PointType>(&nh)` still compiles where ROS exists.
//   * no visualisation publishers (revert_pc / reject_pc / plane viz) and no console banner.
//   * plane fits follow the tests' CPU restatement: float32 moments in a fixed summation order and pcl::eigen33's closed
//     form for the smallest eigenvector in place of Eigen::JacobiSVD (:264-280) — DESIGN.md lists the divergence.
//   * cloudOut / cloudNonground come back in the reference's order (zone, ring, sector; ascending z inside a patch).
#ifndef PATCHWORK_H
#define PATCHWORK_H

#include <chrono>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "quatro.hpp"

template <typename PointT>
class PatchWork {
 public:
  PatchWork() { qtr_pw_default_params(&pw_); }
  explicit PatchWork(const qtr_pw_params& p) : pw_(p) { check_input_parameters_are_correct(); }

  // :47-139 — NodeHandle-like: param(name, var, default) and getParam(name, var)
  template <typename NodeHandleT>
  explicit PatchWork(NodeHandleT* nh) {
    qtr_pw_default_params(&pw_);
    bool global_thr = true;
    std::vector<int> sectors, rings;
    std::vector<double> min_ranges, elev, flat;
    nh->param("/patchwork/sensor_height", pw_.sensor_height, 1.723);
    nh->param("/patchwork/num_iter", pw_.num_iter, 3);
    nh->param("/patchwork/num_lpr", pw_.num_lpr, 20);
    nh->param("/patchwork/num_min_pts", pw_.num_min_pts, 10);
    nh->param("/patchwork/th_seeds", pw_.th_seeds, 0.4);
    nh->param("/patchwork/th_dist", pw_.th_dist, 0.3);
    nh->param("/patchwork/max_r", pw_.max_range, 80.0);
    nh->param("/patchwork/min_r", pw_.min_range, 2.7);
    nh->param("/patchwork/uprightness_thr", pw_.uprightness_thr, 0.5);
    nh->param("/patchwork/adaptive_seed_selection_margin", pw_.adaptive_seed_selection_margin, -1.1);
    nh->param("/patchwork/using_global_elevation", global_thr, true);
    nh->param("/patchwork/global_elevation_threshold", pw_.global_elevation_thr, 0.0);
    pw_.using_global_thr = global_thr ? 1 : 0;
    nh->getParam("/patchwork/czm/num_zones", pw_.num_zones);
    nh->getParam("/patchwork/czm/num_sectors_each_zone", sectors);
    nh->getParam("/patchwork/czm/num_rings_each_zone", rings);
    nh->getParam("/patchwork/czm/min_ranges_each_zone", min_ranges);
    nh->getParam("/patchwork/czm/elevation_thresholds", elev);
    nh->getParam("/patchwork/czm/flatness_thresholds", flat);
    const size_t nz = static_cast<size_t>(pw_.num_zones);
    if (pw_.num_zones < 1 || pw_.num_zones > 4 || sectors.size() != nz || rings.size() != nz || min_ranges.size() != nz)
      throw std::invalid_argument("Some parameters are wrong! the size of parameters should be same");  // :598-604
    if (elev.size() != flat.size() || elev.size() > 8)
      throw std::invalid_argument("Some parameters are wrong! Check the elevation/flatness_thresholds");  // :610
    for (size_t i = 0; i < nz; ++i) {
      pw_.num_sectors_each_zone[i] = sectors[i];
      pw_.num_rings_each_zone[i] = rings[i];
      pw_.min_ranges[i] = min_ranges[i];
    }
    pw_.num_thr = static_cast<int>(elev.size());
    for (size_t i = 0; i < elev.size(); ++i) {
      pw_.elevation_thr[i] = elev[i];
      pw_.flatness_thr[i] = flat[i];
    }
    check_input_parameters_are_correct();
  }

  // :329-476
  void estimate_ground(const pcl::PointCloud<PointT>& cloudIn, pcl::PointCloud<PointT>& cloudOut,
                       pcl::PointCloud<PointT>& cloudNonground, double& time_taken) {
    const auto t0 = std::chrono::steady_clock::now();
    const int P = static_cast<int>(cloudIn.points.size());
    const size_t cap = static_cast<size_t>(P > 0 ? P : 1);
    g_.resize(4 * cap);
    n_.resize(4 * cap);
    int ng = 0, nn = 0;
    qtr_handle* h = quatro_hip::default_handle();
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    const float* in = pack(cloudIn);
    quatro_hip::check(h, qtr_patchwork(h, slot_lease.slot, in, P, &pw_, g_.data(), static_cast<int>(cap), &ng, n_.data(),
                                       static_cast<int>(cap), &nn, QTR_MEM_HOST));
    unpack(g_, ng, cloudOut);
    unpack(n_, nn, cloudNonground);
    time_taken = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }

  const qtr_pw_params& params() const { return pw_; }

 private:
  void check_input_parameters_are_correct() const {  // :590-614
    if (pw_.num_zones < 1 || pw_.num_zones > 4 || pw_.num_thr < 0 || pw_.num_thr > 8)
      throw std::invalid_argument("Some parameters are wrong! the size of parameters should be same");
    if (pw_.min_range != pw_.min_ranges[0])
      throw std::invalid_argument("Setting min. ranges are weired! The first term should be eqaul to min_range_");
  }

  // 16-byte records (pcl::PointXYZ, the KITTI x,y,z,intensity record) go through untouched; 32-byte PCL records with the
  // extra field in the second half (pcl::PointXYZI) are packed to x,y,z,field and unpacked on the way back.
  const float* pack(const pcl::PointCloud<PointT>& c) {
    static_assert(sizeof(PointT) == 16 || sizeof(PointT) == 32, "point type must be a 16- or 32-byte PCL record");
    if (sizeof(PointT) == 16) return reinterpret_cast<const float*>(c.points.data());
    in_.resize(4 * c.points.size());
    for (size_t i = 0; i < c.points.size(); ++i) {
      float f[8];
      std::memcpy(f, &c.points[i], 32);
      in_[4 * i] = f[0], in_[4 * i + 1] = f[1], in_[4 * i + 2] = f[2], in_[4 * i + 3] = f[4];
    }
    return in_.data();
  }
  static void unpack(const std::vector<float>& src, int n, pcl::PointCloud<PointT>& out) {
    out.clear();
    out.points.resize(static_cast<size_t>(n));
    if (sizeof(PointT) == 16) {
      if (n > 0) std::memcpy(static_cast<void*>(out.points.data()), src.data(), static_cast<size_t>(n) * 16);
      return;
    }
    for (int i = 0; i < n; ++i) {
      float f[8] = {src[4 * static_cast<size_t>(i)], src[4 * static_cast<size_t>(i) + 1], src[4 * static_cast<size_t>(i) + 2], 1.f,
                    src[4 * static_cast<size_t>(i) + 3], 0.f, 0.f, 0.f};
      std::memcpy(static_cast<void*>(&out.points[static_cast<size_t>(i)]), f, 32);
    }
  }

  qtr_pw_params pw_;
  std::vector<float> g_, n_, in_;
};

#endif

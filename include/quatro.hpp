// quatro.hpp — drop-in for url-kaist/Quatro's include/quatro.hpp: the same class surface
// (setInputSource / setInputTarget / reset / computeTransformation / getters, struct Params,
// RegistrationSolution, the three enums, public members solution_, noise_bound_, cost_,
// using_pre_estimated_RyRx_, estimated_RyRx_, and the free voxelize<T>() overloads), with every body
// forwarding to the gfx950 kernels through the C ABI of quatro_hip.h.  Host code only; link with
// -lquatro_hip.  Line numbers below cite the reference header this file replaces.
//
// With PCL and Eigen installed (the reference's environment) the real pcl::Registration / Eigen types
// are used.  Without them (e.g. this repository's CI image) a minimal shim of exactly the types that
// appear on the API is provided so the header still compiles and the same caller code runs.
#ifndef QUATRO_H
#define QUATRO_H

#include <array>
#include <cmath>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "quatro_hip.h"
#include "quatro_hip_cxx.hpp"

#if defined(__has_include)
#if __has_include(<pcl/registration/registration.h>) && __has_include(<Eigen/Core>) && !defined(QUATRO_FORCE_SHIM)
#define QUATRO_HAVE_PCL 1
#endif
#endif

#ifdef QUATRO_HAVE_PCL
#include <Eigen/Core>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/registration.h>
#define QUATRO_SHARED_PTR boost::shared_ptr
#else
// ------------------------------------------------------------------ minimal Eigen / PCL stand-ins
namespace Eigen {
template <int R, int C>
struct FixedMatrixD {
  double m[R][C];
  FixedMatrixD() { std::memset(m, 0, sizeof(m)); }
  double& operator()(int r, int c) { return m[r][c]; }
  const double& operator()(int r, int c) const { return m[r][c]; }
  double& operator()(int i) { return (&m[0][0])[i]; }
  const double& operator()(int i) const { return (&m[0][0])[i]; }
  static FixedMatrixD Identity() {
    FixedMatrixD a;
    for (int i = 0; i < (R < C ? R : C); ++i) a.m[i][i] = 1.0;
    return a;
  }
  static FixedMatrixD Zero() { return FixedMatrixD(); }
};
using Matrix4d = FixedMatrixD<4, 4>;
using Matrix3d = FixedMatrixD<3, 3>;
using Vector3d = FixedMatrixD<3, 1>;
}  // namespace Eigen
namespace pcl {
struct PointXYZ {
  float x = 0, y = 0, z = 0, pad = 0;  // 16 bytes, as PCL's EIGEN_ALIGN16 PointXYZ
  PointXYZ() = default;
  PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
template <typename PointT>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); }
  void reserve(std::size_t n) { points.reserve(n); }
  void push_back(const PointT& p) { points.push_back(p); }
  PointT& operator[](std::size_t i) { return points[i]; }
  const PointT& operator[](std::size_t i) const { return points[i]; }
};
template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
 public:
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using Matrix4 = Eigen::Matrix4d;
  virtual ~Registration() = default;
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) { target_ = cloud; }
  const std::string& getClassName() const { return reg_name_; }

 protected:
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
  std::string reg_name_;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  int max_iterations_ = 10;
};
}  // namespace pcl
#define QUATRO_SHARED_PTR std::shared_ptr
#endif  // QUATRO_HAVE_PCL

namespace quatro_hip {
// pcl::PointXYZ and friends are 16-byte records starting with float x,y,z: passed through as xyz4.
template <typename PointT>
inline const float* xyz4(const std::vector<PointT>& pts) {
  static_assert(sizeof(PointT) == 16, "point type must be a 16-byte x,y,z,pad record (pcl::PointXYZ)");
  return reinterpret_cast<const float*>(pts.data());
}
}  // namespace quatro_hip

// voxelize<T>() — reference include/quatro.hpp:49-68 (pcl::VoxelGrid, leaf = voxelSize)
template <typename T>
void voxelize(const QUATRO_SHARED_PTR<pcl::PointCloud<T>> srcPtr, QUATRO_SHARED_PTR<pcl::PointCloud<T>> dstPtr,
              double voxelSize) {
  qtr_handle* h = quatro_hip::default_handle();
  const int P = static_cast<int>(srcPtr->points.size());
  dstPtr->points.assign(static_cast<size_t>(P), T());
  int n = 0;
  quatro_hip::check(h, qtr_voxelize(h, 0, quatro_hip::xyz4(srcPtr->points), P, static_cast<float>(voxelSize),
                                    reinterpret_cast<float*>(dstPtr->points.data()), P, &n, QTR_MEM_HOST));
  dstPtr->points.resize(static_cast<size_t>(n));
}
template <typename T>
void voxelize(pcl::PointCloud<T>& src, QUATRO_SHARED_PTR<pcl::PointCloud<T>> dstPtr, double voxelSize) {
  qtr_handle* h = quatro_hip::default_handle();
  const int P = static_cast<int>(src.points.size());
  dstPtr->points.assign(static_cast<size_t>(P), T());
  int n = 0;
  quatro_hip::check(h, qtr_voxelize(h, 0, quatro_hip::xyz4(src.points), P, static_cast<float>(voxelSize),
                                    reinterpret_cast<float*>(dstPtr->points.data()), P, &n, QTR_MEM_HOST));
  dstPtr->points.resize(static_cast<size_t>(n));
}

template <typename PointSource, typename PointTarget, typename Scalar = double>
class Quatro : public pcl::Registration<PointSource, PointTarget, Scalar> {
 public:
  using Base = pcl::Registration<PointSource, PointTarget, Scalar>;
  using PointCloudSource = typename Base::PointCloudSource;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = typename Base::PointCloudTarget;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using Matrix4 = typename Base::Matrix4;
  using Base::input_;
  using Base::max_iterations_;
  using Base::reg_name_;
  using Base::target_;

  Quatro() : noise_bound_(0.3) { reg_name_ = "Quatro"; }  // reference :111-126
  Quatro(const Quatro&) = delete;                        // reference :136-144
  Quatro(Quatro&&) = delete;
  Quatro& operator=(const Quatro&) = delete;
  Quatro& operator=(Quatro&&) = delete;
  ~Quatro() {}

  bool using_pre_estimated_RyRx_ = false;                         // :158
  Eigen::Matrix3d estimated_RyRx_ = Eigen::Matrix3d::Identity();  // :159

  struct RegistrationSolution {  // :161-168
    bool valid = true;
    double scale = 1.0;
    Eigen::Vector3d translation;
    Eigen::Matrix3d rotation;
  };
  RegistrationSolution solution_;  // :170

  enum class ROTATION_ESTIMATION_ALGORITHM { GNC_TLS = 0, FGR = 1 };                           // :172-175
  enum class INLIER_SELECTION_MODE { PMC_EXACT = 0, PMC_HEU = 1, KCORE_HEU = 2, NONE = 3 };    // :184-189
  enum class INLIER_GRAPH_FORMULATION { CHAIN = 0, COMPLETE = 1 };                             // :197-200

  struct Params {  // :202-268 — same fields, same defaults
    std::string reg_name = "Quatro";
    std::string cote_mode = "median";
    bool using_rot_inliers_when_estimating_cote = false;
    double noise_bound = 0.3;
    double cbar2 = 1;
    bool estimate_scaling = true;  // accepted; the reference forces scale = 1 (:361)
    ROTATION_ESTIMATION_ALGORITHM rotation_estimation_algorithm = ROTATION_ESTIMATION_ALGORITHM::GNC_TLS;
    double rotation_gnc_factor = 1.4;
    size_t rotation_max_iterations = 100;
    double rotation_cost_threshold = 1e-6;
    INLIER_GRAPH_FORMULATION rotation_tim_graph = INLIER_GRAPH_FORMULATION::CHAIN;
    INLIER_SELECTION_MODE inlier_selection_mode = INLIER_SELECTION_MODE::PMC_HEU;
    double kcore_heuristic_threshold = 0.5;
    bool use_max_clique = true;
    bool max_clique_exact_solution = true;
    double max_clique_time_limit = 3600;
  };
  double noise_bound_;  // :269 — used by COTE (:600-601)
  double cost_ = 0;     // :749

  Params getParams() { return params_; }               // :271
  void setParams(Params params) { params_ = params; }  // :273

  void setPreEstaimatedRyRx(Eigen::Matrix4d& estimated_RyRx) {  // :276-279 (sic)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) estimated_RyRx_(r, c) = estimated_RyRx(r, c);
    using_pre_estimated_RyRx_ = true;
  }

  void setInputSource(const PointCloudSourceConstPtr& cloud) override { Base::setInputSource(cloud); }  // :286-289
  void setInputTarget(const PointCloudTargetConstPtr& cloud) override {                                  // :296-305
    if (cloud->points.empty()) {
      std::cerr << "[pcl::" << reg_name_ << "::setInputSource] Invalid or empty point cloud dataset given!\n";
      return;
    }
    Base::setInputTarget(cloud);
  }

  inline void setMaximumIterations(int nr_iterations) { max_iterations_ = nr_iterations; }  // :751

  void reset(const Params& params) {  // :755-765
    reg_name_ = params.reg_name;
    params_ = params;
    max_clique_.clear();
    rotation_inliers_.clear();
    final_inliers_.clear();
  }

  void computeTransformation(PointCloudSource&, const Matrix4&) override {}  // :767 (empty in the reference too)

  // :769-936.  output is left untouched when the max clique has <= 1 member (solution_.valid = false).
  void computeTransformation(Eigen::Matrix4d& output) {
    if (!input_ || !target_) throw std::invalid_argument("[Quatro] input clouds are not set");
    if (input_->points.size() != target_->points.size())
      throw std::invalid_argument("[Quatro] source and target keypoint clouds must have equal length");
    if (reg_name_ != "Quatro")
      throw std::invalid_argument("[solveForRotation] The param is wrong! It should be 'TEASER' or 'Quatro'");  // :410
    if (params_.cote_mode != "median" && params_.cote_mode != "weighted_mean")
      throw std::invalid_argument("[COTE]: Wrong parameter comes!");  // :911
    qtr_handle* h = quatro_hip::default_handle();
    qtr_params p;
    qtr_default_params(&p);
    p.noise_bound = params_.noise_bound;
    p.cbar2 = params_.cbar2;
    p.rotation_gnc_factor = params_.rotation_gnc_factor;
    p.rotation_cost_threshold = params_.rotation_cost_threshold;
    p.kcore_heuristic_threshold = params_.kcore_heuristic_threshold;
    p.cote_noise_bound = noise_bound_;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) p.ryrx[3 * r + c] = estimated_RyRx_(r, c);
    p.rotation_max_iterations = static_cast<int>(params_.rotation_max_iterations);
    p.inlier_selection_mode = static_cast<int>(params_.inlier_selection_mode);
    p.cote_median = params_.cote_mode == "median" ? 1 : 0;
    p.using_rot_inliers_when_estimating_cote = params_.using_rot_inliers_when_estimating_cote ? 1 : 0;
    p.using_pre_estimated_ryrx = using_pre_estimated_RyRx_ ? 1 : 0;
    const int L = static_cast<int>(input_->points.size());
    std::vector<int> clique(static_cast<size_t>(L > 0 ? L : 1)), rot(clique.size()), fin(clique.size());
    qtr_result res;
    const int rc = qtr_solve(h, 0, quatro_hip::xyz4(input_->points), quatro_hip::xyz4(target_->points), L, &p, &res,
                             clique.data(), rot.data(), fin.data(), static_cast<int>(clique.size()), QTR_MEM_HOST);
    quatro_hip::check(h, rc);
    params_.noise_bound *= 2.0;  // the reference persists noise_bound *= 2/scale (:850-852); reset() restores it
    max_clique_.assign(clique.begin(), clique.begin() + res.n_clique);
    num_maxclique_ = res.n_clique;
    if (!res.valid) {  // :809-813
      solution_.valid = false;
      return;
    }
    rotation_inliers_.assign(rot.begin(), rot.begin() + res.n_rot_inliers);
    num_rot_inliers_ = res.n_rot_inliers;
    final_inliers_.assign(fin.begin(), fin.begin() + res.n_final);
    cost_ = res.cost;
    solution_.valid = true;
    solution_.scale = 1.0;
    output = Eigen::Matrix4d::Identity();
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) {
        solution_.rotation(r, c) = res.T[4 * r + c];
        output(r, c) = res.T[4 * r + c];
      }
      solution_.translation(r, 0) = res.T[4 * r + 3];
      output(r, 3) = res.T[4 * r + 3];
    }
  }

  void getMaxCliques(pcl::PointCloud<PointSource>& source_max_clique,
                     pcl::PointCloud<PointSource>& target_max_clique) {  // :949-953
    gather(*input_, source_max_clique, max_clique_);
    gather(*target_, target_max_clique, max_clique_);
  }
  void getFinalInliers(pcl::PointCloud<PointSource>& source_inliers, pcl::PointCloud<PointSource>& target_inliers) {
    gather(*input_, source_inliers, final_inliers_);  // :955-960
    gather(*target_, target_inliers, final_inliers_);
  }
  std::vector<int> getFinalInliersIndices() { return final_inliers_; }  // :962-964
  int getNumRotaionInliers() { return num_rot_inliers_; }               // :966-968 (sic)
  int getNumMaxCliqueInliers() { return num_maxclique_; }               // :970-972

 protected:
  template <typename CloudIn, typename CloudOut>
  static void gather(const CloudIn& raw, CloudOut& out, const std::vector<int>& idx) {
    out.clear();
    out.reserve(idx.size());
    for (int i : idx) out.push_back(raw.points[static_cast<size_t>(i)]);
  }
  Params params_;
  int num_rot_inliers_ = 0;
  int num_maxclique_ = 0;
  std::vector<int> max_clique_;
  std::vector<int> rotation_inliers_;
  std::vector<int> final_inliers_;
};

#endif  // QUATRO_H

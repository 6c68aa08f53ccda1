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
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
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
// Just enough of Eigen::Matrix for the types on the reference's API: fixed or dynamic sizes, (r, c) / (i) access,
// rows() / cols() / resize(), Identity / Zero / Ones.  Row-major storage; the wrappers below only use accessors,
// so they compile unchanged against the real (column-major) Eigen.
constexpr int Dynamic = -1;
template <typename T, int R, int C>
class Matrix {
  using Store = typename std::conditional<std::is_same<T, bool>::value, unsigned char, T>::type;

 public:
  Matrix() : rows_(R == Dynamic ? 0 : R), cols_(C == Dynamic ? 0 : C), d_(static_cast<size_t>(rows_) * cols_, Store()) {}
  Matrix(int r, int c) : rows_(r), cols_(c), d_(static_cast<size_t>(r) * c, Store()) {}
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  void resize(int r, int c) {
    rows_ = r;
    cols_ = c;
    d_.assign(static_cast<size_t>(r) * c, Store());
  }
  Store& operator()(int r, int c) { return d_[static_cast<size_t>(r) * cols_ + c]; }
  const Store& operator()(int r, int c) const { return d_[static_cast<size_t>(r) * cols_ + c]; }
  Store& operator()(int i) { return d_[static_cast<size_t>(i)]; }
  const Store& operator()(int i) const { return d_[static_cast<size_t>(i)]; }
  static Matrix Zero() { return Matrix(); }
  static Matrix Zero(int r, int c) { return Matrix(r, c); }
  static Matrix Ones(int r, int c) {
    Matrix a(r, c);
    for (auto& x : a.d_) x = Store(1);
    return a;
  }
  static Matrix Identity() {
    Matrix a;
    for (int i = 0; i < (a.rows_ < a.cols_ ? a.rows_ : a.cols_); ++i) a(i, i) = Store(1);
    return a;
  }

 private:
  int rows_, cols_;
  std::vector<Store> d_;
};
using Matrix4d = Matrix<double, 4, 4>;
using Matrix3d = Matrix<double, 3, 3>;
using Matrix2d = Matrix<double, 2, 2>;
using Vector3d = Matrix<double, 3, 1>;
using RowVectorXd = Matrix<double, 1, Dynamic>;
}  // namespace Eigen
namespace pcl {
struct PointXYZ {
  float x = 0, y = 0, z = 0, pad = 0;  // 16 bytes, as PCL's EIGEN_ALIGN16 PointXYZ
  PointXYZ() = default;
  PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct PointXYZI {
  float x = 0, y = 0, z = 0, pad = 0;
  float intensity = 0, pad2[3] = {0, 0, 0};  // 32 bytes, as PCL's EIGEN_ALIGN16 PointXYZI
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
#include "conversion.hpp"  // the reference header includes it too (:42): pcl2teaser / pcl2eigen / eigen2pcl / xyzi2xyz

#endif  // QUATRO_HAVE_PCL

namespace quatro_hip {
// pcl::PointXYZ and friends are 16-byte records starting with float x,y,z: passed through as xyz4.
template <typename PointT, typename Alloc>
inline const float* xyz4(const std::vector<PointT, Alloc>& pts) {  // any allocator: PCL's storage uses Eigen::aligned_allocator
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
  decltype(dstPtr->points) out(static_cast<size_t>(P), T());  // a temporary, like pcl::Filter::filter: dstPtr may alias srcPtr
  int n = 0;
  {
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    quatro_hip::check(h, qtr_voxelize(h, slot_lease.slot, quatro_hip::xyz4(srcPtr->points), P, static_cast<float>(voxelSize),
                                      reinterpret_cast<float*>(out.data()), P, &n, QTR_MEM_HOST));
  }
  out.resize(static_cast<size_t>(n));
  dstPtr->points.swap(out);
}
template <typename T>
void voxelize(pcl::PointCloud<T>& src, QUATRO_SHARED_PTR<pcl::PointCloud<T>> dstPtr, double voxelSize) {
  qtr_handle* h = quatro_hip::default_handle();
  const int P = static_cast<int>(src.points.size());
  decltype(dstPtr->points) out(static_cast<size_t>(P), T());  // dstPtr may point at src
  int n = 0;
  {
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    quatro_hip::check(h, qtr_voxelize(h, slot_lease.slot, quatro_hip::xyz4(src.points), P, static_cast<float>(voxelSize),
                                      reinterpret_cast<float*>(out.data()), P, &n, QTR_MEM_HOST));
  }
  out.resize(static_cast<size_t>(n));
  dstPtr->points.swap(out);
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
    if (reg_name_ != "Quatro" && reg_name_ != "TEASER")
      throw std::invalid_argument("[solveForRotation] The param is wrong! It should be 'TEASER' or 'Quatro'");  // :410
    if (reg_name_ == "TEASER" && using_pre_estimated_RyRx_) throw std::invalid_argument("Wrong reg type name is coming!");  // :424-426
    if (params_.cote_mode != "median" && params_.cote_mode != "weighted_mean")
      throw std::invalid_argument("[COTE]: Wrong parameter comes!");  // :911
    qtr_handle* h = quatro_hip::default_handle();
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
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
    p.reg_mode = reg_name_ == "TEASER" ? QTR_REG_TEASER : QTR_REG_QUATRO;
    qtr_result res;
    p.max_clique_time_limit = params_.max_clique_time_limit;  // :800 (PMC_EXACT only); travels with THIS call
    {
      const int rc = qtr_solve(h, slot_lease.slot, quatro_hip::xyz4(input_->points), quatro_hip::xyz4(target_->points), L, &p, &res,
                               clique.data(), rot.data(), fin.data(), static_cast<int>(clique.size()), QTR_MEM_HOST);
      quatro_hip::check(h, rc);
    }
    max_clique_.assign(clique.begin(), clique.begin() + res.n_clique);
    if (!res.valid) {  // :809-813: returns before num_maxclique_ (:822) and the noise-bound update (:850-852)
      solution_.valid = false;
      return;
    }
    num_maxclique_ = res.n_clique;
    params_.noise_bound *= 2.0;  // the reference persists noise_bound *= 2/scale (:850-852); reset() restores it
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

  // ---- the individually callable stages of the reference class, served by the device (stages.hip).  The fused
  // computeTransformation above never materialises TIMs and does not call them.
  // computeTIMs :307-344
  Eigen::Matrix<double, 3, Eigen::Dynamic> computeTIMs(const Eigen::Matrix<double, 3, Eigen::Dynamic>& v,
                                                       Eigen::Matrix<int, 2, Eigen::Dynamic>* map) {
    const int N = static_cast<int>(v.cols());
    const long long K = static_cast<long long>(N) * (N - 1) / 2;
    Eigen::Matrix<double, 3, Eigen::Dynamic> vtilde(3, static_cast<int>(K));
    if (map) map->resize(2, static_cast<int>(K));
    if (K <= 0) return vtilde;
    std::vector<double> in(static_cast<size_t>(3) * N), out(static_cast<size_t>(3) * K);
    std::vector<int> mp(static_cast<size_t>(2) * K);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < N; ++c) in[static_cast<size_t>(r) * N + c] = v(r, c);
    qtr_handle* h = quatro_hip::default_handle();
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    quatro_hip::check(h, qtr_compute_tims(h, slot_lease.slot, in.data(), N, out.data(), mp.data()));
    for (int r = 0; r < 3; ++r)
      for (long long c = 0; c < K; ++c) vtilde(r, static_cast<int>(c)) = out[static_cast<size_t>(r) * K + c];
    if (map)
      for (int r = 0; r < 2; ++r)
        for (long long c = 0; c < K; ++c) (*map)(r, static_cast<int>(c)) = mp[static_cast<size_t>(r) * K + c];
    return vtilde;
  }

  // solveForScale :346-386 (the reference forces scale = 1)
  double solveForScale(const Eigen::Matrix<double, 3, Eigen::Dynamic>& v1,
                       const Eigen::Matrix<double, 3, Eigen::Dynamic>& v2) {
    scale_inliers_mask_.resize(1, static_cast<int>(v1.cols()));
    solveForScale(v1, v2, &solution_.scale, &scale_inliers_mask_);
    return solution_.scale;
  }
  void solveForScale(const Eigen::Matrix<double, 3, Eigen::Dynamic>& src,
                     const Eigen::Matrix<double, 3, Eigen::Dynamic>& dst, double* scale,
                     Eigen::Matrix<bool, 1, Eigen::Dynamic>* inliers) {
    if (src.cols() != dst.cols()) throw std::invalid_argument("[solveForScale] dimension mismatch");
    if (scale) *scale = 1;
    if (!inliers) return;
    const long long K = src.cols();
    inliers->resize(1, static_cast<int>(K));
    if (K == 0) return;
    std::vector<double> a(static_cast<size_t>(3) * K), b(a.size());
    for (int r = 0; r < 3; ++r)
      for (long long c = 0; c < K; ++c) {
        a[static_cast<size_t>(r) * K + c] = src(r, static_cast<int>(c));
        b[static_cast<size_t>(r) * K + c] = dst(r, static_cast<int>(c));
      }
    std::vector<unsigned char> mask(static_cast<size_t>(K));
    qtr_handle* h = quatro_hip::default_handle();
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    quatro_hip::check(h, qtr_scale_mask(h, slot_lease.slot, a.data(), b.data(), K, params_.noise_bound, params_.cbar2, mask.data()));
    for (long long c = 0; c < K; ++c) (*inliers)(0, static_cast<int>(c)) = mask[static_cast<size_t>(c)] != 0;
  }

  // solveForRotation :388-428, solveForRotation2D :430-572
  Eigen::Matrix3d solveForRotation(const Eigen::Matrix<double, 3, Eigen::Dynamic>& v1,
                                   const Eigen::Matrix<double, 3, Eigen::Dynamic>& v2) {
    rotation_inliers_mask_.resize(1, static_cast<int>(v1.cols()));
    if (reg_name_ == "TEASER") {  // the branch the reference names in its message but never implements (:409-411)
      if (using_pre_estimated_RyRx_) throw std::invalid_argument("Wrong reg type name is coming!");  // :424-426
      if (v1.cols() != v2.cols() || !(params_.rotation_gnc_factor > 1))
        throw std::invalid_argument("[solveForRotation] bad arguments");
      const int M = static_cast<int>(v1.cols());
      if (M == 0) return solution_.rotation;
      std::vector<double> a(static_cast<size_t>(3) * M), b(a.size());
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < M; ++c) {
          a[static_cast<size_t>(r) * M + c] = v1(r, c);
          b[static_cast<size_t>(r) * M + c] = v2(r, c);
        }
      double R9[9], cost = 0;
      int iters = 0;
      std::vector<unsigned char> inl(static_cast<size_t>(M));
      qtr_handle* h = quatro_hip::default_handle();
      quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
      quatro_hip::check(h, qtr_gnc_rotation3d(h, slot_lease.slot, a.data(), b.data(), M, params_.noise_bound, params_.rotation_gnc_factor,
                                              static_cast<int>(params_.rotation_max_iterations),
                                              params_.rotation_cost_threshold, R9, &cost, &iters, inl.data()));
      cost_ = cost;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) solution_.rotation(r, c) = R9[3 * r + c];
      for (int c = 0; c < M; ++c) rotation_inliers_mask_(0, c) = inl[static_cast<size_t>(c)] != 0;
      return solution_.rotation;
    }
    if (reg_name_ != "Quatro")
      throw std::invalid_argument("[solveForRotation] The param is wrong! It should be 'TEASER' or 'Quatro'");
    Eigen::Matrix<double, 2, Eigen::Dynamic> src_2d(2, static_cast<int>(v1.cols())), dst_2d(2, static_cast<int>(v2.cols()));
    for (int r = 0; r < 2; ++r) {
      for (int c = 0; c < v1.cols(); ++c) src_2d(r, c) = v1(r, c);
      for (int c = 0; c < v2.cols(); ++c) dst_2d(r, c) = v2(r, c);
    }
    Eigen::Matrix2d rotation_2d = Eigen::Matrix2d::Identity();
    solveForRotation2D(src_2d, dst_2d, &rotation_2d, &rotation_inliers_mask_);
    Eigen::Matrix3d rot_yaw = Eigen::Matrix3d::Identity();
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 2; ++c) rot_yaw(r, c) = rotation_2d(r, c);
    solution_.rotation = rot_yaw;
    if (using_pre_estimated_RyRx_) {  // :419-423: Rz * RyRx
      Eigen::Matrix3d prod;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          prod(r, c) = (rot_yaw(r, 0) * estimated_RyRx_(0, c) + rot_yaw(r, 1) * estimated_RyRx_(1, c)) +
                       rot_yaw(r, 2) * estimated_RyRx_(2, c);
      solution_.rotation = prod;
    }
    return solution_.rotation;
  }
  void solveForRotation2D(const Eigen::Matrix<double, 2, Eigen::Dynamic>& src,
                          const Eigen::Matrix<double, 2, Eigen::Dynamic>& dst, Eigen::Matrix2d* rotation,
                          Eigen::Matrix<bool, 1, Eigen::Dynamic>* inliers) {
    if (!rotation || src.cols() != dst.cols() || !(params_.rotation_gnc_factor > 1))
      throw std::invalid_argument("[solveForRotation2D] bad arguments");
    const int M = static_cast<int>(src.cols());
    if (inliers) inliers->resize(1, M);
    if (M == 0) return;
    std::vector<double> a(static_cast<size_t>(2) * M), b(a.size());
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < M; ++c) {
        a[static_cast<size_t>(r) * M + c] = src(r, c);
        b[static_cast<size_t>(r) * M + c] = dst(r, c);
      }
    double R4[4], cost = 0;
    int iters = 0;
    std::vector<unsigned char> inl(static_cast<size_t>(M));
    qtr_handle* h = quatro_hip::default_handle();
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    quatro_hip::check(h, qtr_gnc_rotation2d(h, slot_lease.slot, a.data(), b.data(), M, params_.noise_bound, params_.rotation_gnc_factor,
                                            static_cast<int>(params_.rotation_max_iterations),
                                            params_.rotation_cost_threshold, R4, &cost, &iters, inl.data()));
    cost_ = cost;
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 2; ++c) (*rotation)(r, c) = R4[2 * r + c];
    if (inliers)
      for (int c = 0; c < M; ++c) (*inliers)(0, c) = inl[static_cast<size_t>(c)] != 0;
  }

  // solveForTranslation :574-615, estimate :618-747
  Eigen::Vector3d solveForTranslation(const Eigen::Matrix<double, 3, Eigen::Dynamic>& v1,
                                      const Eigen::Matrix<double, 3, Eigen::Dynamic>& v2,
                                      bool using_median_selection = false) {
    translation_inliers_mask_.resize(1, static_cast<int>(v1.cols()));
    solveForTranslation(v1, v2, &solution_.translation, &translation_inliers_mask_, using_median_selection);
    return solution_.translation;
  }
  void solveForTranslation(const Eigen::Matrix<double, 3, Eigen::Dynamic>& src,
                           const Eigen::Matrix<double, 3, Eigen::Dynamic>& dst, Eigen::Vector3d* translation,
                           Eigen::Matrix<bool, 1, Eigen::Dynamic>* inliers, bool using_median_selection) {
    if (src.cols() != dst.cols() || !translation) throw std::invalid_argument("[solveForTranslation] bad arguments");
    const int N = static_cast<int>(src.cols());
    const double beta = noise_bound_ * std::sqrt(params_.cbar2);
    Eigen::RowVectorXd row(1, N), alphas(1, N);
    Eigen::Matrix<bool, 1, Eigen::Dynamic> all = Eigen::Matrix<bool, 1, Eigen::Dynamic>::Ones(1, N), tmp(1, N);
    for (int c = 0; c < N; ++c) alphas(0, c) = beta;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < N; ++c) row(0, c) = dst(r, c) - src(r, c);
      double e = 0;
      estimate(row, alphas, &e, &tmp, using_median_selection);
      (*translation)(r, 0) = e;
      for (int c = 0; c < N; ++c) all(0, c) = all(0, c) && tmp(0, c);
    }
    if (inliers) *inliers = all;
  }
  void estimate(const Eigen::RowVectorXd& X, const Eigen::RowVectorXd& ranges, double* estimate_out,
                Eigen::Matrix<bool, 1, Eigen::Dynamic>* inliers, bool using_median_selection = false) {
    const int N = static_cast<int>(X.cols());
    if (ranges.cols() != N || N < 2) throw std::invalid_argument("[estimate] dimension mismatch or a single element");
    bool uniform = true;
    for (int c = 1; c < N; ++c) uniform = uniform && (ranges(0, c) == ranges(0, 0));
    std::vector<double> x(static_cast<size_t>(N)), r(static_cast<size_t>(N));
    for (int c = 0; c < N; ++c) {
      x[static_cast<size_t>(c)] = X(0, c);
      r[static_cast<size_t>(c)] = ranges(0, c);
    }
    std::vector<unsigned char> inl(static_cast<size_t>(N));
    double e = 0;
    int ncard = 0;
    qtr_handle* h = quatro_hip::default_handle();
    quatro_hip::SlotLease slot_lease;  // a free stream slot of the process-wide handle
    if (uniform)
      quatro_hip::check(h, qtr_cote_estimate(h, slot_lease.slot, x.data(), N, ranges(0, 0), using_median_selection ? 1 : 0, &e,
                                             inl.data(), &ncard));
    else
      quatro_hip::check(h, qtr_cote_estimate_ranges(h, slot_lease.slot, x.data(), r.data(), N, using_median_selection ? 1 : 0, &e,
                                                    inl.data(), &ncard));
    if (estimate_out) *estimate_out = e;
    if (inliers) {
      inliers->resize(1, N);
      for (int c = 0; c < N; ++c) (*inliers)(0, c) = inl[static_cast<size_t>(c)] != 0;
    }
  }

  // :938-947
  void setInliers(const Eigen::Matrix<double, 3, Eigen::Dynamic>& raw, pcl::PointCloud<PointSource>& inliers,
                  const std::vector<int>& idx_inliers) {
    inliers.clear();
    inliers.reserve(idx_inliers.size());
    for (const int idx : idx_inliers)
      inliers.push_back(PointSource(static_cast<float>(raw(0, idx)), static_cast<float>(raw(1, idx)),
                                    static_cast<float>(raw(2, idx))));
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
  Eigen::Matrix<bool, 1, Eigen::Dynamic> scale_inliers_mask_, rotation_inliers_mask_, translation_inliers_mask_;  // :1009-1015
};

#include "conversion.hpp"  // the reference header includes it too (:42): pcl2teaser / pcl2eigen / eigen2pcl / xyzi2xyz

#endif  // QUATRO_H

// C entry points around member functions of the REFERENCE's class Quatro (include/quatro.hpp of /root/reference) and its
// include/teaser/utils.h: computeTIMs, solveForScale, solveForRotation2D (+ teaser::utils::svdRot2d),
// solveForTranslation, estimate.  oracle/Makefile (target ref_solver) cuts those definitions out of the header where it
// lies (oracle/ref_extract.py -> a temporary under _ref/), includes them HERE inside a host struct that provides the
// members they touch, and compiles against oracle/ref_shim_solver/ (an Eigen-subset stand-in: eager evaluation,
// index-order reductions, its own small Jacobi SVD).  Test infrastructure: tests/ compare oracle/quatro_oracle.cpp's
// restatement of the back end with this build, and tests/golden/make_golden.py stores its outputs as solver_ref.npz.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <iterator>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/SVD>
#include <pcl/point_cloud.h>
#include "teaser/graph.h"   // the reference's own (teaser::Graph; MaxCliqueSolver is only DECLARED there, see below)
#include "teaser/macros.h"
#include "teaser/utils.h"

using namespace std;  // the reference header does the same (include/quatro.hpp:44)
#include QREF_FREE_INC   // pcl2eigen, cut out of include/conversion.hpp

// teaser::MaxCliqueSolver::findMaxClique lives in src/graph.cc and calls PMC, which is absent.  The stand-in hands the
// graph the REFERENCE code built to a callback (the tests plug the oracle's clique search in) — so everything around
// the clique search in computeTransformation is the reference's text, and the graph it searches is the reference's.
typedef int (*qref_clique_fn)(const uint64_t* bitmap, int L, int W, int mode, double kcore_thr, int* out);
static qref_clique_fn g_clique = nullptr;
extern "C" void qref_set_clique_callback(qref_clique_fn f) { g_clique = f; }
std::vector<int> teaser::MaxCliqueSolver::findMaxClique(teaser::Graph graph) {
  const int L = graph.numVertices(), W = (L + 63) / 64;
  std::vector<uint64_t> bm((size_t)L * (W > 0 ? W : 1), 0);
  for (int i = 0; i < L; ++i)
    for (int j : graph.getEdges(i)) bm[(size_t)i * W + (j >> 6)] |= 1ULL << (j & 63);
  std::vector<int> out((size_t)(L > 0 ? L : 1));
  const int n = g_clique ? g_clique(bm.data(), L, W, (int)params_.solver_mode, params_.kcore_heuristic_threshold, out.data()) : 0;
  out.resize((size_t)n);
  return out;
}

namespace {
struct RefQuatro {
  typedef pcl::PointXYZ PointType;
#include QREF_MEMBERS_INC
  // the members those definitions touch (declared at include/quatro.hpp:158-159, 269, 749, 1003-1036)
  std::string reg_name_ = "Quatro";
  bool using_pre_estimated_RyRx_ = false;
  Eigen::Matrix3d estimated_RyRx_ = Eigen::Matrix3d::Identity();
  RegistrationSolution solution_;
  double noise_bound_ = 0.3;
  double cost_ = 0;
  pcl::PointCloud<pcl::PointXYZ>::ConstPtr input_, target_;
  Eigen::Matrix<double, 3, Eigen::Dynamic> src_matched, tgt_matched;
  Params params_;
  int num_rot_inliers_ = 0, num_maxclique_ = 0;
  teaser::Graph inlier_graph_;
  Eigen::Matrix<bool, 1, Eigen::Dynamic> rotation_inliers_mask_, translation_inliers_mask_, scale_inliers_mask_;
  Eigen::Matrix<double, 3, Eigen::Dynamic> src_tims_, dst_tims_, pruned_src_tims_, pruned_dst_tims_;
  Eigen::Matrix<int, 2, Eigen::Dynamic> src_tims_map_, dst_tims_map_, src_tims_map_rotation_, dst_tims_map_rotation_;
  std::vector<int> max_clique_, rotation_inliers_, translation_inliers_, final_inliers_;
};

struct Quiet {  // the functions narrate on std::cout
  std::streambuf* old;
  std::ostringstream sink;
  Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~Quiet() { std::cout.rdbuf(old); }
};
typedef Eigen::Matrix<double, 3, Eigen::Dynamic> M3;
typedef Eigen::Matrix<double, 2, Eigen::Dynamic> M2;
}  // namespace

// v: 3 x N row-major.  out: 3 x K row-major, map: 2 x K row-major, K = N (N - 1) / 2
extern "C" void qref_compute_tims(const double* v, int N, double* out, int* map) {
  Quiet q;
  RefQuatro r;
  M3 V(3, N);
  for (int a = 0; a < 3; ++a) for (int j = 0; j < N; ++j) V(a, j) = v[(size_t)a * N + j];
  Eigen::Matrix<int, 2, Eigen::Dynamic> mp;
  const M3 t = r.computeTIMs(V, &mp);
  const long long K = t.cols();
  for (int a = 0; a < 3; ++a) for (long long j = 0; j < K; ++j) out[(size_t)a * K + j] = t(a, j);
  for (int a = 0; a < 2; ++a) for (long long j = 0; j < K; ++j) map[(size_t)a * K + j] = mp(a, j);
}
extern "C" void qref_scale_mask(const double* src, const double* dst, long long K, double noise_bound, double cbar2,
                                unsigned char* mask) {
  Quiet q;
  RefQuatro r;
  r.params_.noise_bound = noise_bound;
  r.params_.cbar2 = cbar2;
  M3 A(3, K), B(3, K);
  for (int a = 0; a < 3; ++a) for (long long j = 0; j < K; ++j) { A(a, j) = src[(size_t)a * K + j]; B(a, j) = dst[(size_t)a * K + j]; }
  double scale = 0;
  Eigen::Matrix<bool, 1, Eigen::Dynamic> inl(1, K);
  r.solveForScale(A, B, &scale, &inl);
  for (long long j = 0; j < K; ++j) mask[j] = inl(0, j) ? 1 : 0;
}
// NOTE: solveForRotation2D keeps its noise bound in a function-local static initialised on the FIRST call of the process
// (include/quatro.hpp:466-467); callers that need another bound need another process.
extern "C" void qref_gnc_rotation2d(const double* src, const double* dst, int N, double noise_bound, double gnc_factor,
                                    int max_iterations, double cost_threshold, double* R4, unsigned char* inliers,
                                    double* cost) {
  Quiet q;
  RefQuatro r;
  r.params_.noise_bound = noise_bound;
  r.params_.rotation_gnc_factor = gnc_factor;
  r.params_.rotation_max_iterations = (size_t)max_iterations;
  r.params_.rotation_cost_threshold = cost_threshold;
  M2 A(2, N), B(2, N);
  for (int a = 0; a < 2; ++a) for (int j = 0; j < N; ++j) { A(a, j) = src[(size_t)a * N + j]; B(a, j) = dst[(size_t)a * N + j]; }
  Eigen::Matrix2d R;
  Eigen::Matrix<bool, 1, Eigen::Dynamic> inl(1, N);
  r.solveForRotation2D(A, B, &R, &inl);
  R4[0] = R(0, 0); R4[1] = R(0, 1); R4[2] = R(1, 0); R4[3] = R(1, 1);
  for (int j = 0; j < N; ++j) inliers[j] = inl(0, j) ? 1 : 0;
  *cost = r.cost_;
}
extern "C" void qref_cote_estimate(const double* X, const double* ranges, int N, int median, double* est,
                                   unsigned char* inliers) {
  Quiet q;
  RefQuatro r;
  Eigen::RowVectorXd x(1, N), rg(1, N);
  for (int j = 0; j < N; ++j) { x(j) = X[j]; rg(j) = ranges[j]; }
  Eigen::Matrix<bool, 1, Eigen::Dynamic> inl(1, N);
  r.estimate(x, rg, est, &inl, median != 0);
  for (int j = 0; j < N; ++j) inliers[j] = inl(0, j) ? 1 : 0;
}
// src, dst: 3 x N row-major (src already rotated); t: 3
extern "C" void qref_translation(const double* src, const double* dst, int N, double cote_noise_bound, double cbar2,
                                 int median, double* t, unsigned char* inliers) {
  Quiet q;
  RefQuatro r;
  r.noise_bound_ = cote_noise_bound;
  r.params_.cbar2 = cbar2;
  M3 A(3, N), B(3, N);
  for (int a = 0; a < 3; ++a) for (int j = 0; j < N; ++j) { A(a, j) = src[(size_t)a * N + j]; B(a, j) = dst[(size_t)a * N + j]; }
  Eigen::Vector3d tr;
  Eigen::Matrix<bool, 1, Eigen::Dynamic> inl(1, N);
  r.solveForTranslation(A, B, &tr, &inl, median != 0);
  for (int a = 0; a < 3; ++a) t[a] = tr(a);
  for (int j = 0; j < N; ++j) inliers[j] = inl(0, j) ? 1 : 0;
}
// teaser::utils::svdRot2d / svdRot alone (include/teaser/utils.h:123-166)
extern "C" void qref_svd_rot2d(const double* X, const double* Y, const double* W, int N, double* R4) {
  M2 A(2, N), B(2, N);
  Eigen::Matrix<double, 1, Eigen::Dynamic> w(1, N);
  for (int j = 0; j < N; ++j) { A(0, j) = X[j]; A(1, j) = X[N + j]; B(0, j) = Y[j]; B(1, j) = Y[N + j]; w(j) = W[j]; }
  const Eigen::Matrix2d R = teaser::utils::svdRot2d(A, B, w);
  R4[0] = R(0, 0); R4[1] = R(0, 1); R4[2] = R(1, 0); R4[3] = R(1, 1);
}
extern "C" void qref_svd_rot3d(const double* X, const double* Y, const double* W, int N, double* R9) {
  M3 A(3, N), B(3, N);
  Eigen::Matrix<double, 1, Eigen::Dynamic> w(1, N);
  for (int a = 0; a < 3; ++a) for (int j = 0; j < N; ++j) { A(a, j) = X[(size_t)a * N + j]; B(a, j) = Y[(size_t)a * N + j]; }
  for (int j = 0; j < N; ++j) w(j) = W[j];
  const Eigen::Matrix3d R = teaser::utils::svdRot(A, B, w, 0);
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) R9[3 * a + b] = R(a, b);
}

// The whole back end, Quatro::computeTransformation (:769-936): src / tgt are L x 3 row-major matched keypoints.
// inlier_selection_mode 1 = PMC_HEU, 2 = KCORE_HEU (what the clique callback is asked for).  Outputs: T (4 x 4 row-major),
// the sorted clique, rotation inliers (indices into the clique chain), final inliers (indices into the inputs).
// Returns solution_.valid.  NOTE: the noise bound solveForRotation2D uses is the first call's (see above).
extern "C" int qref_compute_transformation(const float* src, const float* tgt, int L, double noise_bound, double cbar2,
                                           double gnc_factor, int max_iterations, double cost_threshold,
                                           int inlier_selection_mode, double kcore_thr, double cote_noise_bound,
                                           int cote_median, int use_rot_inliers, double* T16, int* clique, int* n_clique,
                                           int* rot_inl, int* n_rot, int* final_inl, int* n_final) {
  Quiet q;
  RefQuatro r;
  auto a = std::make_shared<pcl::PointCloud<pcl::PointXYZ>>(), b = std::make_shared<pcl::PointCloud<pcl::PointXYZ>>();
  for (int i = 0; i < L; ++i) {
    a->points.push_back({src[3 * i], src[3 * i + 1], src[3 * i + 2]});
    b->points.push_back({tgt[3 * i], tgt[3 * i + 1], tgt[3 * i + 2]});
  }
  r.input_ = a;
  r.target_ = b;
  r.params_.noise_bound = noise_bound;
  r.params_.cbar2 = cbar2;
  r.params_.rotation_gnc_factor = gnc_factor;
  r.params_.rotation_max_iterations = (size_t)max_iterations;
  r.params_.rotation_cost_threshold = cost_threshold;
  r.params_.inlier_selection_mode = (RefQuatro::INLIER_SELECTION_MODE)inlier_selection_mode;
  r.params_.kcore_heuristic_threshold = kcore_thr;
  r.params_.cote_mode = cote_median ? "median" : "weighted_mean";
  r.params_.using_rot_inliers_when_estimating_cote = use_rot_inliers != 0;
  r.noise_bound_ = cote_noise_bound;
  Eigen::Matrix4d out = Eigen::Matrix4d::Identity();
  r.solution_.valid = true;
  r.computeTransformation(out);
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T16[4 * i + j] = out(i, j);
  *n_clique = (int)r.max_clique_.size();
  for (size_t i = 0; i < r.max_clique_.size(); ++i) clique[i] = r.max_clique_[i];
  *n_rot = (int)r.rotation_inliers_.size();
  for (size_t i = 0; i < r.rotation_inliers_.size(); ++i) rot_inl[i] = r.rotation_inliers_[i];
  *n_final = r.solution_.valid ? (int)r.final_inliers_.size() : 0;
  for (int i = 0; i < *n_final; ++i) final_inl[i] = r.final_inliers_[(size_t)i];
  return r.solution_.valid ? 1 : 0;
}

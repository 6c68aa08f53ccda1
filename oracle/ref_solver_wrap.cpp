// C entry points around member functions of the REFERENCE's class Quatro (include/quatro.hpp of /root/reference) and its
// include/teaser/utils.h: computeTIMs, solveForScale, solveForRotation2D (+ teaser::utils::svdRot2d),
// solveForTranslation, estimate.  oracle/Makefile (target ref_solver) cuts those definitions out of the header where it
// lies (oracle/ref_extract.py -> a temporary under _ref/), includes them HERE inside a host struct that provides the
// members they touch, and compiles against oracle/ref_shim_solver/ (an Eigen-subset stand-in: eager evaluation,
// index-order reductions, its own small Jacobi SVD).  Test infrastructure: tests/ compare oracle/quatro_oracle.cpp's
// restatement of the back end with this build, and tests/golden/make_golden.py stores its outputs as solver_ref.npz.
#include <cassert>
#include <cmath>
#include <cstddef>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/SVD>
#include "teaser/macros.h"
#include "teaser/utils.h"

namespace {
struct RefQuatro {
  struct Params {  // the fields the functions read (include/quatro.hpp:202-268, same names)
    double noise_bound = 0.3;
    double cbar2 = 1;
    double rotation_gnc_factor = 1.4;
    size_t rotation_max_iterations = 100;
    double rotation_cost_threshold = 1e-6;
  };
  Params params_;
  double noise_bound_ = 0.3;  // :269
  double cost_ = 0;           // :749
  struct {
    bool valid = true;
    double scale = 1;
    Eigen::Vector3d translation;
    Eigen::Matrix3d rotation;
  } solution_;
  Eigen::Matrix<bool, 1, Eigen::Dynamic> scale_inliers_mask_, rotation_inliers_mask_, translation_inliers_mask_;
#include QREF_MEMBERS_INC
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

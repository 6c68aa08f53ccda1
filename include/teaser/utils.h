// teaser/utils.h — drop-in for the helpers of url-kaist/Quatro's include/teaser/utils.h that user code calls around the
// registration classes: svdRot (:123-149), svdRot2d (:151-166), maskVector (:174-186), findNonzero (:192-200),
// calculateDiameter (:109-114).  Host code (a handful of values per call); the weighted rotations use the same closed
// forms as the device path — Horn's quaternion form for 3x3 (include/qtr_math.h) and the angle form for 2x2 — instead
// of Eigen::JacobiSVD; both return the rotation V diag(1, .., det) U^T of the reference to rounding.
// Written against the element accessors only, so it works with Eigen proper and with the stand-in of quatro.hpp.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

#include "../qtr_math.h"
#include "../quatro.hpp"

namespace teaser {
namespace utils {

// rotation R with R * X ~ Y for weighted 3-D pairs (columns of X, Y; weights W)
inline Eigen::Matrix3d svdRot(const Eigen::Matrix<double, 3, Eigen::Dynamic>& X,
                              const Eigen::Matrix<double, 3, Eigen::Dynamic>& Y,
                              const Eigen::Matrix<double, 1, Eigen::Dynamic>& W, int /*static_count*/ = 0) {
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, R[9];
  for (int j = 0; j < static_cast<int>(X.cols()); ++j)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) H[3 * a + b] += (W(0, j) * X(a, j)) * Y(b, j);
  qm_rot3_from_h(H, R);
  Eigen::Matrix3d out;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) out(a, b) = R[3 * a + b];
  return out;
}

inline Eigen::Matrix2d svdRot2d(const Eigen::Matrix<double, 2, Eigen::Dynamic>& X,
                                const Eigen::Matrix<double, 2, Eigen::Dynamic>& Y,
                                const Eigen::Matrix<double, 1, Eigen::Dynamic>& W) {
  double h00 = 0, h01 = 0, h10 = 0, h11 = 0;
  for (int j = 0; j < static_cast<int>(X.cols()); ++j) {
    const double wx0 = W(0, j) * X(0, j), wx1 = W(0, j) * X(1, j);
    h00 += wx0 * Y(0, j);
    h01 += wx0 * Y(1, j);
    h10 += wx1 * Y(0, j);
    h11 += wx1 * Y(1, j);
  }
  const double a = h00 + h11, b = h01 - h10, n = std::sqrt(a * a + b * b);
  const double c = n > 0 ? a / n : 1.0, s = n > 0 ? b / n : 0.0;
  Eigen::Matrix2d out;
  out(0, 0) = c;
  out(0, 1) = -s;
  out(1, 0) = s;
  out(1, 1) = c;
  return out;
}

template <class T>
inline std::vector<T> maskVector(const Eigen::Matrix<bool, 1, Eigen::Dynamic>& mask, const std::vector<T>& elements) {
  std::vector<T> kept;
  for (int i = 0; i < static_cast<int>(mask.cols()) && static_cast<size_t>(i) < elements.size(); ++i)
    if (mask(0, i)) kept.push_back(elements[static_cast<size_t>(i)]);
  return kept;
}

template <class T>
inline std::vector<int> findNonzero(const Eigen::Matrix<T, 1, Eigen::Dynamic>& mask) {
  std::vector<int> idx;
  for (int i = 0; i < static_cast<int>(mask.cols()); ++i)
    if (mask(0, i)) idx.push_back(i);
  return idx;
}

// twice the largest distance of a column from the centre of gravity
template <class T, int D>
float calculateDiameter(const Eigen::Matrix<T, D, Eigen::Dynamic>& X) {
  const int n = static_cast<int>(X.cols()), dim = static_cast<int>(X.rows());
  if (n == 0) return 0.f;
  std::vector<T> cog(static_cast<size_t>(dim), T(0));
  for (int j = 0; j < n; ++j)
    for (int a = 0; a < dim; ++a) cog[static_cast<size_t>(a)] += X(a, j);
  for (int a = 0; a < dim; ++a) cog[static_cast<size_t>(a)] /= static_cast<T>(n);
  T far2 = T(0);
  for (int j = 0; j < n; ++j) {
    T d2 = T(0);
    for (int a = 0; a < dim; ++a) d2 += (X(a, j) - cog[static_cast<size_t>(a)]) * (X(a, j) - cog[static_cast<size_t>(a)]);
    if (d2 > far2) far2 = d2;
  }
  return 2 * std::sqrt(static_cast<float>(far2));
}

}  // namespace utils
}  // namespace teaser

// utils_demo.cpp — teaser::utils::svdRot / svdRot2d / findNonzero / maskVector / calculateDiameter of include/teaser/utils.h.
// Reads "n" then n lines "x0 x1 x2 y0 y1 y2 w" from stdin; prints R3 (9 values), R2 (4 values, from the xy rows), the
// diameter of X, and the indices with w >= 0.5.
#include <cstdio>
#include <vector>

#include "teaser/utils.h"

int main() {
  int n = 0;
  if (std::scanf("%d", &n) != 1 || n < 1) return 2;
  Eigen::Matrix<double, 3, Eigen::Dynamic> X(3, n), Y(3, n);
  Eigen::Matrix<double, 2, Eigen::Dynamic> X2(2, n), Y2(2, n);
  Eigen::Matrix<double, 1, Eigen::Dynamic> W(1, n);
  Eigen::Matrix<bool, 1, Eigen::Dynamic> mask(1, n);
  for (int j = 0; j < n; ++j) {
    double v[7];
    for (double& x : v)
      if (std::scanf("%lf", &x) != 1) return 2;
    for (int a = 0; a < 3; ++a) {
      X(a, j) = v[a];
      Y(a, j) = v[3 + a];
    }
    for (int a = 0; a < 2; ++a) {
      X2(a, j) = v[a];
      Y2(a, j) = v[3 + a];
    }
    W(0, j) = v[6];
    mask(0, j) = v[6] >= 0.5;
  }
  const Eigen::Matrix3d R3 = teaser::utils::svdRot(X, Y, W, 0);
  const Eigen::Matrix2d R2 = teaser::utils::svdRot2d(X2, Y2, W);
  std::printf("R3");
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) std::printf(" %.17g", R3(a, b));
  std::printf("\nR2 %.17g %.17g %.17g %.17g\n", R2(0, 0), R2(0, 1), R2(1, 0), R2(1, 1));
  std::printf("diameter %.9g\n", static_cast<double>(teaser::utils::calculateDiameter<double, 3>(X)));
  std::vector<int> ids(static_cast<size_t>(n));
  for (int j = 0; j < n; ++j) ids[static_cast<size_t>(j)] = 100 + j;
  std::printf("nonzero");
  for (int i : teaser::utils::findNonzero<bool>(mask)) std::printf(" %d", i);
  std::printf("\nmasked");
  for (int i : teaser::utils::maskVector<int>(mask, ids)) std::printf(" %d", i);
  std::printf("\n");
  return 0;
}

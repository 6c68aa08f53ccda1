// qtr_math.h — deterministic scalar math shared by the CPU oracle (g++) and the gfx950 kernels (hipcc).
//
// Why this exists: the hot path's float outputs (FPFH histograms, normals) feed integer decisions
// (histogram bins, nearest-neighbour indices, inlier sets).  libm (glibc) and ROCm's ocml round
// atan2f/acosf/sinf/cosf differently, so "inlier index sets bit-exact" is only achievable if both
// sides evaluate the SAME sequence of IEEE-754 basic operations.  Everything here uses only
// + - * / sqrt on binary64 (correctly rounded on both x86-64 and gfx950) and must be compiled with
// -ffp-contract=off (no FMA contraction) on both compilers.
//
// The functions replace, at the call sites the reference reaches through PCL 1.8.1 (not vendored
// under /root/reference):
//   qm_atan2f  <- atan2f in pcl::computePairFeatures (f1) and std::atan2 in pcl::computeRoots
//   qm_acosf   <- acos(fabs(angle)) role-swap test in pcl::computePairFeatures
//   qm_sincosf <- std::cos/std::sin(theta) in pcl::computeRoots
// Results are the binary64-accurate value rounded once to binary32, i.e. equal to a correctly rounded
// libm result except for ~1e-9 of inputs.
//
// Also: the counter-based RNG that replaces srand(time(NULL))/rand() in the tuple test
// (reference src/teaser_utils/feature_matcher.cc:189-201) and the fixed-shape 64-lane summation
// order used by the GNC-TLS rotation loop (reference include/quatro.hpp:488-531).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define QM_HD __host__ __device__ inline
#else
#define QM_HD inline
#endif

#define QM_PI 3.14159265358979323846
#define QM_PI_2 1.57079632679489661923
#define QM_PI_4 0.78539816339744830962

// atan(x) for finite or +inf x >= 0, binary64, |err| ~ 1e-16.
// Range reduction: x>1 -> pi/2 - atan(1/x); then three half-angle steps
// t <- t / (1 + sqrt(1 + t^2)) bring t below tan(pi/32) ~ 0.0985; odd Taylor series to t^19.
QM_HD double qm_atan_pos(double x) {
  const bool inv = x > 1.0;
  double t = inv ? 1.0 / x : x;
  t = t / (1.0 + sqrt(1.0 + t * t));
  t = t / (1.0 + sqrt(1.0 + t * t));
  t = t / (1.0 + sqrt(1.0 + t * t));
  const double t2 = t * t;
  double s = 1.0 / 19.0;
  s = 1.0 / 17.0 - t2 * s;
  s = 1.0 / 15.0 - t2 * s;
  s = 1.0 / 13.0 - t2 * s;
  s = 1.0 / 11.0 - t2 * s;
  s = 1.0 / 9.0 - t2 * s;
  s = 1.0 / 7.0 - t2 * s;
  s = 1.0 / 5.0 - t2 * s;
  s = 1.0 / 3.0 - t2 * s;
  s = 1.0 - t2 * s;
  const double r = 8.0 * (t * s);
  return inv ? (QM_PI_2 - r) : r;
}

QM_HD double qm_copysign(double mag, double sgn) {
  return signbit(sgn) ? -mag : mag;
}

// C99 atan2 semantics, float in / float out, evaluated in binary64.
QM_HD float qm_atan2f(float yf, float xf) {
  const double y = (double)yf, x = (double)xf;
  if (x != x || y != y) return yf + xf;  // NaN
  if (y == 0.0) return (float)(signbit(x) ? qm_copysign(QM_PI, y) : qm_copysign(0.0, y));
  if (x == 0.0) return (float)qm_copysign(QM_PI_2, y);
  const bool xinf = (x - x) != 0.0, yinf = (y - y) != 0.0;
  if (xinf) {
    if (yinf) return (float)qm_copysign(x > 0 ? QM_PI_4 : 3.0 * QM_PI_4, y);
    return (float)(x > 0 ? qm_copysign(0.0, y) : qm_copysign(QM_PI, y));
  }
  if (yinf) return (float)qm_copysign(QM_PI_2, y);
  double a = qm_atan_pos(fabs(y) / fabs(x));
  if (x < 0) a = QM_PI - a;
  return (float)qm_copysign(a, y);
}

// atan2 in binary64 for finite, not-both-zero arguments (Patchwork's xy2theta); same core as qm_atan2f
QM_HD double qm_atan2d(double y, double x) {
  if (y == 0.0) return signbit(x) ? qm_copysign(QM_PI, y) : qm_copysign(0.0, y);
  if (x == 0.0) return qm_copysign(QM_PI_2, y);
  double a = qm_atan_pos(fabs(y) / fabs(x));
  if (x < 0) a = QM_PI - a;
  return qm_copysign(a, y);
}

// acosf on [-1,1]; NaN outside (as libm). acos(x) = 2*atan(sqrt((1-x)/(1+x))).
QM_HD float qm_acosf(float xf) {
  const double x = (double)xf;
  if (x != x) return xf;
  if (x > 1.0 || x < -1.0) return (float)((x - x) / (x - x) + NAN);
  if (x == -1.0) return (float)QM_PI;
  return (float)(2.0 * qm_atan_pos(sqrt((1.0 - x) / (1.0 + x))));
}

// sin and cos of theta in [0, ~1.2] (computeRoots only produces theta in [0, pi/3]); Taylor series in
// binary64 (degree 25/24), rounded once to float.
QM_HD void qm_sincosf(float thetaf, float* s_out, float* c_out) {
  const double t = (double)thetaf, t2 = t * t;
  // sin: t * (1 - t2/(2*3) * (1 - t2/(4*5) * (...)))
  double s = 1.0;
  s = 1.0 - t2 / (24.0 * 25.0) * s;
  s = 1.0 - t2 / (22.0 * 23.0) * s;
  s = 1.0 - t2 / (20.0 * 21.0) * s;
  s = 1.0 - t2 / (18.0 * 19.0) * s;
  s = 1.0 - t2 / (16.0 * 17.0) * s;
  s = 1.0 - t2 / (14.0 * 15.0) * s;
  s = 1.0 - t2 / (12.0 * 13.0) * s;
  s = 1.0 - t2 / (10.0 * 11.0) * s;
  s = 1.0 - t2 / (8.0 * 9.0) * s;
  s = 1.0 - t2 / (6.0 * 7.0) * s;
  s = 1.0 - t2 / (4.0 * 5.0) * s;
  s = 1.0 - t2 / (2.0 * 3.0) * s;
  double c = 1.0;
  c = 1.0 - t2 / (23.0 * 24.0) * c;
  c = 1.0 - t2 / (21.0 * 22.0) * c;
  c = 1.0 - t2 / (19.0 * 20.0) * c;
  c = 1.0 - t2 / (17.0 * 18.0) * c;
  c = 1.0 - t2 / (15.0 * 16.0) * c;
  c = 1.0 - t2 / (13.0 * 14.0) * c;
  c = 1.0 - t2 / (11.0 * 12.0) * c;
  c = 1.0 - t2 / (9.0 * 10.0) * c;
  c = 1.0 - t2 / (7.0 * 8.0) * c;
  c = 1.0 - t2 / (5.0 * 6.0) * c;
  c = 1.0 - t2 / (3.0 * 4.0) * c;
  c = 1.0 - t2 / (1.0 * 2.0) * c;
  *s_out = (float)(t * s);
  *c_out = (float)c;
}

// Rotation R (row-major 3x3) maximising trace(R * H) for H = sum_j w_j x_j y_j^T (row-major: H[3a+b] = sum w x_a y_b)
// — the rotation teaser::utils::svdRot returns (reference include/teaser/utils.h:123-149: V diag(1,1,det) U^T of
// H = U S V^T).  Horn's unit-quaternion form instead of Eigen::JacobiSVD: the dominant eigenvector of the symmetric
// 4x4 N(H), by cyclic Jacobi rotations with a fixed sweep count (binary64 + - * / sqrt only, so host and device agree
// bit for bit).  Always a proper rotation; H = 0 gives the identity.
QM_HD void qm_rot3_from_h(const double* H, double* R) {
  const double Sxx = H[0], Sxy = H[1], Sxz = H[2], Syx = H[3], Syy = H[4], Syz = H[5], Szx = H[6], Szy = H[7], Szz = H[8];
  double A[4][4] = {{(Sxx + Syy) + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                    {Syz - Szy, (Sxx - Syy) - Szz, Sxy + Syx, Szx + Sxz},
                    {Szx - Sxz, Sxy + Syx, (Syy - Sxx) - Szz, Syz + Szy},
                    {Sxy - Syx, Szx + Sxz, Syz + Szy, (Szz - Sxx) - Syy}};
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int sweep = 0; sweep < 10; ++sweep) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int p = 0; p < 3; ++p) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int q = p + 1; q < 4; ++q) {
        const double apq = A[p][q];
        if (apq == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        const double at = theta < 0 ? -theta : theta;
        double t = 1.0 / (at + sqrt(theta * theta + 1.0));
        if (theta < 0) t = -t;
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k < 4; ++k) {  // columns p, q of A and V
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - sn * akq;
          A[k][q] = sn * akp + c * akq;
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - sn * vkq;
          V[k][q] = sn * vkp + c * vkq;
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k < 4; ++k) {  // rows p, q of A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - sn * aqk;
          A[q][k] = sn * apk + c * aqk;
        }
      }
    }
  }
  double q0 = V[0][0], q1 = V[1][0], q2 = V[2][0], q3 = V[3][0], best = A[0][0];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 1; k < 4; ++k)
    if (A[k][k] > best) {
      best = A[k][k];
      q0 = V[0][k];
      q1 = V[1][k];
      q2 = V[2][k];
      q3 = V[3][k];
    }
  const double nq = sqrt((q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3));
  q0 = q0 / nq;
  q1 = q1 / nq;
  q2 = q2 / nq;
  q3 = q3 / nq;
  R[0] = ((q0 * q0 + q1 * q1) - q2 * q2) - q3 * q3;
  R[1] = 2.0 * (q1 * q2 - q0 * q3);
  R[2] = 2.0 * (q1 * q3 + q0 * q2);
  R[3] = 2.0 * (q1 * q2 + q0 * q3);
  R[4] = ((q0 * q0 - q1 * q1) + q2 * q2) - q3 * q3;
  R[5] = 2.0 * (q2 * q3 - q0 * q1);
  R[6] = 2.0 * (q1 * q3 - q0 * q2);
  R[7] = 2.0 * (q2 * q3 + q0 * q1);
  R[8] = ((q0 * q0 - q1 * q1) - q2 * q2) + q3 * q3;
}

// ---------------------------------------------------------------------------------------------
// Counter-based RNG (SplitMix64 finaliser over seed + counter).  Replaces srand(time(NULL))/rand()
// in the tuple test: trial t draws r_k = qm_rand_u32(seed, 3*t + k) % ncorr, k = 0,1,2.
QM_HD uint64_t qm_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
QM_HD uint32_t qm_rand_u32(uint64_t seed, uint64_t counter) {
  return (uint32_t)(qm_mix64(qm_mix64(seed) ^ (counter * 0xD1342543DE82EF95ULL)) >> 32);
}

// ---------------------------------------------------------------------------------------------
// Fixed-shape summation ("sum64"): the order in which the GNC-TLS loop's reductions are evaluated
// on both sides.  Element j is accumulated sequentially into partial[j & 63] (ascending j), then the
// 64 partials are folded by the shfl_down butterfly: for off = 32,16,...,1: p[l] += p[l+off], l<off.
// qm_sum64_fold performs the fold on a 64-entry array (host oracle); the kernels do the same fold
// with __shfl_down across one wavefront.
QM_HD double qm_sum64_fold(double* p /*[64], clobbered*/) {
  for (int off = 32; off >= 1; off >>= 1)
    for (int l = 0; l < off; ++l) p[l] = p[l] + p[l + off];
  return p[0];
}
// the same fold for binary32 partials (Patchwork's per-patch moment sums)
QM_HD float qm_sum64_fold_f(float* p /*[64], clobbered*/) {
  for (int off = 32; off >= 1; off >>= 1)
    for (int l = 0; l < off; ++l) p[l] = p[l] + p[l + off];
  return p[0];
}

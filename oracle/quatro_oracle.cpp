// quatro_oracle.cpp — CPU restatement ("oracle") of url-kaist/Quatro's registration hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may build, load or call it.  The product path (quatro_amd/ +
// libquatro_hip.so) never links or calls anything in oracle/.
//
// PINNING STATUS.  The reference ships no tests, golden vectors or fixtures, cannot be built as a whole
// here (PCL / FLANN / Eigen / PMC absent) and is itself non-deterministic (time-seeded tuple test,
// 12-thread racy clique heuristic, unstable sorts).  Two parts of its own text DO compile here, against
// stand-ins for the absent libraries, and pin the corresponding functions of this file (oracle/Makefile targets `ref`
// and `ref_solver`, tests/test_ref_cpu.py, tests/golden/matcher_ref.npz and solver_ref.npz):
//   * teaser::Matcher (src/teaser_utils/feature_matcher.cc)            -> match(): identical correspondence lists;
//   * Quatro::computeTIMs / solveForScale / solveForRotation[2D] / solveForTranslation / estimate / computeTransformation,
//     teaser::Graph and teaser::utils::svdRot2d (include/quatro.hpp, include/teaser/*.h; Eigen replaced by a small eager
//     stand-in, PMC's clique search answered by this file's)            -> build_graph(): identical edges; COTE and the
//        translation: identical bits; GNC-TLS yaw: identical inlier sets, rotation / cost to rounding (D6); solve():
//        identical clique / rotation-inlier / final-inlier lists, the 4 x 4 to 4e-15.
// PARITY UNPINNED for the rest — voxel grid, normals / FPFH (PCL), core numbers and the clique search (PMC): there this
// file DEFINES the deterministic semantics the GPU path is compared against.  It follows
// the reference's in-tree code line by line where that exists and restates the published algorithms of
// the un-vendored dependencies (PCL 1.8.1, FLANN 1.9.1, Eigen 3.3, PMC tag `libpmc`) at the
// reference's call sites.  Each function cites what it follows.  Declared divergences:
//   D1 tuple-test RNG: counter-based qm_rand_u32(seed, 3*trial+k) instead of srand(time)/rand()
//      (reference src/teaser_utils/feature_matcher.cc:189,199-201).
//   D2 clique heuristic: single-thread sequential semantics instead of 12 OpenMP threads with a
//      dynamic schedule (reference src/graph.cc:39).  Outer vertex order: by default the canonical
//      (core number, vertex id) ascending order traversed from the back ("canonical"); the
//      Batagelj-Zaversnik bucket order PMC produces is available as order=1 ("bz") for comparison.
//      Within-core order is racy in the reference itself, so no order is "the" reference order.
//   D3 ties: NN search -> lowest index; std::sort sites -> stable by (key, original position).
//   D4 COTE median with n_card<=1 (UB in reference include/quatro.hpp:714-730) -> defined.
//   D5 GNC rotation noise bound taken from the current params, not a function-local static
//      (reference include/quatro.hpp:469-470); identical in the single-call demo flow.
//   D6 2x2 weighted rotation in closed form instead of Eigen::JacobiSVD (utils.h:151-166); the
//      H and cost reductions use the fixed 64-lane order of qtr_math.h (Eigen's GEMM order is
//      build-dependent).  Agreement with an SVD evaluation is checked in tests to 1e-12.
//   D7 libm atan2f/acosf/sinf/cosf replaced by the binary64 evaluations of qtr_math.h.
//   D8 stdout prints removed.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <utility>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/qtr_math.h"

namespace {

int g_threads = 1;

struct P3 {
  float x, y, z;
};
inline P3 ld4(const float* a, int i) { return P3{a[4 * i], a[4 * i + 1], a[4 * i + 2]}; }

// ================================================================================================
// K1  voxel-grid down-sampling.  Follows pcl::VoxelGrid<PointXYZ>::applyFilter (PCL 1.8.1) as called
// from voxelize(), reference include/quatro.hpp:49-68 (leaf given as double, stored as float).
// Within-voxel accumulation order: ascending original point index (PCL's std::sort is unstable; D3).
// Returns n (voxels), or -1 if the grid would overflow int32 (PCL then passes the input through).
int voxelize(const float* xyz4, int P, float leaf, float* out4, int cap) {
  if (P <= 0) return 0;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < P; ++i)
    for (int a = 0; a < 3; ++a) {
      float v = xyz4[4 * i + a];
      mn[a] = v < mn[a] ? v : mn[a];
      mx[a] = v > mx[a] ? v : mx[a];
    }
  const float inv = 1.0f / leaf;  // inverse_leaf_size_ = Array4f::Ones() / leaf_size_
  int64_t d[3];
  for (int a = 0; a < 3; ++a) d[a] = (int64_t)((mx[a] - mn[a]) * inv) + 1;
  if (d[0] * d[1] * d[2] > (int64_t)std::numeric_limits<int32_t>::max()) return -1;
  int minb[3], maxb[3], divb[3];
  for (int a = 0; a < 3; ++a) {
    minb[a] = (int)floorf(mn[a] * inv);
    maxb[a] = (int)floorf(mx[a] * inv);
    divb[a] = maxb[a] - minb[a] + 1;
  }
  const int mul1 = divb[0], mul2 = divb[0] * divb[1];
  std::vector<std::pair<uint32_t, int>> iv(P);
  for (int i = 0; i < P; ++i) {
    int i0 = (int)(floorf(xyz4[4 * i] * inv) - (float)minb[0]);
    int i1 = (int)(floorf(xyz4[4 * i + 1] * inv) - (float)minb[1]);
    int i2 = (int)(floorf(xyz4[4 * i + 2] * inv) - (float)minb[2]);
    iv[i] = {(uint32_t)(i0 + i1 * mul1 + i2 * mul2), i};
  }
  std::sort(iv.begin(), iv.end());  // (idx, point) lexicographic == stable by idx
  int n = 0;
  for (int s = 0; s < P;) {
    int e = s;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    while (e < P && iv[e].first == iv[s].first) {
      cx += xyz4[4 * iv[e].second];
      cy += xyz4[4 * iv[e].second + 1];
      cz += xyz4[4 * iv[e].second + 2];
      ++e;
    }
    const float cnt = (float)(e - s);
    if (n < cap) {
      out4[4 * n] = cx / cnt;
      out4[4 * n + 1] = cy / cnt;
      out4[4 * n + 2] = cz / cnt;
      out4[4 * n + 3] = 0.f;
    }
    ++n;
    s = e;
  }
  return n;
}

// ================================================================================================
// Radius search.  Semantics of pcl::search::KdTree -> KdTreeFLANN::radiusSearch -> FLANN
// KDTreeSingleIndex + RadiusResultSet (sorted): squared distance by flann::L2_Simple<float>
// (((0+dx^2)+dy^2)+dz^2), kept iff d2 < float(double(r)*double(r)), results sorted ascending by
// (d2, index) (FLANN DistanceIndex::operator<).  The query point itself is included.
// Implementation: uniform hash grid (cell = r); exact, order-independent because of the final sort.
struct Neighbors {
  std::vector<int64_t> off;  // n+1
  std::vector<int> idx;
  std::vector<float> d2;
};

inline float l2_simple3(const P3& a, const P3& b) {
  float r = 0.f, d;
  d = a.x - b.x;
  r += d * d;
  d = a.y - b.y;
  r += d * d;
  d = a.z - b.z;
  r += d * d;
  return r;
}

void radius_neighbors(const float* xyz4, int n, double radius, Neighbors& nb) {
  const float r2 = (float)(radius * radius);
  const float cell = (float)radius * 1.001f;  // margin: float cell rounding can never hide a d2 < r2 pair
  float mn[3] = {INFINITY, INFINITY, INFINITY};
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) mn[a] = std::min(mn[a], xyz4[4 * i + a]);
  auto cellof = [&](const P3& p, int* c) {
    c[0] = (int)floorf((p.x - mn[0]) / cell);
    c[1] = (int)floorf((p.y - mn[1]) / cell);
    c[2] = (int)floorf((p.z - mn[2]) / cell);
  };
  auto keyof = [](int a, int b, int c) {
    return ((uint64_t)(uint32_t)(a + 1) << 42) ^ ((uint64_t)(uint32_t)(b + 1) << 21) ^ (uint64_t)(uint32_t)(c + 1);
  };
  std::unordered_map<uint64_t, std::vector<int>> grid;
  grid.reserve((size_t)n);
  for (int i = 0; i < n; ++i) {
    int c[3];
    cellof(ld4(xyz4, i), c);
    grid[keyof(c[0], c[1], c[2])].push_back(i);
  }
  std::vector<std::vector<std::pair<uint64_t, int>>> lists((size_t)n);
#pragma omp parallel for schedule(dynamic, 64) num_threads(g_threads)
  for (int i = 0; i < n; ++i) {
    const P3 p = ld4(xyz4, i);
    int c[3];
    cellof(p, c);
    auto& L = lists[i];
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          auto it = grid.find(keyof(c[0] + dx, c[1] + dy, c[2] + dz));
          if (it == grid.end()) continue;
          for (int j : it->second) {
            const float d2 = l2_simple3(p, ld4(xyz4, j));  // query first, as FLANN distance_(vec, point)
            if (d2 < r2) {
              uint32_t bits;
              memcpy(&bits, &d2, 4);
              L.push_back({((uint64_t)bits << 32) | (uint32_t)j, j});
            }
          }
        }
    std::sort(L.begin(), L.end());
  }
  nb.off.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) nb.off[i + 1] = nb.off[i] + (int64_t)lists[i].size();
  nb.idx.resize((size_t)nb.off[n]);
  nb.d2.resize((size_t)nb.off[n]);
  for (int i = 0; i < n; ++i) {
    int64_t o = nb.off[i];
    for (auto& e : lists[i]) {
      uint32_t bits = (uint32_t)(e.first >> 32);
      float d2;
      memcpy(&d2, &bits, 4);
      nb.idx[o] = e.second;
      nb.d2[o] = d2;
      ++o;
    }
  }
}

// ================================================================================================
// K2  surface normals.  Follows pcl::NormalEstimation<PointXYZ,Normal>::computeFeature (single
// thread class chosen at reference src/teaser_utils/fpfh.cc:58-63) -> computePointNormal ->
// computeMeanAndCovarianceMatrix (single-pass float, 9 accumulators) -> solvePlaneParameters ->
// pcl::eigen33 (smallest eigenpair) -> flipNormalTowardsViewpoint(vp = 0,0,0).
void compute_roots2(float b, float c, float* roots) {
  roots[0] = 0.f;
  float d = (float)((double)(b * b) - 4.0 * (double)c);
  if (d < 0.0f) d = 0.0f;
  const float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

void compute_roots(const float m[9], float* roots) {
  // characteristic polynomial x^3 - c2 x^2 + c1 x - c0
  const float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] -
                   m[8] * m[1] * m[1];
  const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  const float c2 = m[0] + m[4] + m[8];
  if (fabsf(c0) < std::numeric_limits<float>::epsilon()) {
    compute_roots2(c2, c1, roots);
    return;
  }
  const float s_inv3 = (float)(1.0 / 3.0);
  const float s_sqrt3 = sqrtf(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.f) a_over_3 = 0.f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.f) q = 0.f;
  const float rho = sqrtf(-a_over_3);
  const float theta = qm_atan2f(sqrtf(-q), half_b) * s_inv3;
  float sin_theta, cos_theta;
  qm_sincosf(theta, &sin_theta, &cos_theta);
  roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
  if (roots[1] >= roots[2]) {
    std::swap(roots[1], roots[2]);
    if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
  }
  if (roots[0] <= 0.f) compute_roots2(c2, c1, roots);
}

inline void cross3(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// Eigen fixed-size-3 reduction order (redux_novec_unroller): e0 + (e1 + e2).
inline float sum3_tree(float a, float b, float c) { return a + (b + c); }

void eigen33_smallest(const float cov[9], float* eval, float* evec) {
  float scale = 0.f;
  for (int i = 0; i < 9; ++i) scale = std::max(scale, fabsf(cov[i]));
  if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
  float s[9];
  for (int i = 0; i < 9; ++i) s[i] = cov[i] / scale;
  float roots[3];
  compute_roots(s, roots);
  *eval = roots[0] * scale;
  s[0] -= roots[0];
  s[4] -= roots[0];
  s[8] -= roots[0];
  float v1[3], v2[3], v3[3];
  cross3(&s[0], &s[3], v1);
  cross3(&s[0], &s[6], v2);
  cross3(&s[3], &s[6], v3);
  const float l1 = sum3_tree(v1[0] * v1[0], v1[1] * v1[1], v1[2] * v1[2]);
  const float l2 = sum3_tree(v2[0] * v2[0], v2[1] * v2[1], v2[2] * v2[2]);
  const float l3 = sum3_tree(v3[0] * v3[0], v3[1] * v3[1], v3[2] * v3[2]);
  const float* v;
  float l;
  if (l1 >= l2 && l1 >= l3) {
    v = v1;
    l = l1;
  } else if (l2 >= l1 && l2 >= l3) {
    v = v2;
    l = l2;
  } else {
    v = v3;
    l = l3;
  }
  const float sl = sqrtf(l);
  evec[0] = v[0] / sl;
  evec[1] = v[1] / sl;
  evec[2] = v[2] / sl;
}

// normals4: nx, ny, nz, curvature per point.  nbf = neighbours for the FPFH radius (sorted by d2);
// the normal-radius neighbourhood is its prefix with d2 < rn2 (rn <= rf is enforced by the caller,
// reference include/fpfh_manager.hpp:99-102).
void normals_from_neighbors(const float* xyz4, int n, const Neighbors& nbf, float rn2, float* normals4) {
  const float qnan = std::numeric_limits<float>::quiet_NaN();
  for (int i = 0; i < n; ++i) {  // single-threaded in the reference (pcl::NormalEstimation)
    int64_t b = nbf.off[i], e = nbf.off[i + 1];
    int k = 0;
    while (b + k < e && nbf.d2[b + k] < rn2) ++k;
    float* o = normals4 + 4 * i;
    if (k < 3) {
      o[0] = o[1] = o[2] = o[3] = qnan;
      continue;
    }
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < k; ++t) {
      const P3 q = ld4(xyz4, nbf.idx[b + t]);
      acc[0] += q.x * q.x;
      acc[1] += q.x * q.y;
      acc[2] += q.x * q.z;
      acc[3] += q.y * q.y;
      acc[4] += q.y * q.z;
      acc[5] += q.z * q.z;
      acc[6] += q.x;
      acc[7] += q.y;
      acc[8] += q.z;
    }
    const float kf = (float)k;
    for (int t = 0; t < 9; ++t) acc[t] /= kf;
    float cov[9];
    cov[0] = acc[0] - acc[6] * acc[6];
    cov[1] = acc[1] - acc[6] * acc[7];
    cov[2] = acc[2] - acc[6] * acc[8];
    cov[4] = acc[3] - acc[7] * acc[7];
    cov[5] = acc[4] - acc[7] * acc[8];
    cov[8] = acc[5] - acc[8] * acc[8];
    cov[3] = cov[1];
    cov[6] = cov[2];
    cov[7] = cov[5];
    float ev, vec[3];
    eigen33_smallest(cov, &ev, vec);
    const float eig_sum = cov[0] + cov[4] + cov[8];
    float curv = (eig_sum != 0.f) ? fabsf(ev / eig_sum) : 0.f;
    // flipNormalTowardsViewpoint, viewpoint (0,0,0)
    const P3 p = ld4(xyz4, i);
    const float vx = 0.f - p.x, vy = 0.f - p.y, vz = 0.f - p.z;
    const float cos_theta = (vx * vec[0] + vy * vec[1] + vz * vec[2]);
    if (cos_theta < 0) {
      vec[0] *= -1;
      vec[1] *= -1;
      vec[2] *= -1;
    }
    o[0] = vec[0];
    o[1] = vec[1];
    o[2] = vec[2];
    o[3] = curv;
  }
}

// ================================================================================================
// K3  SPFH.  Follows pcl::computePairFeatures + FPFHEstimation::computePointSPFHSignature (PCL 1.8.1)
// as reached from FPFHEstimationOMP::computeFeature (reference src/teaser_utils/fpfh.cc:68-72).
// Eigen Vector4f dot/norm reductions use the SSE packet order (e0+e2)+(e1+e3) with e3 = 0.
inline float dot4_sse(const float* a, const float* b) { return (a[0] * b[0] + a[2] * b[2]) + (a[1] * b[1] + 0.0f); }

// returns false when the pair is skipped; f[0..2] = f1,f2,f3
bool pair_features(const P3& p1, const float* n1, const P3& p2, const float* n2, float* f) {
  float dp[3] = {p2.x - p1.x, p2.y - p1.y, p2.z - p1.z};
  const float f4 = sqrtf(dot4_sse(dp, dp));
  if (f4 == 0.0f) return false;
  float n1c[3] = {n1[0], n1[1], n1[2]}, n2c[3] = {n2[0], n2[1], n2[2]};
  const float angle1 = dot4_sse(n1c, dp) / f4;
  const float angle2 = dot4_sse(n2c, dp) / f4;
  float f3;
  if (qm_acosf(fabsf(angle1)) > qm_acosf(fabsf(angle2))) {
    for (int a = 0; a < 3; ++a) {
      n1c[a] = n2[a];
      n2c[a] = n1[a];
      dp[a] *= -1.f;
    }
    f3 = -angle2;
  } else
    f3 = angle1;
  float v[3];
  cross3(dp, n1c, v);
  const float v_norm = sqrtf(dot4_sse(v, v));
  if (v_norm == 0.0f) return false;
  v[0] /= v_norm;
  v[1] /= v_norm;
  v[2] /= v_norm;
  float w[3];
  cross3(n1c, v, w);
  f[1] = dot4_sse(v, n2c);
  f[0] = qm_atan2f(dot4_sse(w, n2c), dot4_sse(n1c, n2c));
  f[2] = f3;
  return true;
}

inline int bin11(double x) {  // static_cast<int>(floor(x)) with NaN -> 0 (x86 cvttsd2si gives INT_MIN, clamped to 0)
  if (x != x) return 0;
  double fl = floor(x);
  if (fl < 0.0) return 0;
  if (fl >= 11.0) return 10;
  return (int)fl;
}

void spfh_from_neighbors(const float* xyz4, const float* normals4, int n, const Neighbors& nb, float* spfh33) {
  const float d_pi = 1.0f / (2.0f * (float)M_PI);
  memset(spfh33, 0, sizeof(float) * 33 * (size_t)n);
#pragma omp parallel for schedule(dynamic, 64) num_threads(g_threads)
  for (int i = 0; i < n; ++i) {
    const int64_t b = nb.off[i], e = nb.off[i + 1];
    const int k = (int)(e - b);
    float* h = spfh33 + 33 * (size_t)i;
    const float hist_incr = 100.0f / (float)(k - 1);
    const P3 p = ld4(xyz4, i);
    for (int64_t t = b; t < e; ++t) {
      const int j = nb.idx[t];
      if (j == i) continue;
      float f[3];
      if (!pair_features(p, normals4 + 4 * i, ld4(xyz4, j), normals4 + 4 * j, f)) continue;
      h[bin11(11 * (((double)f[0] + M_PI) * (double)d_pi))] += hist_incr;
      h[11 + bin11(11 * (((double)f[1] + 1.0) * 0.5))] += hist_incr;
      h[22 + bin11(11 * (((double)f[2] + 1.0) * 0.5))] += hist_incr;
    }
  }
}

// K4  FPFH weighting.  Follows FPFHEstimation::weightPointSPFHSignature (PCL 1.8.1).
void fpfh_from_spfh(const float* spfh33, int n, const Neighbors& nb, float* desc33) {
#pragma omp parallel for schedule(dynamic, 64) num_threads(g_threads)
  for (int i = 0; i < n; ++i) {
    float* o = desc33 + 33 * (size_t)i;
    for (int t = 0; t < 33; ++t) o[t] = 0.f;
    double sum[3] = {0.0, 0.0, 0.0};
    for (int64_t t = nb.off[i]; t < nb.off[i + 1]; ++t) {
      if (nb.d2[t] == 0) continue;
      const float weight = 1.0f / nb.d2[t];
      const float* s = spfh33 + 33 * (size_t)nb.idx[t];
      for (int blk = 0; blk < 3; ++blk)
        for (int c = 0; c < 11; ++c) {
          const float val = s[11 * blk + c] * weight;
          sum[blk] += val;
          o[11 * blk + c] += val;
        }
    }
    for (int blk = 0; blk < 3; ++blk) {
      if (sum[blk] != 0) sum[blk] = 100.0 / sum[blk];
      for (int c = 0; c < 11; ++c) o[11 * blk + c] *= (float)sum[blk];
    }
  }
}

// ================================================================================================
// K5  33-D nearest neighbour.  flann::L2<float>: groups of four, result += ((d0^2+d1^2)+d2^2)+d3^2,
// then the 1-element tail (dim 33).  Exact (KDTreeSingleIndex, eps = 0); ties -> lowest index (D3).
inline float l2_flann33(const float* a, const float* b) {
  float result = 0.f;
  for (int g = 0; g < 8; ++g) {
    const float d0 = a[4 * g] - b[4 * g], d1 = a[4 * g + 1] - b[4 * g + 1], d2 = a[4 * g + 2] - b[4 * g + 2],
                d3 = a[4 * g + 3] - b[4 * g + 3];
    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  const float d = a[32] - b[32];
  result += d * d;
  return result;
}

int nn33(const float* query, const float* data, int n) {
  int best = -1;
  float bd = INFINITY;
  for (int i = 0; i < n; ++i) {
    const float d = l2_flann33(query, data + 33 * (size_t)i);
    if (d < bd) {  // KNNSimpleResultSet::addPoint: strict improvement only
      bd = d;
      best = i;
    }
  }
  return best < 0 ? 0 : best;
}

// K5-K8  Follows teaser::Matcher::calculateCorrespondences (reference
// include/teaser_utils/feature_matcher.h:42-74) -> normalizePoints (feature_matcher.cc:18-76,
// use_absolute_scale = true) -> advancedMatching (feature_matcher.cc:77-265).
// nn_large_of_small / nn_small_of_large (optional, sized n_small / n_large) expose the raw NN tables.
int match(const float* xyz_s, int ns, const float* desc_s, const float* xyz_t, int nt, const float* desc_t,
          int use_crosscheck, int use_tuple, float tuple_scale, uint64_t seed, int* corr, int cap, int* dbg_nn_i_of_j,
          int* dbg_nn_j_of_i) {
  // --- normalizePoints: mean-centred float copies (sequential float sums)
  const float* xyz[2] = {xyz_s, xyz_t};
  const int np[2] = {ns, nt};
  std::vector<P3> pc[2];
  for (int c = 0; c < 2; ++c) {
    float mx = 0.f, my = 0.f, mz = 0.f;
    for (int i = 0; i < np[c]; ++i) {
      mx = mx + xyz[c][4 * i];
      my = my + xyz[c][4 * i + 1];
      mz = mz + xyz[c][4 * i + 2];
    }
    const float inv_n = (float)np[c];
    mx = mx / inv_n;
    my = my / inv_n;
    mz = mz / inv_n;
    pc[c].resize((size_t)np[c]);
    for (int i = 0; i < np[c]; ++i) pc[c][i] = P3{xyz[c][4 * i] - mx, xyz[c][4 * i + 1] - my, xyz[c][4 * i + 2] - mz};
  }
  const float* feat[2] = {desc_s, desc_t};
  int fi = 0, fj = 1;
  bool swapped = false;
  if (np[fj] > np[fi]) {
    std::swap(fi, fj);
    swapped = true;
  }
  const int nPti = np[fi], nPtj = np[fj];
  if (nPti == 0 || nPtj == 0) return 0;
  // --- initial matching: i = NN_i(f_j) for every j; j' = NN_j(f_i) for every hit i
  std::vector<int> nn_i_of_j((size_t)nPtj), i_to_j((size_t)nPti, -1);
#pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads)
  for (int j = 0; j < nPtj; ++j) nn_i_of_j[j] = nn33(feat[fj] + 33 * (size_t)j, feat[fi], nPti);
  std::vector<char> hit((size_t)nPti, 0);
  for (int j = 0; j < nPtj; ++j) hit[nn_i_of_j[j]] = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads)
  for (int i = 0; i < nPti; ++i)
    if (hit[i]) i_to_j[i] = nn33(feat[fi] + 33 * (size_t)i, feat[fj], nPtj);
  if (dbg_nn_i_of_j) memcpy(dbg_nn_i_of_j, nn_i_of_j.data(), sizeof(int) * (size_t)nPtj);
  if (dbg_nn_j_of_i) memcpy(dbg_nn_j_of_i, i_to_j.data(), sizeof(int) * (size_t)nPti);
  std::vector<std::pair<int, int>> corres;
  if (use_crosscheck) {
    // (i, j) kept iff i_to_j[i] == j and nn_i_of_j[j] == i; emitted in ascending i
    for (int i = 0; i < nPti; ++i) {
      const int j = i_to_j[i];
      if (j >= 0 && nn_i_of_j[j] == i) corres.push_back({i, j});
    }
  } else {
    for (int i = 0; i < nPti; ++i)
      if (i_to_j[i] != -1) corres.push_back({i, i_to_j[i]});
    for (int j = 0; j < nPtj; ++j) corres.push_back({nn_i_of_j[j], j});
  }
  // --- tuple constraint (feature_matcher.cc:187-247), RNG per D1
  if (use_tuple && tuple_scale != 0) {
    const int ncorr = (int)corres.size();
    const int64_t trials = (int64_t)ncorr * 100;
    const float scale = tuple_scale;
    std::vector<std::pair<int, int>> tup;
    auto norm3 = [](const P3& a, const P3& b) {
      const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
      return sqrtf(sum3_tree(dx * dx, dy * dy, dz * dz));
    };
    for (int64_t t = 0; t < trials; ++t) {
      const int r0 = (int)(qm_rand_u32(seed, 3 * (uint64_t)t) % (uint32_t)ncorr);
      const int r1 = (int)(qm_rand_u32(seed, 3 * (uint64_t)t + 1) % (uint32_t)ncorr);
      const int r2 = (int)(qm_rand_u32(seed, 3 * (uint64_t)t + 2) % (uint32_t)ncorr);
      const int idi0 = corres[r0].first, idj0 = corres[r0].second;
      const int idi1 = corres[r1].first, idj1 = corres[r1].second;
      const int idi2 = corres[r2].first, idj2 = corres[r2].second;
      const float li0 = norm3(pc[fi][idi0], pc[fi][idi1]);
      const float li1 = norm3(pc[fi][idi1], pc[fi][idi2]);
      const float li2 = norm3(pc[fi][idi2], pc[fi][idi0]);
      const float lj0 = norm3(pc[fj][idj0], pc[fj][idj1]);
      const float lj1 = norm3(pc[fj][idj1], pc[fj][idj2]);
      const float lj2 = norm3(pc[fj][idj2], pc[fj][idj0]);
      if ((li0 * scale < lj0) && (lj0 < li0 / scale) && (li1 * scale < lj1) && (lj1 < li1 / scale) &&
          (li2 * scale < lj2) && (lj2 < li2 / scale)) {
        tup.push_back({idi0, idj0});
        tup.push_back({idi1, idj1});
        tup.push_back({idi2, idj2});
      }
    }
    corres.swap(tup);
  }
  if (swapped)
    for (auto& c : corres) std::swap(c.first, c.second);
  std::sort(corres.begin(), corres.end());
  corres.erase(std::unique(corres.begin(), corres.end()), corres.end());
  const int L = (int)corres.size();
  for (int i = 0; i < L && i < cap; ++i) {
    corr[2 * i] = corres[i].first;
    corr[2 * i + 1] = corres[i].second;
  }
  return L;
}

// ================================================================================================
// K9-K11  pairwise-consistency graph as a bit matrix.  Follows Quatro::computeTIMs
// (reference include/quatro.hpp:307-344), solveForScale (:355-386) and the addEdge loop (:784-789).
// Points are float at the API (pcl::PointXYZ) and widened to double by pcl2eigen
// (reference include/conversion.hpp:38-44).  Column norms: Eigen fixed-3 reduction e0+(e1+e2).
inline double tim_norm(const double* a, const double* b) {  // || b - a ||
  const double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
  return sqrt(dx * dx + (dy * dy + dz * dz));
}
inline bool scale_consistent(double v1, double v2, double beta) {
  const bool fwd = fabs(v2 / v1 - 1.0) <= beta * (1.0 / v1);  // beta * v1_dist.cwiseInverse()
  const bool rev = fabs(v1 / v2 - 1.0) <= beta * (1.0 / v2);
  return fwd && rev;
}

int graph_words(int L) { return (L + 63) / 64; }

void build_graph(const double* src3, const double* tgt3, int L, double noise_bound, double cbar2, uint64_t* bm) {
  const int W = graph_words(L);
  memset(bm, 0, sizeof(uint64_t) * (size_t)W * (size_t)L);
  const double beta = 2 * noise_bound * sqrt(cbar2);
#pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads)
  for (int i = 0; i < L; ++i)
    for (int j = 0; j < L; ++j) {
      if (i == j) continue;
      const int lo = i < j ? i : j, hi = i < j ? j : i;  // TIM is v_hi - v_lo
      const double a = tim_norm(src3 + 3 * lo, src3 + 3 * hi);
      const double b = tim_norm(tgt3 + 3 * lo, tgt3 + 3 * hi);
      if (scale_consistent(a, b, beta)) bm[(size_t)i * W + (j >> 6)] |= (1ULL << (j & 63));
    }
}

// ================================================================================================
// K12  max-clique heuristic.  Restates teaser::MaxCliqueSolver::findMaxClique (reference
// src/graph.cc:12-104) over PMC (tag `libpmc`, not vendored): pmc_graph::compute_cores
// (Batagelj-Zaversnik, 1-shifted ids, kcore = core+1) and pmc_heu::search_bounds / branch with
// heu_strat "kcore", single thread (D2).
struct Csr {
  std::vector<long long> off;
  std::vector<int> adj;
};
void bitmap_to_csr(const uint64_t* bm, int L, Csr& g) {
  const int W = graph_words(L);
  g.off.assign((size_t)L + 1, 0);
  g.adj.clear();
  for (int i = 0; i < L; ++i) {
    for (int w = 0; w < W; ++w) {
      uint64_t x = bm[(size_t)i * W + w];
      while (x) {
        const int b = __builtin_ctzll(x);
        g.adj.push_back(w * 64 + b);
        x &= x - 1;
      }
    }
    g.off[i + 1] = (long long)g.adj.size();
  }
}

// returns max_core; kcore[v] = core(v)+1 for v in [0,V); order[] = BZ removal order (0-based ids)
int compute_cores_bz(const Csr& g, std::vector<int>& kcore, std::vector<int>& order) {
  const int V = (int)g.off.size() - 1;
  const int n = V + 1;
  std::vector<int> pos((size_t)n), kc((size_t)n, 0), ko((size_t)n, 0);
  int md = 0;
  for (int v = 1; v < n; ++v) {
    kc[v] = (int)(g.off[v] - g.off[v - 1]);
    if (kc[v] > md) md = kc[v];
  }
  const int md_end = md + 1;
  std::vector<int> bin((size_t)md_end, 0);
  for (int v = 1; v < n; ++v) bin[kc[v]]++;
  int start = 1;
  for (int d = 0; d < md_end; ++d) {
    const int num = bin[d];
    bin[d] = start;
    start += num;
  }
  for (int v = 1; v < n; ++v) {
    pos[v] = bin[kc[v]];
    ko[pos[v]] = v;
    bin[kc[v]]++;
  }
  for (int d = md; d > 1; --d) bin[d] = bin[d - 1];
  bin[0] = 1;
  for (int i = 1; i < n; ++i) {
    const int v = ko[i];
    for (long long j = g.off[v - 1]; j < g.off[v]; ++j) {
      const int u = g.adj[j] + 1;
      if (kc[u] > kc[v]) {
        const int du = kc[u], pu = pos[u], pw = bin[du], w = ko[pw];
        if (u != w) {
          pos[u] = pw;
          ko[pu] = w;
          pos[w] = pu;
          ko[pw] = u;
        }
        bin[du]++;
        kc[u]--;
      }
    }
  }
  kcore.assign((size_t)V, 0);
  order.assign((size_t)V, 0);
  for (int v = 0; v < V; ++v) {
    kcore[v] = kc[v + 1] + 1;
    order[v] = ko[v + 1] - 1;
  }
  if (V == 0) return 0;
  return kcore[order[V - 1]] - 1;
}

struct HeuVertex {
  int id, bound;
};

void heu_branch(const Csr& g, const std::vector<int>& K, std::vector<HeuVertex>& P, int sz, int& mc, std::vector<int>& C,
                std::vector<short>& ind) {
  if (!P.empty()) {
    const int u = P.back().id;
    P.pop_back();
    for (long long j = g.off[u]; j < g.off[u + 1]; ++j) ind[g.adj[j]] = 1;
    std::vector<HeuVertex> R;
    R.reserve(P.size());
    for (size_t i = 0; i < P.size(); ++i)
      if (ind[P[i].id])
        if (K[P[i].id] > mc) R.push_back(P[i]);
    for (long long j = g.off[u]; j < g.off[u + 1]; ++j) ind[g.adj[j]] = 0;
    const int mc_prev = mc;
    heu_branch(g, K, R, sz + 1, mc, C, ind);
    if (mc > mc_prev) C.push_back(u);
  } else if (sz > mc)
    mc = sz;
}

// order_mode 0: canonical (K, id) ascending, traversed from the back; 1: BZ order from compute_cores.
int heu_search(const Csr& g, const std::vector<int>& K, const std::vector<int>& bz_order, int ub, int order_mode,
               std::vector<int>& C_max) {
  const int V = (int)g.off.size() - 1;
  std::vector<int> order;
  if (order_mode == 1)
    order = bz_order;
  else {
    order.resize((size_t)V);
    for (int v = 0; v < V; ++v) order[v] = v;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return K[a] < K[b]; });
  }
  std::vector<short> ind((size_t)V, 0);
  std::vector<int> C;
  std::vector<HeuVertex> P;
  int mc = 0;
  bool found_ub = false;
  C_max.clear();
  for (int i = V - 1; i >= 0; --i) {
    if (found_ub) continue;
    const int v = order[i];
    const int mc_prev = mc;
    int mc_cur = mc;
    if (K[v] > mc) {
      for (long long j = g.off[v]; j < g.off[v + 1]; ++j)
        if (K[g.adj[j]] > mc) P.push_back(HeuVertex{g.adj[j], K[g.adj[j]]});
      if ((int)P.size() > mc_cur) {
        // std::sort(incr_heur) is unstable in PMC; defined here as stable -> (K, id) ascending (D3)
        std::stable_sort(P.begin(), P.end(), [](const HeuVertex& a, const HeuVertex& b) { return a.bound < b.bound; });
        heu_branch(g, K, P, 1, mc_cur, C, ind);
        if (mc_cur > mc_prev) {
          if (mc < mc_cur) {
            mc = mc_cur;
            C.push_back(v);
            C_max = C;
            if (mc >= ub) found_ub = true;
          }
        }
      }
      C.clear();
      P.clear();
    }
  }
  return (int)C_max.size();
}

// Exact maximum clique ("next" row (f)4; reference src/graph.cc:106-127 hands the graph to PMC's branch-and-bound,
// pmcx_maxclique::search / search_dense: k-core pruning, neighbourhood-core ordering, greedy-colouring bounds, on
// OpenMP threads that race for the incumbent — which maximum clique comes back is not defined there).  Defined here
// (D10) so that a parallel search can reproduce it:
//   * H = the heuristic's clique (PMC_HEU semantics above), lb = |H|.  If lb = max_core + 1 the answer is H
//     (src/graph.cc:100-102).
//   * otherwise only vertices with core + 1 > lb can lie in a larger clique; they are searched in the canonical
//     (core, id) order: roots ascending, candidates of a root = its later neighbours, children descending.  The
//     incumbent changes only on a strictly larger clique, so the result is H when omega = lb, else the FIRST clique
//     of size omega in that depth-first order — whatever bounds are used for pruning.
// Bounds: |C| + |P| and a greedy sequential colouring of P (classes built from the highest-ranked vertex down).
struct ExactSearch {
  int n = 0, W = 0;
  std::vector<uint64_t> adj;  // compact graph, rank order
  int best = 0;
  std::vector<int> C, bestC;
  long long nodes = 0;
  static int count(const uint64_t* p, int W) {
    int c = 0;
    for (int w = 0; w < W; ++w) c += __builtin_popcountll(p[w]);
    return c;
  }
  static int highest(const uint64_t* p, int W) {
    for (int w = W - 1; w >= 0; --w)
      if (p[w]) return w * 64 + 63 - __builtin_clzll(p[w]);
    return -1;
  }
  // number of colour classes of a greedy colouring, stopped at limit + 1
  int colour_bound(const uint64_t* p, int limit) {
    std::vector<uint64_t> u(p, p + W), q((size_t)W);
    int colours = 0;
    while (highest(u.data(), W) >= 0) {
      if (++colours > limit) return limit + 1;
      q = u;
      int v;
      while ((v = highest(q.data(), W)) >= 0) {
        q[(size_t)(v >> 6)] &= ~(1ULL << (v & 63));
        u[(size_t)(v >> 6)] &= ~(1ULL << (v & 63));
        const uint64_t* row = &adj[(size_t)v * W];
        for (int w = 0; w < W; ++w) q[(size_t)w] &= ~row[w];
      }
    }
    return colours;
  }
  void expand(std::vector<uint64_t>& P) {
    std::vector<uint64_t> R((size_t)W);
    while (true) {
      ++nodes;
      const int cnt = count(P.data(), W), size = (int)C.size();
      if (cnt == 0) {
        return;
      }
      if (size + cnt <= best) return;
      if (size + colour_bound(P.data(), best - size) <= best) return;
      const int u = highest(P.data(), W);
      P[(size_t)(u >> 6)] &= ~(1ULL << (u & 63));
      const uint64_t* row = &adj[(size_t)u * W];
      bool any = false;
      for (int w = 0; w < W; ++w) {
        R[(size_t)w] = P[(size_t)w] & row[w];
        any |= R[(size_t)w] != 0;
      }
      C.push_back(u);
      if (!any) {
        if ((int)C.size() > best) {
          best = (int)C.size();
          bestC = C;
        }
      } else {
        std::vector<uint64_t> child(R);
        expand(child);
      }
      C.pop_back();
    }
  }
};

// returns true when a clique larger than |C_io| exists; C_io then holds it (original ids)
bool exact_improve(const uint64_t* bm, int L, const std::vector<int>& K /* core + 1 */, std::vector<int>& C_io,
                   long long* nodes_out) {
  const int lb = (int)C_io.size(), W0 = graph_words(L);
  std::vector<int> order((size_t)L);
  for (int v = 0; v < L; ++v) order[(size_t)v] = v;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return K[(size_t)a] < K[(size_t)b]; });
  std::vector<int> verts;  // candidates, canonical order
  for (int i = 0; i < L; ++i)
    if (K[(size_t)order[(size_t)i]] > lb) verts.push_back(order[(size_t)i]);
  ExactSearch S;
  S.n = (int)verts.size();
  S.W = graph_words(S.n > 0 ? S.n : 1);
  S.adj.assign((size_t)S.n * S.W, 0);
  std::vector<int> pos((size_t)L, -1);
  for (int i = 0; i < S.n; ++i) pos[(size_t)verts[(size_t)i]] = i;
  for (int i = 0; i < S.n; ++i) {
    const uint64_t* row = bm + (size_t)verts[(size_t)i] * W0;
    for (int w = 0; w < W0; ++w) {
      uint64_t x = row[w];
      while (x) {
        const int j = w * 64 + __builtin_ctzll(x);
        x &= x - 1;
        if (j < L && pos[(size_t)j] >= 0 && j != verts[(size_t)i])
          S.adj[(size_t)i * S.W + (size_t)(pos[(size_t)j] >> 6)] |= 1ULL << (pos[(size_t)j] & 63);
      }
    }
  }
  S.best = lb;
  for (int r = 0; r < S.n; ++r) {
    if (K[(size_t)verts[(size_t)r]] <= S.best) continue;  // core + 1 bounds every clique through r
    std::vector<uint64_t> P((size_t)S.W, 0);
    for (int w = r >> 6; w < S.W; ++w) {
      uint64_t x = S.adj[(size_t)r * S.W + (size_t)w];
      if (w == (r >> 6)) x &= (r & 63) == 63 ? 0ULL : ~((2ULL << (r & 63)) - 1ULL);
      P[(size_t)w] = x;
    }
    S.C.assign(1, r);
    if (ExactSearch::count(P.data(), S.W) == 0) continue;  // a single vertex never beats lb >= 1 ... (lb >= 1 whenever L >= 1)
    S.expand(P);
  }
  if (nodes_out) *nodes_out = S.nodes;
  if (S.best <= lb) return false;
  C_io.clear();
  for (int c : S.bestC) C_io.push_back(verts[(size_t)c]);
  return true;
}

// mode: 0 PMC_EXACT, 1 PMC_HEU, 2 KCORE_HEU.  Returns clique (unsorted, as PMC returns it).
int find_max_clique(const uint64_t* bm, int L, int mode, double kcore_thr, int order_mode, std::vector<int>& C,
                    int* max_core_out, std::vector<int>* core_out) {
  Csr g;
  bitmap_to_csr(bm, L, g);
  std::vector<int> K, order;
  const int max_core = compute_cores_bz(g, K, order);
  if (max_core_out) *max_core_out = max_core;
  if (core_out) {
    core_out->resize((size_t)L);
    for (int v = 0; v < L; ++v) (*core_out)[v] = K[v] - 1;
  }
  C.clear();
  if (mode == 2 && kcore_thr != 1 && max_core > (int)(kcore_thr * (double)L)) {
    // reference src/graph.cc:67-82 incl. its shifted indexing: k_cores has V+1 entries in PMC
    // (entry V is a stale leftover of the shift); entry i (1..V) is tested, vertex i-1 is pushed.
    std::vector<int> kc_shift((size_t)L + 1, 0);
    for (int v = 0; v < L; ++v) kc_shift[v] = K[v];
    kc_shift[L] = K[L - 1] - 1;  // PMC leaves kcore[n-1] at its pre-shift value = core(V-1)
    for (int i = 1; i <= L; ++i)
      if (kc_shift[i] >= max_core) C.push_back(i - 1);
    return (int)C.size();
  }
  const int ub = max_core + 1;
  heu_search(g, K, order, ub, order_mode, C);
  if (mode == 0 && (int)C.size() < ub) exact_improve(bm, L, K, C, nullptr);
  return (int)C.size();
}

// ================================================================================================
// K14  GNC-TLS 2-D rotation.  Follows Quatro::solveForRotation2D (reference include/quatro.hpp:
// 430-572) with svdRot2d (include/teaser/utils.h:151-166) in closed form (D6).
struct GncOut {
  double R[4];
  double cost;
  int iters;
};

double sum64_strided(const std::vector<double>& v) {
  double p[64];
  for (int l = 0; l < 64; ++l) p[l] = 0.0;
  for (size_t j = 0; j < v.size(); ++j) p[j & 63] = p[j & 63] + v[j];
  return qm_sum64_fold(p);
}

void rot2d_closed_form(double h00, double h01, double h10, double h11, double* R) {
  const double a = h00 + h11, b = h01 - h10;
  const double nrm = sqrt(a * a + b * b);
  double c = 1.0, s = 0.0;
  if (nrm > 0.0) {
    c = a / nrm;
    s = b / nrm;
  }
  R[0] = c;
  R[1] = -s;
  R[2] = s;
  R[3] = c;
}

void gnc_rotation2d(const double* src2, const double* dst2, int M, double noise_bound, double gnc_factor, int max_iter,
                    double cost_thr, GncOut& out, std::vector<char>& inliers) {
  double mu = 1, prev_cost = INFINITY;
  out.cost = INFINITY;
  out.iters = 0;
  double nb_sq = noise_bound * noise_bound;
  if (nb_sq < 1e-16) nb_sq = 1e-2;
  std::vector<double> w((size_t)M, 1.0), r2((size_t)M), t0((size_t)M), t1((size_t)M), t2((size_t)M), t3((size_t)M);
  out.R[0] = out.R[3] = 1;
  out.R[1] = out.R[2] = 0;
  for (int it = 0; it < max_iter; ++it) {
    out.iters = it + 1;
    for (int j = 0; j < M; ++j) {
      const double x0 = src2[2 * j], x1 = src2[2 * j + 1], y0 = dst2[2 * j], y1 = dst2[2 * j + 1];
      const double wx0 = w[j] * x0, wx1 = w[j] * x1;
      t0[j] = wx0 * y0;
      t1[j] = wx0 * y1;
      t2[j] = wx1 * y0;
      t3[j] = wx1 * y1;
    }
    rot2d_closed_form(sum64_strided(t0), sum64_strided(t1), sum64_strided(t2), sum64_strided(t3), out.R);
    double max_r = -INFINITY;
    for (int j = 0; j < M; ++j) {
      const double x0 = src2[2 * j], x1 = src2[2 * j + 1], y0 = dst2[2 * j], y1 = dst2[2 * j + 1];
      const double e0 = y0 - (out.R[0] * x0 + out.R[1] * x1), e1 = y1 - (out.R[2] * x0 + out.R[3] * x1);
      r2[j] = e0 * e0 + e1 * e1;
      if (r2[j] > max_r) max_r = r2[j];
    }
    if (it == 0) {
      mu = 1 / (2 * max_r / nb_sq - 1);
      if (mu <= 0) break;
    }
    const double th1 = (mu + 1) / mu * nb_sq, th2 = mu / (mu + 1) * nb_sq;
    for (int j = 0; j < M; ++j) t0[j] = w[j] * r2[j];
    out.cost = sum64_strided(t0);
    for (int j = 0; j < M; ++j) {
      if (r2[j] >= th1)
        w[j] = 0;
      else if (r2[j] <= th2)
        w[j] = 1;
      else
        w[j] = sqrt(nb_sq * mu * (mu + 1) / r2[j]) - mu;
    }
    const double cost_diff = fabs(out.cost - prev_cost);
    mu = mu * gnc_factor;
    prev_cost = out.cost;
    if (cost_diff < cost_thr) break;
  }
  inliers.resize((size_t)M);
  for (int j = 0; j < M; ++j) inliers[j] = w[j] >= 0.4;
}

// "Next" row (f)4, second half: the 3-DoF rotation the reference's API names but never reaches ("TEASER" reg_name:
// solveForRotation throws, include/quatro.hpp:409-411; teaser::utils::svdRot, include/teaser/utils.h:123-149, is the
// only piece present).  Restated as TEASER++'s GNC-TLS rotation solver — the loop solveForRotation2D (:430-572) was
// derived from — with 3-D residuals and the 3x3 weighted rotation of qtr_math.h (Horn quaternion form instead of
// JacobiSVD, fixed summation order: same kind of divergence as D6).  src3 / dst3: M x 3 row-major.
struct Gnc3Out {
  double R[9];
  double cost;
  int iters;
};
void gnc_rotation3d(const double* src3, const double* dst3, int M, double noise_bound, double gnc_factor, int max_iter,
                    double cost_thr, Gnc3Out& out, std::vector<char>& inliers) {
  double mu = 1, prev_cost = INFINITY;
  out.cost = INFINITY;
  out.iters = 0;
  double nb_sq = noise_bound * noise_bound;
  if (nb_sq < 1e-16) nb_sq = 1e-2;
  std::vector<double> w((size_t)M, 1.0), r2((size_t)M), t((size_t)M);
  for (int i = 0; i < 9; ++i) out.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int it = 0; it < max_iter; ++it) {
    out.iters = it + 1;
    double H[9];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        for (int j = 0; j < M; ++j) t[(size_t)j] = (w[(size_t)j] * src3[3 * j + a]) * dst3[3 * j + b];
        H[3 * a + b] = sum64_strided(t);
      }
    qm_rot3_from_h(H, out.R);
    const double* R = out.R;
    double max_r = -INFINITY;
    for (int j = 0; j < M; ++j) {
      const double x0 = src3[3 * j], x1 = src3[3 * j + 1], x2 = src3[3 * j + 2];
      const double e0 = dst3[3 * j] - ((R[0] * x0 + R[1] * x1) + R[2] * x2);
      const double e1 = dst3[3 * j + 1] - ((R[3] * x0 + R[4] * x1) + R[5] * x2);
      const double e2 = dst3[3 * j + 2] - ((R[6] * x0 + R[7] * x1) + R[8] * x2);
      r2[(size_t)j] = (e0 * e0 + e1 * e1) + e2 * e2;
      if (r2[(size_t)j] > max_r) max_r = r2[(size_t)j];
    }
    if (it == 0) {
      mu = 1 / (2 * max_r / nb_sq - 1);
      if (mu <= 0) break;
    }
    const double th1 = (mu + 1) / mu * nb_sq, th2 = mu / (mu + 1) * nb_sq;
    for (int j = 0; j < M; ++j) t[(size_t)j] = w[(size_t)j] * r2[(size_t)j];
    out.cost = sum64_strided(t);
    for (int j = 0; j < M; ++j) {
      if (r2[(size_t)j] >= th1)
        w[(size_t)j] = 0;
      else if (r2[(size_t)j] <= th2)
        w[(size_t)j] = 1;
      else
        w[(size_t)j] = sqrt(nb_sq * mu * (mu + 1) / r2[(size_t)j]) - mu;
    }
    const double cost_diff = fabs(out.cost - prev_cost);
    mu = mu * gnc_factor;
    prev_cost = out.cost;
    if (cost_diff < cost_thr) break;
  }
  inliers.resize((size_t)M);
  for (int j = 0; j < M; ++j) inliers[(size_t)j] = w[(size_t)j] >= 0.4;
}

// ================================================================================================
// K15  COTE (component-wise translation estimate).  Follows Quatro::estimate (reference
// include/quatro.hpp:618-747): adaptive-voting sweep over 2N interval endpoints.  Sort is stable by
// (value, insertion position) (D3); median of the last n_card sweep members (quirk kept), n_card<=1
// defined (D4).
// R: one range per element (estimate() takes a vector of ranges; the class itself only ever passes equal ones)
double cote_estimate_ranges(const std::vector<double>& X, const std::vector<double>& R, bool median_sel,
                            std::vector<char>& inl, int* n_card_out) {
  const int N = (int)X.size();
  struct Ev {
    double v;
    int id;  // +(i+1) lower, -(i+1) upper
    int pos;
  };
  std::vector<Ev> h((size_t)(2 * N));
  for (int i = 0; i < N; ++i) {
    h[2 * i] = Ev{X[i] - R[i], i + 1, 2 * i};
    h[2 * i + 1] = Ev{X[i] + R[i], -i - 1, 2 * i + 1};
  }
  // a NaN endpoint (NaN coordinate in, which the reference's std::sort leaves undefined) sorts as +inf, ties by position:
  // the order is total whatever comes in
  for (auto& e : h)
    if (e.v != e.v) e.v = INFINITY;
  std::sort(h.begin(), h.end(), [](const Ev& a, const Ev& b) { return a.v < b.v || (a.v == b.v && a.pos < b.pos); });
  const int nc = 2 * N;
  std::vector<double> x_hat((size_t)nc), x_cost((size_t)nc);
  std::vector<int> card((size_t)nc);
  double ranges_inverse_sum = 0;  // ranges.sum(): sequential add (Eigen's packet order for unequal ranges is not restated)
  for (int i = 0; i < N; ++i) ranges_inverse_sum += R[i];
  double dot_X_weights = 0, dot_weights_consensus = 0, sum_xi = 0, sum_xi_square = 0;
  int consensus = 0;
  for (int i = 0; i < nc; ++i) {
    const int idx = std::abs(h[i].id) - 1;
    const int eps = h[i].id > 0 ? 1 : -1;
    const double range = R[idx];
    const double weight = 1.0 / (range * range);  // weights = ranges.square().inverse()
    consensus += eps;
    dot_weights_consensus += eps * weight;
    dot_X_weights += eps * weight * X[idx];
    ranges_inverse_sum -= eps * range;
    sum_xi += eps * X[idx];
    sum_xi_square += eps * X[idx] * X[idx];
    card[i] = consensus;
    x_hat[i] = dot_X_weights / dot_weights_consensus;
    const double residual = consensus * x_hat[i] * x_hat[i] + sum_xi_square - 2 * sum_xi * x_hat[i];
    x_cost[i] = residual + ranges_inverse_sum;
  }
  int min_idx = 0;  // Eigen minCoeff(&idx): first strict minimum, NaN never selected unless first
  for (int i = 1; i < nc; ++i)
    if (x_cost[i] < x_cost[min_idx]) min_idx = i;
  double est = x_hat[min_idx];
  const int n_card = card[min_idx];
  if (n_card_out) *n_card_out = n_card;
  if (median_sel) {
    if (n_card >= 2) {
      std::vector<double> cand;
      for (int j = 0; j < n_card; ++j) cand.push_back(X[std::abs(h[min_idx - j].id) - 1]);
      std::sort(cand.begin(), cand.end());
      est = (cand[cand.size() / 2 - 1] + cand[cand.size() / 2]) / 2.0;
    } else if (n_card == 1) {
      est = X[std::abs(h[min_idx].id) - 1];
    }
  }
  inl.resize((size_t)N);
  for (int i = 0; i < N; ++i) inl[i] = fabs(X[i] - est) <= R[i];
  return est;
}
double cote_estimate(const std::vector<double>& X, double range, bool median_sel, std::vector<char>& inl,
                     int* n_card_out) {
  return cote_estimate_ranges(X, std::vector<double>(X.size(), range), median_sel, inl, n_card_out);
}

}  // namespace

// ================================================================================================
// C API (ctypes-friendly)
// =================================================================================================
// "Next" row (f)2 of SURVEY.md section 8: Patchwork ground segmentation (reference include/patchwork.hpp:
// estimate_ground :329-476, pc2czm :512-546, extract_initial_seeds_ :285-318, estimate_plane_ :271-283,
// extract_piecewiseground :549-586; parameters config/patchwork_params.yaml).  Restated with these declared
// choices where the reference leaves the result to its libraries:
//   * the z sort is stable (std::sort in the reference: order of equal heights unspecified),
//   * the per-patch moment sums (pcl::computeMeanAndCovarianceMatrix, float) are evaluated in the fixed sum64
//     order of qtr_math.h instead of sequentially,
//   * the plane normal is the smallest eigenvector from the closed-form pcl::eigen33 restatement used for the
//     normals (instead of Eigen::JacobiSVD), oriented so that n_z >= 0 (the SVD's sign is an artefact; for
//     ground-like patches it points up), and the singular values are the |eigenvalues| of the same solve,
//   * atan2 / sqrt of the polar binning in binary64 through qm_atan2d.
struct PwParams {
  double sensor_height;
  int num_iter, num_lpr, num_min_pts;
  double th_seeds, th_dist, max_range, min_range, uprightness_thr, adaptive_seed_selection_margin;
  int using_global_thr;
  double global_elevation_thr;
  int num_zones;
  int num_sectors_each_zone[4], num_rings_each_zone[4];
  double min_ranges[4];
  int num_thr;  // num_rings_of_interest = size of the two threshold vectors
  double elevation_thr[8], flatness_thr[8];
};

struct PwPlane {
  float normal[3];
  float mean[3];
  float sv[3];      // descending |eigenvalues|
  float th_dist_d;
};


static void pw_estimate_plane(const float* xyz4, const int* ids, int n, const std::vector<char>& in_ground,
                              double th_dist, PwPlane& pl) {
  // nine moment sums over the ground set, sum256 order: the point at patch position t goes to partial[t & 255];
  // the 256 partials fold as four sum64 folds combined (s0 + s1) + (s2 + s3)
  float part[9][256];
  for (int a = 0; a < 9; ++a)
    for (int l = 0; l < 256; ++l) part[a][l] = 0.f;
  int cnt = 0;
  for (int t = 0; t < n; ++t) {
    if (!in_ground[(size_t)t]) continue;
    const float* q = xyz4 + 4 * (size_t)ids[t];
    const int l = t & 255;
    part[0][l] += q[0] * q[0];
    part[1][l] += q[0] * q[1];
    part[2][l] += q[0] * q[2];
    part[3][l] += q[1] * q[1];
    part[4][l] += q[1] * q[2];
    part[5][l] += q[2] * q[2];
    part[6][l] += q[0];
    part[7][l] += q[1];
    part[8][l] += q[2];
    ++cnt;
  }
  float acc[9];
  for (int a = 0; a < 9; ++a) {
    const float s0 = qm_sum64_fold_f(part[a]), s1 = qm_sum64_fold_f(part[a] + 64);
    const float s2 = qm_sum64_fold_f(part[a] + 128), s3 = qm_sum64_fold_f(part[a] + 192);
    acc[a] = (s0 + s1) + (s2 + s3);
  }
  const float kk = (float)cnt;
  for (int a = 0; a < 9; ++a) acc[a] /= kk;
  float cov[9];
  cov[0] = acc[0] - acc[6] * acc[6];
  cov[1] = acc[1] - acc[6] * acc[7];
  cov[2] = acc[2] - acc[6] * acc[8];
  cov[4] = acc[3] - acc[7] * acc[7];
  cov[5] = acc[4] - acc[7] * acc[8];
  cov[8] = acc[5] - acc[8] * acc[8];
  cov[3] = cov[1];
  cov[6] = cov[2];
  cov[7] = cov[5];
  float ev, vec[3];
  eigen33_smallest(cov, &ev, vec);
  // all three eigenvalues of the scaled matrix, as eigen33_smallest computes them
  float scale = 0.f;
  for (int i = 0; i < 9; ++i) scale = std::max(scale, fabsf(cov[i]));
  if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
  float sm[9], roots[3];
  for (int i = 0; i < 9; ++i) sm[i] = cov[i] / scale;
  compute_roots(sm, roots);
  float a0 = fabsf(roots[0] * scale), a1 = fabsf(roots[1] * scale), a2 = fabsf(roots[2] * scale);
  if (a0 < a1) std::swap(a0, a1);
  if (a1 < a2) std::swap(a1, a2);
  if (a0 < a1) std::swap(a0, a1);
  pl.sv[0] = a0;
  pl.sv[1] = a1;
  pl.sv[2] = a2;
  const bool flip = vec[2] < 0.f || (vec[2] == 0.f && (vec[1] < 0.f || (vec[1] == 0.f && vec[0] < 0.f)));
  for (int i = 0; i < 3; ++i) pl.normal[i] = flip ? -vec[i] : vec[i];
  pl.mean[0] = acc[6];
  pl.mean[1] = acc[7];
  pl.mean[2] = acc[8];
  const float d = -((pl.normal[0] * pl.mean[0] + pl.normal[1] * pl.mean[1]) + pl.normal[2] * pl.mean[2]);
  pl.th_dist_d = (float)(th_dist - (double)d);
}

// ground4 / nonground4: x,y,z,w of the input points in the reference's output order; returns counts
static void patchwork(const float* xyz4, int P, const PwParams& pw, float* ground4, int* n_ground, float* nonground4,
                      int* n_nonground, int* patch_of /* optional [P]: patch id or -1 */) {
  std::vector<int> order((size_t)P);
  for (int i = 0; i < P; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return xyz4[4 * (size_t)a + 2] < xyz4[4 * (size_t)b + 2]; });
  int base[5] = {0, 0, 0, 0, 0};
  for (int k = 0; k < pw.num_zones; ++k) base[k + 1] = base[k] + pw.num_rings_each_zone[k] * pw.num_sectors_each_zone[k];
  const int npatch = base[pw.num_zones];
  double ring_size[4], sector_size[4];
  for (int k = 0; k < pw.num_zones; ++k) {
    const double hi = (k + 1 < pw.num_zones) ? pw.min_ranges[k + 1] : pw.max_range;
    ring_size[k] = (hi - pw.min_ranges[k]) / pw.num_rings_each_zone[k];
    sector_size[k] = 2 * M_PI / pw.num_sectors_each_zone[k];
  }
  std::vector<std::vector<int>> patch((size_t)npatch);
  if (patch_of) std::fill(patch_of, patch_of + P, -1);
  for (int t = 0; t < P; ++t) {
    const int i = order[t];
    const float* q = xyz4 + 4 * (size_t)i;
    if ((double)q[2] < -1.8 * pw.sensor_height) continue;  // mirror reflections under the ground (:351-361)
    const double x = q[0], y = q[1];
    const double r = sqrt(x * x + y * y);
    if (!((r <= pw.max_range) && (r > pw.min_range))) continue;
    const double at = qm_atan2d(y, x);
    const double theta = at > 0 ? at : at + 2 * M_PI;
    int k = pw.num_zones - 1;
    for (int z = 1; z < pw.num_zones; ++z)
      if (r < pw.min_ranges[z]) {
        k = z - 1;
        break;
      }
    const int ring = std::min((int)((r - pw.min_ranges[k]) / ring_size[k]), pw.num_rings_each_zone[k] - 1);
    const int sector = std::min((int)(theta / sector_size[k]), pw.num_sectors_each_zone[k] - 1);
    const int pid = base[k] + ring * pw.num_sectors_each_zone[k] + sector;
    patch[(size_t)pid].push_back(i);
    if (patch_of) patch_of[i] = pid;
  }
  int ng = 0, nn = 0;
  auto emit = [&](float* dst, int& cnt, int id) {
    const float* q = xyz4 + 4 * (size_t)id;
    dst[4 * (size_t)cnt] = q[0];
    dst[4 * (size_t)cnt + 1] = q[1];
    dst[4 * (size_t)cnt + 2] = q[2];
    dst[4 * (size_t)cnt + 3] = q[3];
    ++cnt;
  };
  const double margin = (pw.sensor_height == 0.0) ? -0.1 : pw.adaptive_seed_selection_margin * pw.sensor_height;
  int concentric_idx = 0;
  for (int k = 0; k < pw.num_zones; ++k)
    for (int ring = 0; ring < pw.num_rings_each_zone[k]; ++ring) {
      for (int sector = 0; sector < pw.num_sectors_each_zone[k]; ++sector) {
        const std::vector<int>& ids = patch[(size_t)(base[k] + ring * pw.num_sectors_each_zone[k] + sector)];
        const int n = (int)ids.size();
        if (!(n > pw.num_min_pts)) continue;
        // seeds (:285-318)
        int init_idx = 0;
        if (k == 0)
          while (init_idx < n && (double)xyz4[4 * (size_t)ids[init_idx] + 2] < margin) ++init_idx;
        double sum = 0;
        int cnt = 0;
        for (int t = init_idx; t < n && cnt < pw.num_lpr; ++t) {
          sum += (double)xyz4[4 * (size_t)ids[t] + 2];
          ++cnt;
        }
        const double lpr_height = cnt != 0 ? sum / cnt : 0;
        std::vector<char> in_ground((size_t)n, 0), is_ground((size_t)n, 0);
        for (int t = 0; t < n; ++t) in_ground[t] = ((double)xyz4[4 * (size_t)ids[t] + 2] < lpr_height + pw.th_seeds) ? 1 : 0;
        PwPlane pl;
        for (int it = 0; it < pw.num_iter; ++it) {
          pw_estimate_plane(xyz4, ids.data(), n, in_ground, pw.th_dist, pl);
          for (int t = 0; t < n; ++t) {
            const float* q = xyz4 + 4 * (size_t)ids[t];
            const float res = (q[0] * pl.normal[0] + q[1] * pl.normal[1]) + q[2] * pl.normal[2];
            const char g = res < pl.th_dist_d ? 1 : 0;
            if (it < pw.num_iter - 1)
              in_ground[t] = g;
            else
              is_ground[t] = g;
          }
        }
        const double ground_z_vec = fabs((double)pl.normal[2]);
        const double ground_z_elevation = pl.mean[2];
        const double surface_variable = (double)pl.sv[2] / (double)((pl.sv[0] + pl.sv[1]) + pl.sv[2]);
        bool reject_all = false;
        if (ground_z_vec < pw.uprightness_thr)
          reject_all = true;
        else if (concentric_idx < pw.num_thr) {
          const int ti = ring + 2 * k;  // the reference's index (sic), :395
          if (ground_z_elevation > pw.elevation_thr[ti] && !(pw.flatness_thr[ti] > surface_variable)) reject_all = true;
        } else if (pw.using_global_thr && ground_z_elevation > pw.global_elevation_thr)
          reject_all = true;
        for (int t = 0; t < n; ++t)
          if (is_ground[t]) {
            if (reject_all)
              emit(nonground4, nn, ids[t]);
            else
              emit(ground4, ng, ids[t]);
          }
        for (int t = 0; t < n; ++t)
          if (!is_ground[t]) emit(nonground4, nn, ids[t]);
      }
      ++concentric_idx;
    }
  *n_ground = ng;
  *n_nonground = nn;
}

// =================================================================================================
// "Next" row (f)1 of SURVEY.md section 8: range-image projection + sub-cluster rejection
// (reference include/imageProjection.hpp: projectPointCloud :308-352, maskGround :354-364 in "Patchwork" mode,
// cloudSegmentation :424-483, labelComponents :485-581; LeGO-LOAM lineage).  Restated as: last-writer-wins
// projection, connected components of the symmetric angle criterion (what the BFS computes), the BFS's validity
// rule (size >= num_min_pts, or size >= 5 and >= 3 rows holding a pixel other than the seed), labels numbered in
// row-major order of the components' first pixels.  Declared divergences: atan2f / sin / cos come from qtr_math.h
// (D7); the float -> size_t conversions of negative values (undefined behaviour in the reference) are defined as
// truncation toward zero followed by the range check, which is what x86-64 does.
struct IpParams {
  int n_scan, horizon_scan;
  float ang_res_x, ang_res_y, ang_bottom;
  int neighbor_mode;  // 0: 4-neighbour, 1: 8-neighbour, 2: 4-cross-neighbour
  int num_min_pts;    // numMinPtsForSubclustering (30)
  float segment_theta;  // 60 deg in rad
  int valid_point_num, valid_line_num;  // 5, 3
};

static bool ip_pixel_of(const IpParams& ip, float x, float y, float z, int* row, int* col, float* range) {
  const float va = (float)((double)(qm_atan2f(z, sqrtf(x * x + y * y)) * 180) / M_PI);
  const float rf = (va + ip.ang_bottom) / ip.ang_res_y;
  const long long r = (long long)rf;  // size_t conversion in the reference; negative values wrap and fail the test
  if (r < 0 || r >= ip.n_scan) return false;
  const float ha = (float)((double)(qm_atan2f(x, y) * 180) / M_PI);
  const double cd = -round(((double)ha - 90.0) / (double)ip.ang_res_x) + (double)(ip.horizon_scan / 2);
  long long c = (long long)cd;
  if (c < 0) return false;
  if (c >= ip.horizon_scan) c -= ip.horizon_scan;
  if (c < 0 || c >= ip.horizon_scan) return false;
  const float rg = sqrtf(x * x + y * y + z * z);
  if (rg < 0.1) return false;
  *row = (int)r;
  *col = (int)c;
  *range = rg;
  return true;
}

static void ip_alpha_trig(const IpParams& ip, float* sx, float* cx, float* sy, float* cy) {
  const float ax = (float)((double)ip.ang_res_x / 180.0 * M_PI), ay = (float)((double)ip.ang_res_y / 180.0 * M_PI);
  qm_sincosf(ax, sx, cx);
  qm_sincosf(ay, sy, cy);
}

static inline bool ip_edge(float ra, float rb, float sn, float cs, float theta) {
  const float d1 = ra > rb ? ra : rb, d2 = ra > rb ? rb : ra;
  const float angle = qm_atan2f(d2 * sn, (d1 - d2 * cs));
  return angle > theta;
}

static int ip_neighbors(int mode, int (*off)[2]) {
  static const int n4[4][2] = {{-1, 0}, {0, 1}, {0, -1}, {1, 0}};
  static const int n8[8][2] = {{-1, 0}, {0, 1}, {0, -1}, {1, 0}, {-1, -1}, {-1, 1}, {1, 1}, {1, -1}};
  static const int nx[4][2] = {{-1, -1}, {-1, 1}, {1, 1}, {1, -1}};
  const int cnt = mode == 1 ? 8 : 4;
  for (int i = 0; i < cnt; ++i) {
    const int* q = mode == 0 ? n4[i] : mode == 1 ? n8[i] : nx[i];
    off[i][0] = q[0];
    off[i][1] = q[1];
  }
  return cnt;
}

// labelmat: 0 never (every pixel ends up -1, 999999 or a label >= 1); out4: x,y,z,label; outl4: x,y,z,row+col/1e4
static void segment_cloud(const float* xyz4, int P, const IpParams& ip, float* out4, int* n_valid, float* outl4,
                          int* n_outl, int* labelmat, float* rangemat) {
  const int H = ip.horizon_scan, NS = ip.n_scan, NP = NS * H;
  std::vector<int> owner((size_t)NP, -1);
  std::vector<float> range((size_t)NP, FLT_MAX);
  for (int i = 0; i < P; ++i) {
    int r, c;
    float rg;
    if (!ip_pixel_of(ip, xyz4[4 * i], xyz4[4 * i + 1], xyz4[4 * i + 2], &r, &c, &rg)) continue;
    owner[(size_t)r * H + c] = i;  // later points overwrite earlier ones
    range[(size_t)r * H + c] = rg;
  }
  float sx, cx, sy, cy;
  ip_alpha_trig(ip, &sx, &cx, &sy, &cy);
  int off[8][2];
  const int nn = ip_neighbors(ip.neighbor_mode, off);
  std::vector<int> label((size_t)NP, 0);
  for (int p = 0; p < NP; ++p)
    if (owner[p] < 0) label[p] = -1;
  int label_count = 1;
  std::vector<int> queue((size_t)NP), pushed((size_t)NP);
  std::vector<char> line((size_t)NS);
  for (int r0 = 0; r0 < NS; ++r0)
    for (int c0 = 0; c0 < H; ++c0) {
      if (label[(size_t)r0 * H + c0] != 0) continue;
      // breadth-first labelling exactly as labelComponents does it
      std::fill(line.begin(), line.end(), 0);
      int qs = 0, qe = 1, np = 1;
      queue[0] = r0 * H + c0;
      pushed[0] = queue[0];
      while (qs < qe) {
        const int from = queue[qs++];
        const int fr = from / H, fc = from % H;
        label[from] = label_count;
        for (int q = 0; q < nn; ++q) {
          const int tr = fr + off[q][0];
          int tc = fc + off[q][1];
          if (tr < 0 || tr >= NS) continue;
          if (tc < 0) tc = H - 1;
          if (tc >= H) tc = 0;
          const int to = tr * H + tc;
          if (label[to] != 0) continue;
          const bool horiz = off[q][0] == 0;
          if (ip_edge(range[from], range[to], horiz ? sx : sy, horiz ? cx : cy, ip.segment_theta)) {
            queue[qe++] = to;
            label[to] = label_count;
            line[tr] = 1;
            pushed[np++] = to;
          }
        }
      }
      bool feasible = false;
      if (np >= ip.num_min_pts)
        feasible = true;
      else if (np >= ip.valid_point_num) {
        int lc = 0;
        for (int r = 0; r < NS; ++r) lc += line[r];
        if (lc >= ip.valid_line_num) feasible = true;
      }
      if (feasible)
        ++label_count;
      else
        for (int i = 0; i < np; ++i) label[pushed[i]] = 999999;
    }
  int nv = 0, no = 0;
  for (int p = 0; p < NP; ++p) {
    if (label[p] > 0 && label[p] != 999999) {
      const float* q = xyz4 + 4 * (size_t)owner[p];
      out4[4 * nv] = q[0];
      out4[4 * nv + 1] = q[1];
      out4[4 * nv + 2] = q[2];
      out4[4 * nv + 3] = (float)label[p];
      ++nv;
    } else if (label[p] == 999999) {
      const float* q = xyz4 + 4 * (size_t)owner[p];
      outl4[4 * no] = q[0];
      outl4[4 * no + 1] = q[1];
      outl4[4 * no + 2] = q[2];
      outl4[4 * no + 3] = (float)(p / H) + (float)(p % H) / 10000.0f;
      ++no;
    }
  }
  *n_valid = nv;
  *n_outl = no;
  if (labelmat) std::copy(label.begin(), label.end(), labelmat);
  if (rangemat) std::copy(range.begin(), range.end(), rangemat);
}


extern "C" {

struct qo_params {
  double noise_bound;         // Params::noise_bound (0.3)
  double cbar2;               // Params::cbar2 (1.0)
  double rotation_gnc_factor; // 1.4
  double rotation_cost_threshold;
  double kcore_heuristic_threshold;
  double cote_noise_bound;    // Quatro::noise_bound_ member (0.3), reference quatro.hpp:115,601
  double ryrx[9];             // estimated_RyRx_ (row-major)
  int rotation_max_iterations;
  int inlier_selection_mode;  // 0 PMC_EXACT, 1 PMC_HEU, 2 KCORE_HEU, 3 NONE
  int cote_median;            // 1 = "median", 0 = "weighted_mean"
  int using_rot_inliers_when_estimating_cote;
  int using_pre_estimated_ryrx;
  int clique_order;           // 0 canonical, 1 bz  (oracle-only knob, D2)
  int reg_mode;               // 0 = reg_name "Quatro" (yaw), 1 = "TEASER" (3-DoF rotation, row (f)4)
};

struct qo_result {
  int status;  // 0 ok, 1 clique too small (solution invalid), 2 unsupported mode
  int valid;
  double T[16];  // row-major 4x4
  double cost;
  int gnc_iters;
  int n_clique;
  int n_rot_inliers;
  int n_final;
  int max_core;
  int n_edges;  // undirected edge count of the consistency graph
  int n_card[3];
};

void qo_set_threads(int t) { g_threads = t < 1 ? 1 : t; }

// host evaluation of include/qtr_math.h (pins host/device bit-equality in tests):
// fn 0 atan2f(a,b), 1 acosf(a), 2 sinf(a), 3 cosf(a); 4 the SPFH bin of atan2f(a,b); 5 the role-swap decision
// acosf(|a|) > acosf(|b|) exactly as the pair-feature code evaluates it (the device has a shortcut for it)
void qo_math(int fn, const float* a, const float* b, float* out, int n) {
  const float d_pi = 1.0f / (2.0f * (float)M_PI);
  for (int i = 0; i < n; ++i) {
    float s, c;
    switch (fn) {
      case 0: out[i] = qm_atan2f(a[i], b[i]); break;
      case 1: out[i] = qm_acosf(a[i]); break;
      case 2: qm_sincosf(a[i], &s, &c); out[i] = s; break;
      case 3: qm_sincosf(a[i], &s, &c); out[i] = c; break;
      case 4: out[i] = (float)bin11(11 * (((double)qm_atan2f(a[i], b[i]) + M_PI) * (double)d_pi)); break;
      default: out[i] = (qm_acosf(fabsf(a[i])) > qm_acosf(fabsf(b[i]))) ? 1.f : 0.f; break;
    }
  }
}
unsigned int qo_rand_u32(unsigned long long seed, unsigned long long counter) { return qm_rand_u32(seed, counter); }
int qo_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

struct qo_ip_params {
  int n_scan, horizon_scan;
  float ang_res_x, ang_res_y, ang_bottom;
  int neighbor_mode, num_min_pts;
  float segment_theta;
  int valid_point_num, valid_line_num;
};
// ImageProjection::segmentCloud in "Patchwork" mode + getValidSegments / getOutliers (capacity of the outputs:
// n_scan * horizon_scan points each)
int qo_segment_cloud(const float* xyz4, int P, const qo_ip_params* ipp, float* out4, int* n_valid, float* outl4,
                     int* n_outl, int* labelmat, float* rangemat) {
  IpParams ip;
  ip.n_scan = ipp->n_scan;
  ip.horizon_scan = ipp->horizon_scan;
  ip.ang_res_x = ipp->ang_res_x;
  ip.ang_res_y = ipp->ang_res_y;
  ip.ang_bottom = ipp->ang_bottom;
  ip.neighbor_mode = ipp->neighbor_mode;
  ip.num_min_pts = ipp->num_min_pts;
  ip.segment_theta = ipp->segment_theta;
  ip.valid_point_num = ipp->valid_point_num;
  ip.valid_line_num = ipp->valid_line_num;
  segment_cloud(xyz4, P, ip, out4, n_valid, outl4, n_outl, labelmat, rangemat);
  return 0;
}

struct qo_pw_params {
  double sensor_height;
  int num_iter, num_lpr, num_min_pts;
  double th_seeds, th_dist, max_range, min_range, uprightness_thr, adaptive_seed_selection_margin;
  int using_global_thr;
  double global_elevation_thr;
  int num_zones;
  int num_sectors_each_zone[4], num_rings_each_zone[4];
  double min_ranges[4];
  int num_thr;
  double elevation_thr[8], flatness_thr[8];
};
// PatchWork::estimate_ground: ground4 / nonground4 have capacity P points each
int qo_patchwork(const float* xyz4, int P, const qo_pw_params* pp, float* ground4, int* n_ground, float* nonground4,
                 int* n_nonground, int* patch_of) {
  static_assert(sizeof(qo_pw_params) == sizeof(PwParams), "same layout");
  PwParams pw;
  memcpy(&pw, pp, sizeof(pw));
  patchwork(xyz4, P, pw, ground4, n_ground, nonground4, n_nonground, patch_of);
  return 0;
}

int qo_voxelize(const float* xyz4, int P, float leaf, float* out4, int cap) { return voxelize(xyz4, P, leaf, out4, cap); }

// neighbour lists for tests: returns total count; off has n+1 entries; idx/d2 filled up to cap
long long qo_radius_neighbors(const float* xyz4, int n, double radius, long long* off, int* idx, float* d2,
                              long long cap) {
  Neighbors nb;
  radius_neighbors(xyz4, n, radius, nb);
  for (int i = 0; i <= n; ++i) off[i] = nb.off[i];
  const long long tot = nb.off[n];
  for (long long t = 0; t < tot && t < cap; ++t) {
    idx[t] = nb.idx[t];
    d2[t] = nb.d2[t];
  }
  return tot;
}

// reference src/teaser_utils/fpfh.cc:44-75 (4-argument overload used by FPFHManager)
int qo_fpfh(const float* xyz4, int n, double r_normal, double r_fpfh, float* normals4, float* spfh33, float* desc33) {
  if (n <= 0) return 0;
  Neighbors nb;
  radius_neighbors(xyz4, n, r_fpfh, nb);
  normals_from_neighbors(xyz4, n, nb, (float)(r_normal * r_normal), normals4);
  std::vector<float> spfh_local;
  float* sp = spfh33;
  if (!sp) {
    spfh_local.resize(33 * (size_t)n);
    sp = spfh_local.data();
  }
  spfh_from_neighbors(xyz4, normals4, n, nb, sp);
  fpfh_from_spfh(sp, n, nb, desc33);
  return 0;
}

// pcl::eigen33 smallest eigenpair of n symmetric 3x3 matrices (row-major 9 floats each) — exported so that the
// restatement can be checked against an independent eigen-solver on many random covariances
void qo_eigen33(const float* cov9, int n, float* evals, float* evecs3) {
  for (int i = 0; i < n; ++i) eigen33_smallest(cov9 + 9 * (size_t)i, evals + i, evecs3 + 3 * (size_t)i);
}

int qo_nn33(const float* query, int nq, const float* data, int n, int* out) {
#pragma omp parallel for schedule(dynamic, 16) num_threads(g_threads)
  for (int q = 0; q < nq; ++q) out[q] = nn33(query + 33 * (size_t)q, data, n);
  return 0;
}

int qo_match(const float* xyz_s, int ns, const float* desc_s, const float* xyz_t, int nt, const float* desc_t,
             int use_crosscheck, int use_tuple, float tuple_scale, unsigned long long seed, int* corr, int cap,
             int* dbg_nn_i_of_j, int* dbg_nn_j_of_i) {
  return match(xyz_s, ns, desc_s, xyz_t, nt, desc_t, use_crosscheck, use_tuple, tuple_scale, seed, corr, cap,
               dbg_nn_i_of_j, dbg_nn_j_of_i);
}

int qo_graph_words(int L) { return graph_words(L); }

// src4/tgt4: float xyz(+pad) as pcl::PointXYZ; bitmap: L * words uint64
int qo_build_graph(const float* src4, const float* tgt4, int L, double noise_bound, double cbar2,
                   unsigned long long* bitmap) {
  std::vector<double> s(3 * (size_t)L), t(3 * (size_t)L);
  for (int i = 0; i < L; ++i)
    for (int a = 0; a < 3; ++a) {
      s[3 * i + a] = (double)src4[4 * i + a];
      t[3 * i + a] = (double)tgt4[4 * i + a];
    }
  build_graph(s.data(), t.data(), L, noise_bound, cbar2, (uint64_t*)bitmap);
  return 0;
}

// core numbers (not shifted) + BZ order; returns max core
int qo_kcore(const unsigned long long* bitmap, int L, int* core, int* bz_order) {
  Csr g;
  bitmap_to_csr((const uint64_t*)bitmap, L, g);
  std::vector<int> K, order;
  const int mc = compute_cores_bz(g, K, order);
  for (int v = 0; v < L; ++v) {
    if (core) core[v] = K[v] - 1;
    if (bz_order) bz_order[v] = order[v];
  }
  return mc;
}

// returns clique size; clique[] sorted ascending
int qo_max_clique(const unsigned long long* bitmap, int L, int mode, double kcore_thr, int order_mode, int* clique) {
  std::vector<int> C;
  find_max_clique((const uint64_t*)bitmap, L, mode, kcore_thr, order_mode, C, nullptr, nullptr);
  std::sort(C.begin(), C.end());
  for (size_t i = 0; i < C.size(); ++i) clique[i] = C[i];
  return (int)C.size();
}

int qo_rot3_from_h(const double* H9, double* R9) {
  qm_rot3_from_h(H9, R9);
  return 0;
}

int qo_gnc_rotation3d(const double* src3, const double* dst3, int M, double noise_bound, double gnc_factor, int max_iter,
                      double cost_thr, double* R9, double* cost, int* iters, unsigned char* inliers) {
  Gnc3Out o;
  std::vector<char> inl;
  gnc_rotation3d(src3, dst3, M, noise_bound, gnc_factor, max_iter, cost_thr, o, inl);
  for (int i = 0; i < 9; ++i) R9[i] = o.R[i];
  *cost = o.cost;
  *iters = o.iters;
  for (int j = 0; j < M; ++j) inliers[j] = (unsigned char)inl[j];
  return 0;
}

int qo_gnc_rotation2d(const double* src2, const double* dst2, int M, double noise_bound, double gnc_factor, int max_iter,
                      double cost_thr, double* R4, double* cost, int* iters, unsigned char* inliers) {
  GncOut o;
  std::vector<char> inl;
  gnc_rotation2d(src2, dst2, M, noise_bound, gnc_factor, max_iter, cost_thr, o, inl);
  for (int i = 0; i < 4; ++i) R4[i] = o.R[i];
  *cost = o.cost;
  *iters = o.iters;
  for (int j = 0; j < M; ++j) inliers[j] = (unsigned char)inl[j];
  return 0;
}

double qo_cote_estimate(const double* X, int N, double range, int median_sel, unsigned char* inliers, int* n_card) {
  std::vector<double> x(X, X + N);
  std::vector<char> inl;
  const double e = cote_estimate(x, range, median_sel != 0, inl, n_card);
  for (int i = 0; i < N; ++i) inliers[i] = (unsigned char)inl[i];
  return e;
}

double qo_cote_estimate_ranges(const double* X, const double* R, int N, int median_sel, unsigned char* inliers,
                               int* n_card) {
  std::vector<double> x(X, X + N), r(R, R + N);
  std::vector<char> inl;
  const double e = cote_estimate_ranges(x, r, median_sel != 0, inl, n_card);
  for (int i = 0; i < N; ++i) inliers[i] = (unsigned char)inl[i];
  return e;
}

// Back end: Quatro::computeTransformation(Eigen::Matrix4d&) (reference include/quatro.hpp:769-936).
// clique / rot_inliers / final_inliers: caller buffers of capacity L.
int qo_solve(const float* src4, const float* tgt4, int L, const qo_params* prm, qo_result* res, int* clique,
             int* rot_inliers, int* final_inliers) {
  memset(res, 0, sizeof(*res));
  for (int i = 0; i < 4; ++i) res->T[5 * i] = 1.0;
  res->cost = INFINITY;
  if (prm->inlier_selection_mode == 3) {
    // NONE leaves max_clique_ empty in the reference (chain TIMs over an empty clique, then UB): outside this
    // restatement.
    res->status = 2;
    return 2;
  }
  std::vector<double> s(3 * (size_t)L), t(3 * (size_t)L);
  for (int i = 0; i < L; ++i)
    for (int a = 0; a < 3; ++a) {
      s[3 * i + a] = (double)src4[4 * i + a];
      t[3 * i + a] = (double)tgt4[4 * i + a];
    }
  const int W = graph_words(L);
  std::vector<uint64_t> bm((size_t)W * (size_t)L);
  build_graph(s.data(), t.data(), L, prm->noise_bound, prm->cbar2, bm.data());
  long long deg_sum = 0;
  for (uint64_t x : bm) deg_sum += __builtin_popcountll(x);
  res->n_edges = (int)(deg_sum / 2);
  std::vector<int> C;
  find_max_clique(bm.data(), L, prm->inlier_selection_mode, prm->kcore_heuristic_threshold, prm->clique_order, C,
                  &res->max_core, nullptr);
  std::sort(C.begin(), C.end());
  const int M = (int)C.size();
  res->n_clique = M;
  for (int i = 0; i < M; ++i) clique[i] = C[i];
  if (M <= 1) {
    res->valid = 0;
    res->status = 1;
    return 1;
  }
  // chain TIMs (quatro.hpp:817-844), XY rows only for the 2-D solver (:396-402)
  const double rot_nb = prm->noise_bound * (2 / 1.0);  // params.noise_bound *= 2/scale (:850-852)
  std::vector<char> rmask;
  double R[9];
  if (prm->reg_mode == 1) {
    if (prm->using_pre_estimated_ryrx) {  // "Wrong reg type name is coming!" (:424-426)
      res->status = 2;
      return 2;
    }
    std::vector<double> ps(3 * (size_t)M), pd(3 * (size_t)M);
    for (int i = 0; i < M; ++i) {
      const int root = C[i], leaf = (i != M - 1) ? C[i + 1] : C[0];
      for (int a = 0; a < 3; ++a) {
        ps[3 * i + a] = s[3 * leaf + a] - s[3 * root + a];
        pd[3 * i + a] = (t[3 * leaf + a] - t[3 * root + a]) * (1 / 1.0);
      }
    }
    Gnc3Out g3;
    gnc_rotation3d(ps.data(), pd.data(), M, rot_nb, prm->rotation_gnc_factor, prm->rotation_max_iterations,
                   prm->rotation_cost_threshold, g3, rmask);
    res->cost = g3.cost;
    res->gnc_iters = g3.iters;
    memcpy(R, g3.R, sizeof(R));
  } else {
    std::vector<double> ps(2 * (size_t)M), pd(2 * (size_t)M);
    for (int i = 0; i < M; ++i) {
      const int root = C[i], leaf = (i != M - 1) ? C[i + 1] : C[0];
      for (int a = 0; a < 2; ++a) {
        ps[2 * i + a] = s[3 * leaf + a] - s[3 * root + a];
        pd[2 * i + a] = (t[3 * leaf + a] - t[3 * root + a]) * (1 / 1.0);  // pruned_dst_tims_ *= 1/scale, scale = 1
      }
    }
    GncOut g;
    gnc_rotation2d(ps.data(), pd.data(), M, rot_nb, prm->rotation_gnc_factor, prm->rotation_max_iterations,
                   prm->rotation_cost_threshold, g, rmask);
    res->cost = g.cost;
    res->gnc_iters = g.iters;
    const double Ry[9] = {g.R[0], g.R[1], 0, g.R[2], g.R[3], 0, 0, 0, 1};
    memcpy(R, Ry, sizeof(R));
  }
  if (prm->using_pre_estimated_ryrx) {  // solution_.rotation * estimated_RyRx_ (:419-423)
    double Rn[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        Rn[3 * r + c] = (R[3 * r] * prm->ryrx[c] + R[3 * r + 1] * prm->ryrx[3 + c]) + R[3 * r + 2] * prm->ryrx[6 + c];
    memcpy(R, Rn, sizeof(R));
  }
  // rotation inliers, cyclic chain rule (:857-874)
  std::vector<int> rot;
  for (int i = 0; i < M; ++i) {
    const int prev = (i == 0) ? M - 1 : i - 1;
    if (rmask[prev] && rmask[i]) rot.push_back(i);
  }
  res->n_rot_inliers = (int)rot.size();
  for (size_t i = 0; i < rot.size(); ++i) rot_inliers[i] = rot[i];
  const int NR = (int)rot.size();
  // points for COTE (:879-899)
  std::vector<int> sel;
  const bool use_rot = prm->using_rot_inliers_when_estimating_cote && NR > 0;
  std::vector<double> cs, cd;
  if (use_rot) {
    for (int i = 0; i < NR; ++i) sel.push_back(C[rot[i]]);
    for (int v : sel)
      for (int a = 0; a < 3; ++a) {
        cs.push_back(s[3 * v + a]);
        cd.push_back(t[3 * v + a]);
      }
  } else {
    sel = C;
    const double* Y = prm->ryrx;
    for (int v : sel) {
      const double x = s[3 * v], y = s[3 * v + 1], z = s[3 * v + 2];
      if (prm->using_pre_estimated_ryrx) {
        cs.push_back((Y[0] * x + Y[1] * y) + Y[2] * z);
        cs.push_back((Y[3] * x + Y[4] * y) + Y[5] * z);
        cs.push_back((Y[6] * x + Y[7] * y) + Y[8] * z);
      } else {
        cs.push_back(x);
        cs.push_back(y);
        cs.push_back(z);
      }
      for (int a = 0; a < 3; ++a) cd.push_back(t[3 * v + a]);
    }
  }
  const int N = (int)sel.size();
  // raw_translation = dst - scale * R * src (:597, :906)
  std::vector<double> raw[3];
  for (int a = 0; a < 3; ++a) raw[a].resize((size_t)N);
  for (int i = 0; i < N; ++i) {
    const double x = cs[3 * i], y = cs[3 * i + 1], z = cs[3 * i + 2];
    for (int a = 0; a < 3; ++a) raw[a][i] = cd[3 * i + a] - ((R[3 * a] * x + R[3 * a + 1] * y) + R[3 * a + 2] * z);
  }
  const double beta = prm->cote_noise_bound * sqrt(prm->cbar2);
  std::vector<char> inl((size_t)N, 1), tmp;
  double tr[3];
  for (int a = 0; a < 3; ++a) {
    tr[a] = cote_estimate(raw[a], beta, prm->cote_median != 0, tmp, &res->n_card[a]);
    for (int i = 0; i < N; ++i) inl[i] = inl[i] && tmp[i];
  }
  int nf = 0;
  for (int i = 0; i < N; ++i)
    if (inl[i]) final_inliers[nf++] = sel[i];
  res->n_final = nf;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) res->T[4 * r + c] = R[3 * r + c];
    res->T[4 * r + 3] = tr[r];
  }
  res->T[12] = res->T[13] = res->T[14] = 0;
  res->T[15] = 1;
  res->valid = 1;
  res->status = 0;
  return 0;
}

// Full path as driven by the reference demo (examples/run_global_registration.cpp:206-246):
// voxelize x2 -> FPFHManager::setFeaturePair -> Quatro::computeTransformation.
// counts_out: [n_src_vox, n_tgt_vox, L].  Work buffers are internal.
int qo_register_pair(const float* src_raw4, int Ps, const float* tgt_raw4, int Pt, float leaf, double r_normal,
                     double r_fpfh, float tuple_scale, unsigned long long seed, const qo_params* prm, qo_result* res,
                     int* counts_out, int* clique, int* final_inliers, int cap) {
  std::vector<float> sv(4 * (size_t)Ps), tv(4 * (size_t)Pt);
  int ns = voxelize(src_raw4, Ps, leaf, sv.data(), Ps);
  int nt = voxelize(tgt_raw4, Pt, leaf, tv.data(), Pt);
  if (ns < 0) {
    memcpy(sv.data(), src_raw4, sizeof(float) * 4 * (size_t)Ps);
    ns = Ps;
  }
  if (nt < 0) {
    memcpy(tv.data(), tgt_raw4, sizeof(float) * 4 * (size_t)Pt);
    nt = Pt;
  }
  std::vector<float> nrm_s(4 * (size_t)ns), nrm_t(4 * (size_t)nt), d_s(33 * (size_t)ns), d_t(33 * (size_t)nt);
  qo_fpfh(sv.data(), ns, r_normal, r_fpfh, nrm_s.data(), nullptr, d_s.data());
  qo_fpfh(tv.data(), nt, r_normal, r_fpfh, nrm_t.data(), nullptr, d_t.data());
  const int ccap = std::min(ns, nt);
  std::vector<int> corr(2 * (size_t)std::max(ccap, 1));
  const int L = match(sv.data(), ns, d_s.data(), tv.data(), nt, d_t.data(), 1, 1, tuple_scale, seed, corr.data(), ccap,
                      nullptr, nullptr);
  counts_out[0] = ns;
  counts_out[1] = nt;
  counts_out[2] = L;
  std::vector<float> ms(4 * (size_t)std::max(L, 1)), mt(4 * (size_t)std::max(L, 1));
  for (int i = 0; i < L; ++i)
    for (int a = 0; a < 4; ++a) {
      ms[4 * i + a] = a < 3 ? sv[4 * corr[2 * i] + a] : 0.f;
      mt[4 * i + a] = a < 3 ? tv[4 * corr[2 * i + 1] + a] : 0.f;
    }
  std::vector<int> cl((size_t)std::max(L, 1)), rot((size_t)std::max(L, 1)), fin((size_t)std::max(L, 1));
  const int st = qo_solve(ms.data(), mt.data(), L, prm, res, cl.data(), rot.data(), fin.data());
  for (int i = 0; i < res->n_clique && i < cap; ++i) clique[i] = cl[i];
  for (int i = 0; i < res->n_final && i < cap; ++i) final_inliers[i] = fin[i];
  return st;
}

}  // extern "C"

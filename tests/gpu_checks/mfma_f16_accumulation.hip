// mfma_f16_accumulation.hip — guard for the one hardware assumption in k_nn_f16's rounding bound (quatro_amd/csrc/match.hip):
// the f32 accumulation of v_mfma_f32_32x32x16_f16, inside one instruction and across a chain of seven, stays within
// 16 u * sum|terms| of the exact sum of the (exact) f16 x f16 products, u = 2^-24.  The bound budgets 16 u; this program
// measures the worst case over many random operand tiles shaped like the kernel's (two-way split halves of f32 values
// scaled by 128 / -256, norm pieces against 16384, zero padding) and over adversarial exponent spreads, and fails when
// it exceeds 8 u (half the budget).  Run by tests/test_gpu_parity.py::test_mfma_f16_accumulation_stays_inside_the_budget.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_f16_accumulation.hip -o mfma_f16_accumulation
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define K 112
#define CHECK(x)                                                         \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      return 2;                                                          \
    }                                                                    \
  } while (0)

// A[tile][32][K], B[tile][32][K] (row = base row resp. query column), C[tile][32][32] = sum_k A[i][k] B[j][k]
__global__ void k_chain(const _Float16* A, const _Float16* B, float* C) {
  const int lane = threadIdx.x, i = lane & 31, g = lane >> 5, t = blockIdx.x;
  const _Float16* a = A + (size_t)t * 32 * K + (size_t)i * K;
  const _Float16* b = B + (size_t)t * 32 * K + (size_t)i * K;
  f32x16 acc = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int m = 0; m < K / 16; ++m) {
    h8 av, bv;
    for (int e = 0; e < 8; ++e) {
      av[e] = a[16 * m + 8 * g + e];
      bv[e] = b[16 * m + 8 * g + e];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) C[(size_t)t * 1024 + (size_t)(8 * (r >> 2) + 4 * g + (r & 3)) * 32 + i] = acc[r];
}

static void split(float x, float scale, _Float16& h1, _Float16& h2) {
  const float xs = x * scale;
  h1 = (_Float16)xs;
  h2 = (_Float16)(xs - (float)h1);
}

int main() {
  const int tiles = 512;
  std::vector<_Float16> A((size_t)tiles * 32 * K), B(A.size());
  std::mt19937 rng(2026);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  double worst_all = 0;
  for (int mode = 0; mode < 4; ++mode) {
    for (size_t row = 0; row < (size_t)tiles * 32; ++row) {
      _Float16* a = &A[row * K];
      _Float16* b = &B[row * K];
      for (int k = 0; k < K; ++k) a[k] = b[k] = (_Float16)0.f;
      if (mode <= 1) {  // the kernel's operand shape: histogram-like rows (mode 0), or rows with a wide dynamic range (mode 1)
        float va[33], vb[33];
        double nb = 0;
        for (int k = 0; k < 33; ++k) {
          va[k] = mode == 0 ? U(rng) * 100.f * (U(rng) < 0.3f ? 0.f : 1.f) : std::ldexp(U(rng) + 1.f, (int)(U(rng) * 30) - 23);
          vb[k] = mode == 0 ? U(rng) * 100.f * (U(rng) < 0.3f ? 0.f : 1.f) : std::ldexp(U(rng) + 1.f, (int)(U(rng) * 30) - 23);
          nb += (double)va[k] * va[k];
        }
        _Float16 a1[33], a2[33], q1[33], q2[33];
        for (int k = 0; k < 33; ++k) {
          split(va[k], 128.f, a1[k], a2[k]);
          split(vb[k], -256.f, q1[k], q2[k]);
        }
        for (int k = 0; k < 33; ++k) {
          a[k] = a1[k]; b[k] = q1[k];
          a[33 + k] = a2[k]; b[33 + k] = q1[k];
          a[66 + k] = a1[k]; b[66 + k] = q2[k];
        }
        float nbf = (float)std::fmin(nb, 60000.0);
        _Float16 c1 = (_Float16)nbf;
        float r1 = nbf - (float)c1;
        _Float16 c2 = (_Float16)r1;
        _Float16 c3 = (_Float16)(r1 - (float)c2);
        a[99] = c1; a[100] = c2; a[101] = c3;
        b[99] = b[100] = b[101] = (_Float16)16384.f;
      } else {  // adversarial: every slot used, exponents spread over 2^24 (mode 2: mixed signs; mode 3: one sign)
        for (int k = 0; k < K; ++k) {
          float x = std::ldexp(U(rng) + 1.f, (int)(U(rng) * 24) - 12), y = std::ldexp(U(rng) + 1.f, (int)(U(rng) * 24) - 12);
          if (mode == 2 && (rng() & 1)) y = -y;
          a[k] = (_Float16)x;
          b[k] = (_Float16)y;
        }
      }
    }
    _Float16 *dA, *dB;
    float* dC;
    CHECK(hipMalloc(&dA, A.size() * 2));
    CHECK(hipMalloc(&dB, B.size() * 2));
    CHECK(hipMalloc(&dC, (size_t)tiles * 1024 * 4));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_chain, dim3(tiles), dim3(64), 0, 0, dA, dB, dC);
    CHECK(hipDeviceSynchronize());
    std::vector<float> C((size_t)tiles * 1024);
    CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < tiles; ++t)
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double s = 0, sa = 0;
          const _Float16* a = &A[((size_t)t * 32 + i) * K];
          const _Float16* b = &B[((size_t)t * 32 + j) * K];
          for (int k = 0; k < K; ++k) {
            const double p = (double)(float)a[k] * (double)(float)b[k];
            s += p;
            sa += std::fabs(p);
          }
          const double got = C[(size_t)t * 1024 + (size_t)i * 32 + j];
          if (!(got == got)) {
            std::printf("NaN at mode %d tile %d (%d,%d)\n", mode, t, i, j);
            return 1;
          }
          if (sa > 0) worst = std::fmax(worst, std::fabs(got - s) / (sa * 5.9604644775390625e-08));
        }
    std::printf("mode %d: max |error| = %.3f u * sum|terms| over %d entries\n", mode, worst, tiles * 1024);
    worst_all = std::fmax(worst_all, worst);
    CHECK(hipFree(dA));
    CHECK(hipFree(dB));
    CHECK(hipFree(dC));
  }
  std::printf("worst %.3f u (budget in the bound: 16 u; this check allows 8 u)\n", worst_all);
  return worst_all <= 8.0 ? 0 : 1;
}

// common.h — shared declarations for the gfx950 kernels and the C-ABI implementation.
// gfx950 only: wavefront = 64 lanes is hard-coded throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/qtr_math.h"
#include "../../include/quatro_hip.h"

#define QK_WAVE 64

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- wave-level helpers ------------------------------------------------------------------------
__device__ __forceinline__ int qk_lane() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  return __shfl(v, 0, 64);
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
// The "sum64" fold of include/qtr_math.h: p[l] += p[l+off] for off = 32..1; result broadcast from lane 0.
__device__ __forceinline__ double wave_sum64_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = __shfl_down(v, off, 64);
    v = v + o;  // lanes >= off compute garbage that is never consumed by lanes < off of later steps
  }
  return __shfl(v, 0, 64);
}
__device__ __forceinline__ float wave_sum64_f32(float v) {  // qm_sum64_fold_f across one wavefront
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_down(v, off, 64);
    v = v + o;
  }
  return __shfl(v, 0, 64);
}
__device__ __forceinline__ u64 lanemask_lt() { return (1ULL << qk_lane()) - 1ULL; }

// exclusive prefix sum of one int per lane across the wave
__device__ __forceinline__ int wave_excl_scan_i32(int v, int* total) {
  int x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off, 64);
    if (qk_lane() >= off) x += y;
  }
  *total = __shfl(x, 63, 64);
  return x - v;
}

// ---- host-side plumbing ------------------------------------------------------------------------
struct QtrDeviceBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

#define QTR_HIP_TRY(h, expr)                                                                       \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      snprintf((h)->err, sizeof((h)->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,           \
               hipGetErrorString(_e));                                                             \
      return QTR_ERR_HIP;                                                                          \
    }                                                                                              \
  } while (0)

static inline int qtr_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

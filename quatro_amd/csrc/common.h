// common.h — shared declarations for the gfx950 kernels and the C-ABI implementation.
// gfx950 only: wavefront = 64 lanes is hard-coded throughout.
#pragma once
// QTR_TEST_ENGINES (quatro_amd/build.py: libquatro_hip_testengines.so only): the comparison engines — all-exact and f32-MFMA
// nearest-neighbour search, the one-workgroup matcher tails, the peeling / sweep core-number kernels, the quadratic
// ranking — and the experiment knobs that select them through environment variables.  The product library is built
// WITHOUT it: one path, nothing of the above instantiated, the variables not even read.
#ifdef QTR_TEST_ENGINES
#define QTR_ENGINE_ENV(name) getenv(name)
#else
#define QTR_ENGINE_ENV(name) ((const char*)nullptr)
#endif
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/qtr_math.h"
#include "../../include/quatro_hip.h"

// -DQTR_NN_TIMING (diagnostic build, tests/probe/nn_stamps.py — never the shipped library): QTR_STAMP(kernel, point) lets
// thread 0 of the first 32 workgroups (x) of a launch's (y, z) = (0, 0) plane record the shader clock and the 100 MHz wall
// clock; the last launch of each kernel is what the probe reads back.
#ifdef QTR_NN_TIMING
#define QTR_STAMP_KERNELS 12
#define QTR_STAMP_POINTS 8
__device__ unsigned long long g_stamp[QTR_STAMP_KERNELS][32][QTR_STAMP_POINTS][2];
#define QTR_STAMP(kid, pt)                                                                   \
  if (threadIdx.x == 0 && blockIdx.x < 32 && blockIdx.y == 0 && blockIdx.z == 0) {          \
    g_stamp[kid][blockIdx.x][pt][0] = clock64();                                             \
    g_stamp[kid][blockIdx.x][pt][1] = wall_clock64();                                        \
  }
#else
#define QTR_STAMP(kid, pt)
#endif
enum { STAMP_DESC_PREP = 0, STAMP_HALF_TABLES, STAMP_SCATTER, STAMP_RECHECK, STAMP_CENTROIDS, STAMP_HIT_COMPACT, STAMP_CROSS,
       STAMP_NN_FINISH, STAMP_RECHECK1, STAMP_SPFH, STAMP_FPFH, STAMP_FINALIZE };


#define QK_WAVE 64

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- wave-level helpers ------------------------------------------------------------------------
__device__ __forceinline__ int qk_lane() { return (int)(threadIdx.x & 63); }

// Integer reductions across a wave whose 64 lanes are ALL active (every caller's case), on DPP row shifts and four
// scalar reads: ~11 instructions.  The shuffle form (seven __shfl_down, each an LDS-crossbar permute plus its index
// arithmetic: ~50 instructions and seven LDS round trips) cost 0.4 us wherever a wave was alone on a dependent chain —
// k_hcore_async's rows spent most of their time in it.  Results are uniform.
#define QK_DPP_SHR(n) (0x110 | (n))
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, QK_DPP_SHR(1), 0xf, 0xf, true);  // lanes shifted in from outside the row read 0
  v += __builtin_amdgcn_update_dpp(0, v, QK_DPP_SHR(2), 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, QK_DPP_SHR(4), 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, QK_DPP_SHR(8), 0xf, 0xf, true);  // lane 15 of every row of 16 holds the row's sum
  return (__builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31)) +
         (__builtin_amdgcn_readlane(v, 47) + __builtin_amdgcn_readlane(v, 63));
}
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, QK_DPP_SHR(1), 0xf, 0xf, false));  // (outside the row: the lane's own value)
  v = max(v, __builtin_amdgcn_update_dpp(v, v, QK_DPP_SHR(2), 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, QK_DPP_SHR(4), 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, QK_DPP_SHR(8), 0xf, 0xf, false));
  return max(max(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)),
             max(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, QK_DPP_SHR(1), 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, QK_DPP_SHR(2), 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, QK_DPP_SHR(4), 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, QK_DPP_SHR(8), 0xf, 0xf, false));
  return min(min(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)),
             min(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}
// exclusive prefix sum of one int per lane across the wave (all 64 lanes active)
__device__ __forceinline__ int wave_excl_scan_i32(int v, int* total) {
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, QK_DPP_SHR(1), 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, QK_DPP_SHR(2), 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, QK_DPP_SHR(4), 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, QK_DPP_SHR(8), 0xf, 0xf, true);  // inclusive inside every row of 16
  const int t0 = __builtin_amdgcn_readlane(x, 15), t1 = __builtin_amdgcn_readlane(x, 31), t2 = __builtin_amdgcn_readlane(x, 47),
            t3 = __builtin_amdgcn_readlane(x, 63);
  const int row = (int)(threadIdx.x & 63) >> 4;
  x += row == 0 ? 0 : row == 1 ? t0 : row == 2 ? t0 + t1 : t0 + t1 + t2;
  *total = (t0 + t1) + (t2 + t3);
  return x - v;
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
// The same fold — the same additions in the same tree — for partials that sit in BIT-REVERSED lane order: physical lane
// p holds the partial of logical lane brev6(p) (callers walk their elements with sum64_slot(lane) instead of lane).
// The fold's first four steps (logical offsets 32, 16, 8, 4 = physical offsets 1, 2, 4, 8) then stay inside a row of 16
// lanes and run on DPP row shifts instead of six LDS-crossbar permutes of a double; the last two (physical 16, 32) only
// concern four lanes, read as scalars.  The result is uniform.  (GNC-TLS: five such sums per iteration, ~17 iterations.)
__device__ __forceinline__ int sum64_slot(int lane) { return (int)(__brev((unsigned)lane) >> 26); }
template <int N>
__device__ __forceinline__ double dpp_row_shl_f64(double v) {  // lane i reads lane i + N of its row (garbage past the row)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 | N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 | N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum64_f64_brev(double v) {
  v = v + dpp_row_shl_f64<1>(v);  // logical p[l] += p[l + 32]
  v = v + dpp_row_shl_f64<2>(v);  // += p[l + 16]
  v = v + dpp_row_shl_f64<4>(v);  // += p[l + 8]
  v = v + dpp_row_shl_f64<8>(v);  // += p[l + 4]
  const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
  return (a + b) + (c + d);       // p[0] += p[2], p[1] += p[3]; p[0] += p[1]
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

// Ordered multi-workgroup compactions (the matcher's tail: k_cross_multi, k_pairs_multi; the voxel grid's centroids): every
// workgroup publishes the number
// of entries it keeps in its own word (count + 1; k_match_init zeroes the words) and reads its predecessors' words as
// they appear — device-scope relaxed atomics, no chain: a workgroup only ever waits for counts, which every workgroup
// publishes before it waits for anything, and workgroups are dispatched in index order.  Returns the number of entries
// in front of workgroup w, or -1 when a predecessor's word never appeared (bounded wait): the caller then writes NOTHING
// (its offsets would be wrong) and the failure travels to the host in the counters (MC_TAILERR -> MC_NCORR = -1).
__device__ __forceinline__ int tail_lookback(int* words, int w, int total, int* s_red /* [5] LDS */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) __hip_atomic_store(words + w, total + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int sum = 0, missing = 0;
  for (int u = tid; u < w; u += 256) {
    int v = 0;
    for (unsigned polls = 0; polls < (1u << 22); ++polls) {  // (bounded: a word that never appears cannot hang the device)
      v = __hip_atomic_load(words + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != 0) break;
      __builtin_amdgcn_s_sleep(2);
    }
    missing |= (v == 0);  // (kept apart from the sum: any number of missing words is one flag, never an overflow)
    sum += max(v - 1, 0);
  }
  sum = wave_sum_i32(sum);  // (a wave's sum stays far below 2^31: bit 31 of its word is free for the wave's flag)
  const u32 word = (u32)sum | (__ballot(missing) ? 0x80000000u : 0u);
  __syncthreads();  // (s_red may still be read from an earlier use)
  if (lane == 0) s_red[wave] = (int)word;
  __syncthreads();
  const u32 w0 = (u32)s_red[0], w1 = (u32)s_red[1], w2 = (u32)s_red[2], w3 = (u32)s_red[3];
  if ((w0 | w1 | w2 | w3) & 0x80000000u) return -1;
  return (int)(((w0 & 0x7fffffffu) + (w1 & 0x7fffffffu)) + ((w2 & 0x7fffffffu) + (w3 & 0x7fffffffu)));
}


// Views of a batched launch that live in device memory reach the kernels through this small by-value struct (NOT a
// bare pointer parameter and NOT a member of a large argument struct): with this shape the compiler's kernel-argument
// promotion also marks the pointers LOADED from the view as global-memory pointers; in the other two shapes every
// access behind the view degrades to flat_load / flat_store (checked in the ISA, hipcc 7.2).
template <typename V>
struct ViewExt {
  const V* ext;
  int pad[3];
};

// ---- host mailbox lines --------------------------------------------------------------------------
// A mailbox payload line is 16 ints in pinned host memory: words 0..14 carry data, word 15 a tag = seq ^ xor(data) ^
// MAIL_TAG_SALT.  The host accepts a line only when the tag matches what it recomputes, so it never consumes a line
// whose words have not all arrived — whatever order the stores reach host memory in (seen on MI355X: roughly one
// phase in 10^4 delivered the sequence word before the counters although a system-scope fence separates them).
#define MAIL_TAG_SALT 0x5bd1e995
// called by the 16 lanes of an ALIGNED 16-lane group (t = 0..15); lane 15's `value` is ignored
__device__ __forceinline__ void mail_store_line(int* __restrict__ line, int t, int value, int seq) {
  int x = (t < 15) ? value : 0;
  x ^= __shfl_xor(x, 1, 16);
  x ^= __shfl_xor(x, 2, 16);
  x ^= __shfl_xor(x, 4, 16);
  x ^= __shfl_xor(x, 8, 16);
  line[t] = (t == 15) ? (seq ^ x ^ MAIL_TAG_SALT) : value;
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

// ---- per-item views of a batched launch -----------------------------------------------------------
// Every kernel of the registration path takes the views (pointers + sizes) of the items it works on — clouds,
// pairs — and picks its own with a block index.  One or two items travel inside the kernel arguments; a larger batch
// (qtr_submit_batch) puts the array in device memory: the host fills a pinned staging area and one async copy on the
// launch stream carries it over.  A stage is rewound when the chunk that used it has completed.
struct ViewStage {
  char* h = nullptr;  // pinned host
  char* d = nullptr;  // device
  size_t cap = 0, off = 0;
};
static inline const void* stage_push(ViewStage* s, const void* src, size_t bytes, hipStream_t st) {
  const size_t padded = (bytes + 255) & ~(size_t)255;
  if (!s || !s->h || s->off + padded > s->cap) return nullptr;
  memcpy(s->h + s->off, src, bytes);
  if (hipMemcpyAsync(s->d + s->off, s->h + s->off, padded, hipMemcpyHostToDevice, st) != hipSuccess) return nullptr;
  const void* r = s->d + s->off;
  s->off += padded;
  return r;
}

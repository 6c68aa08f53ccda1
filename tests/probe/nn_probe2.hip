// nn_probe2.hip — ablation harness for the k_nn_mfma inner loop, round 2 (not part of the product).
// Variants of the top-2 epilogue and of its placement relative to the MFMA chains; clocks per 32-row tile and
// whole-launch rate.  Build: hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form] nn_probe2.hip -o nn_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32;
#define NN_K2 17

// ORDER 0: kk-outer, four accumulators, epilogue after the 68 MFMAs (the product kernel)
// ORDER 1: c-outer, two accumulator sets, the fold of block c-1 interleaved with the chain of block c (sched_group_barrier)
// ORDER 2: like 1 without the sched_group_barrier pins (source order only)
// PACK 1: accumulator-register index packed into the low mantissa bits (3 VALU / value); 0: value only (2 VALU / value)
template <int ORDER, int PACK, int WPS, int LD = 1, int FD = 1>
__global__ __launch_bounds__(256, WPS) void k_nn(const float* __restrict__ baseT, int nb_pad,
                                                 const float* __restrict__ queryT, int nq_pad, int tiles_per_split,
                                                 float* __restrict__ out, long long* clk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int qbase = (blockIdx.x * 4 + wave) * 128 + col;
  float q[4][NN_K2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk) q[a][kk] = queryT[(size_t)(2 * kk + half) * nq_pad + qbase + 32 * a];
  float b1[4], b2[4];
  int it1[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    b1[a] = b2[a] = INFINITY;
    it1[a] = -1;
  }
  const int ntiles = nb_pad / 32;
  const int t_begin = blockIdx.y * tiles_per_split, t_end = min(ntiles, t_begin + tiles_per_split);
  const float* bp = baseT + (size_t)half * nb_pad + col;
  float m0[NN_K2], m1[NN_K2];
  auto load_tile = [&](float* m, int t) {
    const float* p = bp + (size_t)(LD ? t : t_begin) * 32;
    if (!LD && t != t_begin) return;
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk) m[kk] = p[(size_t)(2 * kk) * nb_pad];
  };
  const f32x16 zero16 = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define FOLD(PACC, PC, R)                                                                                     \
  if (FD) {                                                                                                   \
    const float v_ = PACK ? __uint_as_float((__float_as_uint(PACC[R]) & 0xfffffff0u) | (u32)(R)) : PACC[R];    \
    b2[PC] = __builtin_amdgcn_fmed3f(b1[PC], b2[PC], v_);                                                     \
    b1[PC] = __builtin_amdgcn_fmed3f(b1[PC], v_, -INFINITY);                                                  \
  }
  f32x16 acc0, acc1;
  auto compute_tile = [&](const float* m, int t, bool pending) {
    if (ORDER == 0) {
      f32x16 acc[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = zero16;
#pragma unroll
      for (int kk = 0; kk < NN_K2; ++kk)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(m[kk], q[a][kk], acc[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float before = b1[a];
#pragma unroll
        for (int r = 0; r < 16; ++r) FOLD(acc[a], a, r)
        it1[a] = (b1[a] != before) ? t : it1[a];
      }
    } else {
#define CHAIN(ACC, C)                                                                                        \
  {                                                                                                          \
    ACC = zero16;                                                                                            \
    _Pragma("unroll") for (int kk = 0; kk < NN_K2; ++kk)                                                     \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(m[kk], q[C][kk], ACC, 0, 0, 0);                           \
  }
#define CHAIN_FOLD(ACC, C, PACC, PC, PT)                                                                     \
  {                                                                                                          \
    ACC = zero16;                                                                                            \
    const float before_ = b1[PC];                                                                            \
    _Pragma("unroll") for (int kk = 0; kk < NN_K2; ++kk) {                                                   \
      ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(m[kk], q[C][kk], ACC, 0, 0, 0);                             \
      if (kk < 16) FOLD(PACC, PC, kk)                                                                        \
      if (ORDER == 1) {                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
        __builtin_amdgcn_sched_group_barrier(0x002, PACK ? 3 : 2, 0);                                        \
      }                                                                                                      \
    }                                                                                                        \
    it1[PC] = (b1[PC] != before_) ? (PT) : it1[PC];                                                          \
  }
      if (pending) CHAIN_FOLD(acc0, 0, acc1, 3, t - 1)
      else CHAIN(acc0, 0)
      CHAIN_FOLD(acc1, 1, acc0, 0, t)
      CHAIN_FOLD(acc0, 2, acc1, 1, t)
      CHAIN_FOLD(acc1, 3, acc0, 2, t)
    }
  };
  const long long c0 = clock64();
  if (t_begin < t_end) load_tile(m0, t_begin);
  for (int t = t_begin; t < t_end; t += 2) {
    if (t + 1 < t_end) load_tile(m1, t + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute_tile(m0, t, t != t_begin);
    if (t + 1 < t_end) {
      if (t + 2 < t_end) load_tile(m0, t + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute_tile(m1, t + 1, true);
    }
  }
  if (ORDER != 0 && t_begin < t_end) {
    const float before_ = b1[3];
#pragma unroll
    for (int r = 0; r < 16; ++r) FOLD(acc1, 3, r)
    it1[3] = (b1[3] != before_) ? (t_end - 1) : it1[3];
  }
  const long long c1 = clock64();
  float s = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a) s += b1[a] + b2[a] + it1[a];
  out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 1) clk[0] = (c1 - c0) / (t_end - t_begin);
}


// Alternating pair: a 512-thread workgroup puts TWO waves on every SIMD; the halves of the workgroup take turns —
// one half issues the 68 MFMAs of its tile while the other folds the tile it computed before — separated by s_barrier,
// so the matrix pipe of a SIMD always has exactly one wave feeding it and the fold costs it nothing.
template <int PACK, int PRIO>
__global__ __launch_bounds__(512, 2) void k_nn_alt(const float* __restrict__ baseT, int nb_pad,
                                                    const float* __restrict__ queryT, int nq_pad, int tiles_per_split,
                                                    float* __restrict__ out, long long* clk) {
  constexpr int FD = 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = wave >> 2;
  const int col = lane & 31, half = lane >> 5;
  const int qbase = (blockIdx.x * 8 + wave) * 128 + col;
  float q[4][NN_K2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk) q[a][kk] = queryT[(size_t)(2 * kk + half) * nq_pad + qbase + 32 * a];
  float b1[4], b2[4];
  int it1[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    b1[a] = b2[a] = INFINITY;
    it1[a] = -1;
  }
  const int ntiles = nb_pad / 32;
  const int t_begin = blockIdx.y * tiles_per_split, t_end = min(ntiles, t_begin + tiles_per_split);
  const float* bp = baseT + (size_t)half * nb_pad + col;
  float m0[NN_K2];
  f32x16 acc[4];
  const f32x16 zero16 = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto load_tile = [&](float* m, int t) {
    const float* p = bp + (size_t)min(t, t_end - 1) * 32;
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk) m[kk] = p[(size_t)(2 * kk) * nb_pad];
  };
  auto mfma_tile = [&](const float* m) {
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = zero16;
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(m[kk], q[a][kk], acc[a], 0, 0, 0);
  };
  auto fold_tile = [&](int t) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float before = b1[a];
#pragma unroll
      for (int r = 0; r < 16; ++r) FOLD(acc[a], a, r)
      it1[a] = (b1[a] != before) ? t : it1[a];
    }
  };
  const long long c0 = clock64();
  load_tile(m0, t_begin);
  if (grp == 1) __syncthreads();  // the second half starts one phase later
  for (int t = t_begin; t < t_end; ++t) {
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    mfma_tile(m0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    load_tile(m0, t + 1);
    __syncthreads();
    fold_tile(t);
    __syncthreads();
  }
  if (grp == 0) __syncthreads();
  const long long c1 = clock64();
  float s = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a) s += b1[a] + b2[a] + it1[a];
  out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 1) clk[0] = (c1 - c0) / (t_end - t_begin);
}

template <typename KERN>
void run_k(KERN kern, const char* name, int nq, int nb, int slices, int threads = 256) {
  const int nq_pad = (nq + 511) / 512 * 512, nb_pad = (nb + 511) / 512 * 512;
  float *bT, *qT, *out;
  long long* clk;
  hipMalloc(&bT, (size_t)34 * nb_pad * 4);
  hipMalloc(&qT, (size_t)34 * nq_pad * 4);
  std::vector<float> h((size_t)34 * (nb_pad > nq_pad ? nb_pad : nq_pad));
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.01f;
  hipMemcpy(bT, h.data(), (size_t)34 * nb_pad * 4, hipMemcpyHostToDevice);
  hipMemcpy(qT, h.data(), (size_t)34 * nq_pad * 4, hipMemcpyHostToDevice);
  const int ntiles = nb_pad / 32, tps = (ntiles + slices - 1) / slices, ns = (ntiles + tps - 1) / tps;
  dim3 grid(nq_pad / (2 * threads), ns);
  hipMalloc(&out, (size_t)grid.x * grid.y * 512 * 4);
  hipMalloc(&clk, 8);
  hipMemset(clk, 0, 8);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, grid, dim3(threads), 0, 0, bT, nb_pad, qT, nq_pad, tps, out, clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  long long c;
  hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("%-44s nq %d nb %d grid %dx%d tiles/wg %d: %.1f us  %.1f TF/s  clk/tile %lld\n", name, nq, nb, grid.x, grid.y, tps,
         best * 1e3, 66.0 * nq_pad * nb_pad / (best * 1e-3) / 1e12, c);
  hipFree(bT);
  hipFree(qT);
  hipFree(out);
  hipFree(clk);
}

int main() {
  const int nq = 16384, nb = 17920;  // 32 query blocks, 560 tiles
  for (int sl : {16, 32}) {
    run_k(k_nn_alt<1, 0>, "alternating halves pack   512 thr", nq, nb, sl, 512);
    run_k(k_nn_alt<0, 0>, "alternating halves nopack 512 thr", nq, nb, sl, 512);
    run_k(k_nn_alt<0, 1>, "alternating halves nopack prio", nq, nb, sl, 512);
  }
  for (int sl : {8}) {
    run_k(k_nn<0, 0, 1, 0, 0>, "kk-outer: no loads, no fold", nq, nb, sl);
    run_k(k_nn<0, 0, 1, 1, 0>, "kk-outer: loads, no fold", nq, nb, sl);
    run_k(k_nn<0, 0, 1, 0, 1>, "kk-outer: no loads, fold", nq, nb, sl);
    run_k(k_nn<2, 0, 1, 0, 1>, "c-outer src: no loads, fold", nq, nb, sl);
    run_k(k_nn<1, 0, 1, 0, 1>, "c-outer sgb: no loads, fold", nq, nb, sl);
    run_k(k_nn<2, 0, 1, 1, 0>, "c-outer src: loads, no fold", nq, nb, sl);
  }
  for (int sl : {8}) {           // 256 workgroups
    run_k(k_nn<0, 1, 1>, "kk-outer pack            1 wave/SIMD", nq, nb, sl);
    run_k(k_nn<0, 1, 2>, "kk-outer pack            2 waves/SIMD", nq, nb, sl);
    run_k(k_nn<0, 0, 1>, "kk-outer nopack          1 wave/SIMD", nq, nb, sl);
    run_k(k_nn<0, 0, 2>, "kk-outer nopack          2 waves/SIMD", nq, nb, sl);
    run_k(k_nn<1, 1, 1>, "c-outer fold sgb pack    1 wave/SIMD", nq, nb, sl);
    run_k(k_nn<1, 0, 1>, "c-outer fold sgb nopack  1 wave/SIMD", nq, nb, sl);
    run_k(k_nn<1, 0, 2>, "c-outer fold sgb nopack  2 waves/SIMD", nq, nb, sl);
    run_k(k_nn<2, 0, 1>, "c-outer fold src nopack  1 wave/SIMD", nq, nb, sl);
    run_k(k_nn<2, 0, 2>, "c-outer fold src nopack  2 waves/SIMD", nq, nb, sl);
  }
  return 0;
}

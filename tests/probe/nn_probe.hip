// nn_probe.hip — ablation harness for the k_nn_mfma inner loop (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32;
#define NN_K2 17

template <int MODE>  // bit0: real loads per tile, bit1: epilogue, bit2: sched barriers
__global__ __launch_bounds__(256, 2) void k_nn(const float* __restrict__ baseT, int nb_pad, const float* __restrict__ queryT,
                                                 int nq_pad, int tiles_per_split, float* __restrict__ out, long long* clk) {
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
    const float* p = bp + (size_t)((MODE & 1) ? t : t_begin) * 32;
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk) m[kk] = p[(size_t)(2 * kk) * nb_pad];
  };
  auto compute_tile = [&](const float* m, int t) {
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (MODE & 8) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int kk = 0; kk < NN_K2; ++kk) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(m[kk], q[a][kk], acc[a], 0, 0, 0);
        const float before = b1[a];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = __uint_as_float((__float_as_uint(acc[a][r]) & 0xfffffff0u) | (u32)r);
          b2[a] = __builtin_amdgcn_fmed3f(b1[a], b2[a], v);
          b1[a] = __builtin_amdgcn_fmed3f(b1[a], v, -INFINITY);
        }
        it1[a] = (b1[a] != before) ? t : it1[a];
      }
      if (MODE & 16) {  // hand-placed interleave: epilogue of accumulator a-1 rides under the MFMA chain of a
        __builtin_amdgcn_sched_group_barrier(0x008, 17, 0);
#pragma unroll
        for (int i = 0; i < 51; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(m[kk], q[a][kk], acc[a], 0, 0, 0);
    if (MODE & 2) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float before = b1[a];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = __uint_as_float((__float_as_uint(acc[a][r]) & 0xfffffff0u) | (u32)r);
          b2[a] = __builtin_amdgcn_fmed3f(b1[a], b2[a], v);
          b1[a] = __builtin_amdgcn_fmed3f(b1[a], v, -INFINITY);
        }
        it1[a] = (b1[a] != before) ? t : it1[a];
      }
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a) b1[a] = fminf(b1[a], acc[a][0] + acc[a][5] + acc[a][10] + acc[a][15]);
    }
  };
  const long long c0 = clock64();
  if (MODE & 1) {
    if (t_begin < t_end) load_tile(m0, t_begin);
    for (int t = t_begin; t < t_end; t += 2) {
      if (t + 1 < t_end) load_tile(m1, t + 1);
      if (MODE & 4) __builtin_amdgcn_sched_barrier(0);
      compute_tile(m0, t);
      if (t + 1 < t_end) {
        if (t + 2 < t_end) load_tile(m0, t + 2);
        if (MODE & 4) __builtin_amdgcn_sched_barrier(0);
        compute_tile(m1, t + 1);
      }
    }
  } else {
    load_tile(m0, t_begin);
    for (int t = t_begin; t < t_end; ++t) compute_tile(m0, t);
  }
  const long long c1 = clock64();
  float s = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a) s += b1[a] + b2[a] + it1[a];
  out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 1) clk[0] = (c1 - c0) / (t_end - t_begin);
}

// software-pipelined variant: one wave per SIMD (512 registers available), MFMAs of tile t+1 interleaved with the
// epilogue of tile t inside the wave
template <int VPM>
__global__ __launch_bounds__(256, 1) void k_nn_pipe(const float* __restrict__ baseT, int nb_pad,
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
  f32x16 accA[4], accB[4];
  auto load_tile = [&](float* m, int t) {
    const float* p = bp + (size_t)min(t, t_end - 1) * 32;
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk) m[kk] = p[(size_t)(2 * kk) * nb_pad];
  };
  auto mfma_tile = [&](const float* m, f32x16* acc) {
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(m[kk], q[a][kk], acc[a], 0, 0, 0);
  };
  auto epi_tile = [&](const f32x16* acc, int t) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float before = b1[a];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = __uint_as_float((__float_as_uint(acc[a][r]) & 0xfffffff0u) | (u32)r);
        b2[a] = __builtin_amdgcn_fmed3f(b1[a], b2[a], v);
        b1[a] = __builtin_amdgcn_fmed3f(b1[a], v, -INFINITY);
      }
      it1[a] = (b1[a] != before) ? t : it1[a];
    }
  };
  auto interleave = [&]() {
    if (VPM > 0) {
#pragma unroll
      for (int i = 0; i < 68; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      }
    }
  };
  const long long c0 = clock64();
  load_tile(m0, t_begin);
  load_tile(m1, t_begin + 1);
  mfma_tile(m0, accA);
  int t = t_begin;
  for (; t + 2 < t_end; t += 2) {
    load_tile(m0, t + 2);
    mfma_tile(m1, accB);
    epi_tile(accA, t);
    interleave();
    load_tile(m1, t + 3);
    mfma_tile(m0, accA);
    epi_tile(accB, t + 1);
    interleave();
  }
  // tail: accA holds tile t; m1 holds tile t+1 (if it exists)
  epi_tile(accA, t);
  if (t + 1 < t_end) {
    mfma_tile(m1, accB);
    epi_tile(accB, t + 1);
  }
  const long long c1 = clock64();
  float s = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a) s += b1[a] + b2[a] + it1[a];
  out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 1) clk[0] = (c1 - c0) / (t_end - t_begin);
}

template <typename KERN>
void run_k(KERN kern, const char* name, int nq, int nb, int slices) {
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
  dim3 grid(nq_pad / 512, ns);
  hipMalloc(&out, (size_t)grid.x * grid.y * 256 * 4);
  hipMalloc(&clk, 8);
  hipMemset(clk, 0, 8);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, bT, nb_pad, qT, nq_pad, tps, out, clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  long long c;
  hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("%-34s nq %d nb %d grid %dx%d tiles/wave %d: %.1f us  %.1f TF/s  clk/tile %lld\n", name, nq, nb, grid.x, grid.y, tps,
         best * 1e3, 66.0 * nq_pad * nb_pad / (best * 1e-3) / 1e12, c);
  hipFree(bT);
  hipFree(qT);
  hipFree(out);
  hipFree(clk);
}

int main() {
  for (int sl : {15}) {
    run_k(k_nn<0>, "mfma only (no loads, no epilogue)", 8525, 9027, sl);
    run_k(k_nn<7>, "loads+epilogue+schedbarrier", 8525, 9027, sl);
    run_k(k_nn<15>, "a-outer order", 8525, 9027, sl);
    run_k(k_nn<11>, "a-outer order, no schedbarrier", 8525, 9027, sl);
    run_k(k_nn<27>, "a-outer + sched_group_barrier", 8525, 9027, sl);
    run_k(k_nn<31>, "a-outer + sgb + schedbarrier", 8525, 9027, sl);
  }
  return 0;
}

// mfma_probe.hip — ground truth for the issue rate of v_mfma_f32_32x32x2_f32 on gfx950 (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int VALU_PER_MFMA>
__global__ __launch_bounds__(256) void k_probe(float* out, long long* clk, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) acc[a] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = x + q;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < VALU_PER_MFMA; ++q) v[q & 7] = fminf(v[q & 7], v[(q + 1) & 7] + 1.0f);
      }
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}

// Does a top-2 epilogue that READS a finished accumulator overlap with MFMAs that write OTHER accumulators?
// PATTERN 0: 8 accumulators, epilogue on the tuple finished in the previous group while 4 others are being written
// PATTERN 1: same MFMAs, the epilogue works on plain registers (no accumulator reads)
template <int PATTERN>
__global__ __launch_bounds__(256) void k_probe2(float* out, long long* clk, int iters) {
  f32x16 accA[4], accB[4];
  for (int a = 0; a < 4; ++a) {
    accA[a] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    accB[a] = accA[a];
  }
  float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
  float b1[4] = {1e30f, 1e30f, 1e30f, 1e30f}, b2[4] = {1e30f, 1e30f, 1e30f, 1e30f};
  float plain[16];
  for (int q = 0; q < 16; ++q) plain[q] = x + q;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    // group 1: MFMAs into accA, epilogue on accB
#pragma unroll
    for (int kk = 0; kk < 17; ++kk)
#pragma unroll
      for (int a = 0; a < 4; ++a) accA[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, accA[a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float src = PATTERN == 0 ? accB[a][r] : plain[r];
        const float v = __uint_as_float((__float_as_uint(src) & 0xfffffff0u) | (unsigned)r);
        b2[a] = __builtin_amdgcn_fmed3f(b1[a], b2[a], v);
        b1[a] = __builtin_amdgcn_fmed3f(b1[a], v, -INFINITY);
      }
    // group 2: MFMAs into accB, epilogue on accA
#pragma unroll
    for (int kk = 0; kk < 17; ++kk)
#pragma unroll
      for (int a = 0; a < 4; ++a) accB[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, accB[a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float src = PATTERN == 0 ? accA[a][r] : plain[r];
        const float v = __uint_as_float((__float_as_uint(src) & 0xfffffff0u) | (unsigned)r);
        b2[a] = __builtin_amdgcn_fmed3f(b1[a], b2[a], v);
        b1[a] = __builtin_amdgcn_fmed3f(b1[a], v, -INFINITY);
      }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int a = 0; a < 4; ++a) {
    s += b1[a] + b2[a];
    for (int r = 0; r < 16; ++r) s += accA[a][r] + accB[a][r];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}
template <int PATTERN>
void run2(const char* name) {
  float* out;
  long long* clk;
  hipMalloc(&out, (size_t)256 * 256 * 4);
  hipMalloc(&clk, 16);
  const int iters = 100;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_probe2<PATTERN>, dim3(256), dim3(256), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
  }
  long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-50s clk per 68-MFMA tile %.0f (68 x 64 = 4352)\n", name, (double)h[0] / (2.0 * iters));
  hipFree(out);
  hipFree(clk);
}


// The c-outer design in isolation: a dependent chain of 17 MFMAs into one accumulator while the 16 values of the OTHER
// accumulator (finished one chain ago) are folded into a running top-2, one value per MFMA slot.
// SGB 1: order pinned with sched_group_barrier; PACK 1: index packed into the mantissa (3 VALU / value), 0: 2 VALU
template <int SGB, int PACK, int SKIP0>
__global__ __launch_bounds__(256) void k_probe3(float* out, long long* clk, int iters) {
  f32x16 acc0 = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
  const f32x16 zero16 = acc0;
  float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
  float b1 = 1e30f, b2 = 1e30f;
  float ninf = -INFINITY;
  asm volatile("" : "+v"(ninf));  // opaque: keeps med3(b1, v, -inf) a v_med3 (a literal folds to canonicalise + v_min)
  const long long c0 = clock64();
#define P3_FOLD(PACC, R)                                                                              \
  {                                                                                                   \
    const float v_ = PACK ? __uint_as_float((__float_as_uint(PACC[R]) & 0xfffffff0u) | (unsigned)(R)) : PACC[R]; \
    b2 = __builtin_amdgcn_fmed3f(b1, b2, v_);                                                         \
    b1 = __builtin_amdgcn_fmed3f(b1, v_, ninf);                                                       \
  }
#define P3_CHAIN(ACC, PACC)                                                                           \
  {                                                                                                   \
    ACC = zero16;                                                                                     \
    asm volatile("" : "+v"(x), "+v"(y));  /* opaque operands: the four chains are not common subexpressions */ \
    _Pragma("unroll") for (int kk = 0; kk < 17; ++kk) {                                               \
      ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, ACC, 0, 0, 0);                                 \
      if (SKIP0 ? (kk >= 1) : (kk < 16)) P3_FOLD(PACC, (SKIP0 ? kk - 1 : kk))                         \
      if (SGB) {                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
        __builtin_amdgcn_sched_group_barrier(0x002, PACK ? 3 : 2, 0);                                 \
      }                                                                                               \
    }                                                                                                 \
  }
  for (int it = 0; it < iters; ++it) {
    P3_CHAIN(acc0, acc1)
    P3_CHAIN(acc1, acc0)
    P3_CHAIN(acc0, acc1)
    P3_CHAIN(acc1, acc0)
  }
  const long long c1 = clock64();
  float s = b1 + b2;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}
template <int SGB, int PACK, int SKIP0>
void run3(const char* name) {
  float* out;
  long long* clk;
  hipMalloc(&out, (size_t)256 * 256 * 4);
  hipMalloc(&clk, 16);
  const int iters = 100;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k_probe3<SGB, PACK, SKIP0>), dim3(256), dim3(256), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
  }
  long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-50s clk per 68-MFMA tile %.0f (68 x 64 = 4352)\n", name, (double)h[0] / iters);
  hipFree(out);
  hipFree(clk);
}

template <int NACC, int VPM>
void run(const char* name, int blocks, int threads) {
  float* out;
  long long* clk;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  hipMalloc(&clk, 16);
  const int iters = 200;
  hipLaunchKernelGGL((k_probe<NACC, VPM>), dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe<NACC, VPM>), dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double nm = (double)iters * 16 * NACC;
  printf("%-28s blocks %4d thr %3d: clk/mfma %.1f  wall_ns/mfma %.2f  => clock %.2f GHz  kernel %.1f us  TF/s %.1f\n", name,
         blocks, threads, h[0] / nm, h[1] * 10.0 / nm, (double)h[0] / (h[1] * 10.0), ms * 1e3,
         nm * 4096.0 * blocks * (threads / 64) / (ms * 1e-3) / 1e12);
  hipFree(out);
  hipFree(clk);
}

// occupancy check: a kernel that holds NV live VGPRs; do two workgroups of 256 threads share a CU?
template <int NV>
__global__ __launch_bounds__(256, 2) void k_occ(float* out, unsigned* stats, int spin) {
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 0.5f + i;
  const unsigned w0 = (unsigned)wall_clock64();
  for (int it = 0; it < spin; ++it) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = fmaf(v[i], 1.0001f, v[(i + 1) % NV]);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    atomicMax(&stats[0], ~w0);
    atomicMax(&stats[1], w0);
  }
}
template <int NV>
void occ(int blocks) {
  float* out;
  unsigned* st;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&st, 16);
  hipMemset(st, 0, 16);
  hipLaunchKernelGGL(k_occ<NV>, dim3(blocks), dim3(256), 0, 0, out, st, 2000);
  hipDeviceSynchronize();
  unsigned h[2];
  hipMemcpy(h, st, 8, hipMemcpyDeviceToHost);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)k_occ<NV>);
  int nb = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_occ<NV>, 256, 0);
  printf("live %3d -> numRegs %3d, runtime says %d blocks/CU; %d blocks: start spread %.2f us\n", NV, fa.numRegs, nb, blocks,
         (h[1] - ~h[0]) / 100.0);
  hipFree(out);
  hipFree(st);
}

int main() {
  run3<0, 1, 0>("chain + fold, source order, pack");
  run3<1, 1, 0>("chain + fold, sgb, pack");
  run3<1, 0, 0>("chain + fold, sgb, nopack");
  run3<0, 0, 0>("chain + fold, source order, nopack");
  run3<1, 1, 1>("chain + fold, sgb, pack, skip slot 0");
  run3<1, 0, 1>("chain + fold, sgb, nopack, skip slot 0");
  run2<0>("epilogue reads the other accumulator set");
  run2<1>("epilogue on plain registers");
  occ<100>(512);
  occ<120>(512);
  occ<200>(512);
  occ<230>(512);
  occ<245>(512);
  run<4, 0>("4 acc, no valu", 1, 64);
  run<4, 0>("4 acc, no valu", 256, 256);
  run<4, 0>("4 acc, no valu", 512, 256);
  run<4, 0>("4 acc, no valu", 1024, 256);
  run<1, 0>("1 acc (dependent)", 256, 256);
  run<4, 4>("4 acc, 4 valu/mfma", 256, 256);
  run<4, 7>("4 acc, 7 valu/mfma", 256, 256);
  run<4, 7>("4 acc, 7 valu/mfma", 512, 256);
  run<4, 12>("4 acc, 12 valu/mfma", 256, 256);
  run<4, 12>("4 acc, 12 valu/mfma", 512, 256);
  return 0;
}

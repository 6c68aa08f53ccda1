// stages.hip — the individually callable stages of the reference class (computeTIMs, solveForScale,
// solveForRotation2D, estimate; include/quatro.hpp:307-386,430-572,618-747) as small kernels over the SAME device
// code the fused path uses (pair_consistent, gnc_wave, cote_axis4 from solver.hip).  computeTransformation never
// materialises TIMs and never calls these; they exist so that code written against the reference's public stage
// methods keeps working on the device.
#include "common.h"
#include "solver.h"

// tims[:, start(i) + (j-i-1)] = v[:, j] - v[:, i], map = (i, j); start(i) = i*N - i(i+1)/2   (reference :318-341)
__global__ __launch_bounds__(256) void k_compute_tims(const double* __restrict__ v, int N, long long K,
                                                      double* __restrict__ tims, int* __restrict__ map) {
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  // invert the triangular index: largest i with start(i) <= k
  const double Nd = (double)N;
  long long i = (long long)floor((2.0 * Nd - 1.0 - sqrt((2.0 * Nd - 1.0) * (2.0 * Nd - 1.0) - 8.0 * (double)k)) * 0.5);
  if (i < 0) i = 0;
  while (i > 0 && i * N - i * (i + 1) / 2 > k) --i;
  while ((i + 1) * N - (i + 1) * (i + 2) / 2 <= k) ++i;
  const long long j = k - (i * N - i * (i + 1) / 2) + i + 1;
#pragma unroll
  for (int r = 0; r < 3; ++r) tims[(size_t)r * K + k] = v[(size_t)r * N + j] - v[(size_t)r * N + i];
  map[k] = (int)i;
  map[(size_t)K + k] = (int)j;
}

// scale-consistency mask over TIM columns (reference solveForScale :355-386, scale == 1), same predicate and
// evaluation order as k_graph_build
__global__ __launch_bounds__(256) void k_scale_mask(const double* __restrict__ a, const double* __restrict__ b,
                                                    long long K, double beta, unsigned char* __restrict__ mask) {
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const double dx = a[k], dy = a[(size_t)K + k], dz = a[2 * (size_t)K + k];
  const double ex = b[k], ey = b[(size_t)K + k], ez = b[2 * (size_t)K + k];
  const double s = dx * dx + (dy * dy + dz * dz), t = ex * ex + (ey * ey + ez * ez);
  mask[k] = pair_consistent(s, t, beta, beta * beta) ? 1 : 0;
}

// GNC-TLS yaw on one wavefront.  src/dst: row-major 2 x M.  out: R[4], cost, then (as doubles) iters; weights in wt.
__global__ __launch_bounds__(64) void k_gnc_only(const double* __restrict__ src, const double* __restrict__ dst, int M,
                                                 double rot_nb, double gnc_factor, int max_it, double cost_thr,
                                                 double* __restrict__ wt, double* __restrict__ out,
                                                 unsigned char* __restrict__ inl) {
  const int lane = threadIdx.x;
  for (int j = lane; j < M; j += 64) wt[j] = 1.0;
  __syncthreads();
  double R[4], cost;
  int iters;
  gnc_wave(sum64_slot(lane), src, src + M, dst, dst + M, wt, M, rot_nb, gnc_factor, max_it, cost_thr, R, &cost, &iters);
  __syncthreads();
  for (int j = lane; j < M; j += 64) inl[j] = (wt[j] >= 0.4) ? 1 : 0;  // reference :566-570
  if (lane == 0) {
    out[0] = R[0];
    out[1] = R[1];
    out[2] = R[2];
    out[3] = R[3];
    out[4] = cost;
    out[5] = (double)iters;
  }
}

// GNC-TLS 3-DoF rotation on one wavefront.  src/dst: row-major 3 x M.  out: R[9], cost, then (as double) iters.
__global__ __launch_bounds__(64) void k_gnc3d_only(const double* __restrict__ src, const double* __restrict__ dst, int M,
                                                   double rot_nb, double gnc_factor, int max_it, double cost_thr,
                                                   double* __restrict__ wt, double* __restrict__ out,
                                                   unsigned char* __restrict__ inl) {
  const int lane = threadIdx.x;
  for (int j = lane; j < M; j += 64) wt[j] = 1.0;
  __syncthreads();
  double R[9], cost;
  int iters;
  gnc3_wave(sum64_slot(lane), src, src + M, src + 2 * (size_t)M, dst, dst + M, dst + 2 * (size_t)M, wt, M, rot_nb, gnc_factor, max_it,
            cost_thr, R, &cost, &iters);
  __syncthreads();
  for (int j = lane; j < M; j += 64) inl[j] = (wt[j] >= 0.4) ? 1 : 0;
  if (lane == 0) {
    for (int a = 0; a < 9; ++a) out[a] = R[a];
    out[9] = cost;
    out[10] = (double)iters;
  }
}

// COTE on one group of four wavefronts.  R: per-element ranges (Quatro::estimate accepts any, reference
// include/quatro.hpp:618-747) or null for the uniform `range` the class itself passes.  scratch: 14*N doubles + 2*N ints
// of global memory.
__global__ __launch_bounds__(256) void k_cote_only(const double* __restrict__ X, int N, double range,
                                                   const double* __restrict__ R, int median_sel,
                                                   double* __restrict__ scratch_f, int* __restrict__ scratch_i,
                                                   double* __restrict__ out, unsigned char* __restrict__ inl) {
  __shared__ double s_bc[4], s_redc[4];
  __shared__ int s_redi[4];
  const int nc = 2 * N;
  const CoteOut co = cote_axis4(true, (int)threadIdx.x, X, N, nc, range, R, median_sel, scratch_f, scratch_i,
                                scratch_f + 2 * (size_t)N, s_bc, s_redc, s_redi, nullptr, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 256) inl[i] = (fabs(X[i] - co.est) <= (R ? R[i] : range)) ? 1 : 0;  // :741-744
  if (threadIdx.x == 0) {
    out[0] = co.est;
    out[1] = (double)co.ncard;
  }
}

// match.hip — 33-D reciprocal nearest-neighbour matching, cross-check, tuple test, compaction.
//
// Replaces teaser::Matcher::calculateCorrespondences (reference include/teaser_utils/feature_matcher.h:
// 42-74) / advancedMatching (src/teaser_utils/feature_matcher.cc:77-265) and its two FLANN kd-trees.
// The exact distance is flann::L2<float>'s: groups of four, result += ((d0^2+d1^2)+d2^2)+d3^2, then the
// tail term; ties go to the lowest index.
//
// Two nearest-neighbour engines produce bit-identical tables:
//  * k_nn_exact : VALU evaluation of the exact distance for every pair (LDS-tiled, broadcast reads);
//  * k_nn_mfma  : the dense form |a|^2+|b|^2-2ab through v_mfma_f32_32x32x2_f32 with per-row best /
//                 second-best tracking; rows whose two best candidates are closer than a rigorous
//                 rounding bound are re-decided by k_nn_exact_rows.  (Selected by match_enqueue.)
#include "common.h"
#include "frontend.h"

#define NN_TILE 64
#define NN_STRIDE 36  // floats per staged descriptor row (33 + 3 zero pad; 16-byte aligned rows)

__device__ __forceinline__ float l2_flann33(const float* a, const float* b /* LDS row, stride-36 */) {
  float result = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const float4 bv = *(const float4*)(b + 4 * g);
    const float d0 = a[4 * g] - bv.x, d1 = a[4 * g + 1] - bv.y, d2 = a[4 * g + 2] - bv.z, d3 = a[4 * g + 3] - bv.w;
    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  const float d = a[32] - b[32];
  result += d * d;
  return result;
}

__device__ __forceinline__ float l2_flann33_g(const float* a, const float* __restrict__ b /* global row */) {
  float result = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const float d0 = a[4 * g] - b[4 * g], d1 = a[4 * g + 1] - b[4 * g + 1], d2 = a[4 * g + 2] - b[4 * g + 2],
                d3 = a[4 * g + 3] - b[4 * g + 3];
    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  const float d = a[32] - b[32];
  result += d * d;
  return result;
}


// grid (ceil(nA/256), nsplit): thread = one query of A, blockIdx.y = slice of B
__global__ __launch_bounds__(256) void k_nn_exact(const float* __restrict__ A, int nA, const float* __restrict__ B,
                                                  int nB, u64* __restrict__ best, const int* __restrict__ rows,
                                                  int nrows) {
  __shared__ __attribute__((aligned(16))) float tile[NN_TILE * NN_STRIDE];
  const int q = blockIdx.x * 256 + threadIdx.x;
  // optional indirection: only the listed rows of A are (re)computed
  const int nq = rows ? nrows : nA;
  const int a_idx = (q < nq) ? (rows ? rows[q] : q) : -1;
  float a[33];
#pragma unroll
  for (int t = 0; t < 33; ++t) a[t] = (a_idx >= 0) ? A[(size_t)a_idx * 33 + t] : 0.f;
  const int per = (nB + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nB, b0 + per);
  float bd = INFINITY;
  int bi = -1;
  for (int base = b0; base < b1; base += NN_TILE) {
    const int m = min(NN_TILE, b1 - base);
    __syncthreads();
    for (int e = threadIdx.x; e < NN_TILE * NN_STRIDE; e += 256) {
      const int r = e / NN_STRIDE, c = e - r * NN_STRIDE;
      tile[e] = (r < m && c < 33) ? B[(size_t)(base + r) * 33 + c] : 0.f;
    }
    __syncthreads();
    if (a_idx >= 0) {
      for (int r = 0; r < m; ++r) {
        const float d = l2_flann33(a, tile + r * NN_STRIDE);
        if (d < bd) {
          bd = d;
          bi = base + r;
        }
      }
    }
  }
  if (a_idx >= 0 && bi >= 0) atomicMin(&best[a_idx], ((u64)__float_as_uint(bd) << 32) | (u32)bi);
}


// =================================================================================================
// MFMA engine.  d~(a,b) = |a|^2 + (|b|^2 - 2 a.b): the bracket is ONE f32 MFMA chain over K = 34
// (33 descriptor bins + one slot carrying |b|^2 against a constant 1), issued as 17 x
// v_mfma_f32_32x32x2_f32.  The streamed operand is the BASE cloud (M side, 32 rows per tile), the
// stationary operand the QUERIES (N side): in the 32x32 accumulator layout a lane owns ONE query column
// (col = lane & 31) and 16 base rows, so the running best / second-best per query live in that lane's
// registers and no cross-lane reduction is needed until the very end.  Descriptors are pre-transposed
// to k-major ([34][n_pad]) so every fragment load is two coalesced 128-byte segments; no LDS at all.
// A query's approximate winner is accepted only when second-best - best exceeds twice a rigorous
// rounding bound; otherwise the row is re-decided by k_nn_exact (bit-identical tables either way).
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define NN_K2 17      // k pairs: K = 34
#define NN_QPW 128    // queries per wave (4 accumulators of 32 columns)
#define NN_QPB 512    // queries per workgroup (4 waves)

// desc[n][33] -> baseT[34][n_pad] (row 33 = |b|^2; pad rows get 1e30 so they never win) and
// queryT[34][n_pad] (-2 * desc, row 33 = 1); also the norms (binary64 sum rounded once) and the
// largest norm (as ordered bits).
__device__ __forceinline__ void d_desc_prep(const float* __restrict__ desc, int n, int n_pad,
                                                   float* __restrict__ baseT, float* __restrict__ queryT,
                                                   float* __restrict__ norms, u32* __restrict__ max_norm_bits) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float nrm = 0.f;
  if (i < n_pad) {
    if (i < n) {
      double acc = 0.0;
      for (int k = 0; k < 33; ++k) {
        const float v = desc[(size_t)i * 33 + k];
        acc += (double)v * (double)v;
        baseT[(size_t)k * n_pad + i] = v;
        queryT[(size_t)k * n_pad + i] = -2.0f * v;
      }
      nrm = (float)acc;
      baseT[(size_t)33 * n_pad + i] = nrm;
      queryT[(size_t)33 * n_pad + i] = 1.0f;
      norms[i] = nrm;
    } else {
      for (int k = 0; k < 33; ++k) {
        baseT[(size_t)k * n_pad + i] = 0.f;
        queryT[(size_t)k * n_pad + i] = 0.f;
      }
      baseT[(size_t)33 * n_pad + i] = 1e30f;
      queryT[(size_t)33 * n_pad + i] = 1.0f;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) nrm = fmaxf(nrm, __shfl_xor(nrm, off, 64));
  if (qk_lane() == 0 && nrm > 0.f) atomicMax(max_norm_bits, __float_as_uint(nrm));
}

__global__ __launch_bounds__(256) void k_desc_prep2(const float* d0, int n0, int p0, float* bT0, float* qT0, float* nr0,
                                                    u32* mx0, const float* d1, int n1, int p1, float* bT1, float* qT1,
                                                    float* nr1, u32* mx1) {
  if (blockIdx.y == 0)
    d_desc_prep(d0, n0, p0, bT0, qT0, nr0, mx0);
  else
    d_desc_prep(d1, n1, p1, bT1, qT1, nr1, mx1);
}

// best and second-best approximate distance of a query over one base slice (index of the best only)
struct NnPartial {
  float b1, b2;
  int i1;
  int pad;
};

// grid.x = nq_pad / 512 query groups, grid.y = base slices.  Each wave keeps 4 x 32 query columns
// stationary (68 VGPRs), streams 32-row base tiles (17 coalesced dword loads per lane, software
// prefetched one tile ahead in a second register set) and issues 68 MFMAs per tile.  The epilogue is
// five branch-free VALU ops per accumulator value (cmp / max / min / min / cndmask), i.e. ~0.15 of the
// MFMA issue time, so the kernel is matrix-pipe bound.
__global__ __launch_bounds__(256, 2) void k_nn_mfma(const float* __restrict__ baseT, int nb_pad,
                                                    const float* __restrict__ queryT, int nq_pad,
                                                    int tiles_per_split, NnPartial* __restrict__ partial,
                                                    int* __restrict__ dbg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long dbg_c0 = clock64(), dbg_w0 = wall_clock64();
  const int col = lane & 31, half = lane >> 5;
  const int qbase = (blockIdx.x * 4 + wave) * NN_QPW + col;
  float q[4][NN_K2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk) q[a][kk] = queryT[(size_t)(2 * kk + half) * nq_pad + qbase + 32 * a];
  // running best / second best per query column.  The accumulator register r a value came from rides in the
  // four low mantissa bits of the value itself (v_and_or_b32), so the update is three VALU ops per value —
  // pack, second = med3(best, second, v), best = min(best, v) — and the winning TILE is found once per tile
  // by noticing that best changed.  The <16 ulp perturbation is part of the rounding bound of the finish.
  float b1[4], b2[4];
  int it1[4];  // tile of the best
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
    const float* p = bp + (size_t)t * 32;
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk) m[kk] = p[(size_t)(2 * kk) * nb_pad];
  };
  auto compute_tile = [&](const float* m, int t) {
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < NN_K2; ++kk)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(m[kk], q[a][kk], acc[a], 0, 0, 0);
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
  if (t_begin < t_end) load_tile(m0, t_begin);
  for (int t = t_begin; t < t_end; t += 2) {
    if (t + 1 < t_end) load_tile(m1, t + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (the scheduler sank it otherwise)
    compute_tile(m0, t);
    if (t + 1 < t_end) {
      if (t + 2 < t_end) load_tile(m0, t + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute_tile(m1, t + 1);
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    // decode the best's row, then merge the two lanes (half 0 / half 1) that own the same query column
    const int r = (int)(__float_as_uint(b1[a]) & 15u);
    int row = (it1[a] < 0) ? -1 : (it1[a] * 32 + 4 * half + (r & 3) + 8 * (r >> 2));
    const float ob1 = __shfl_xor(b1[a], 32, 64), ob2 = __shfl_xor(b2[a], 32, 64);
    const int orow = __shfl_xor(row, 32, 64);
    const bool take = (ob1 < b1[a]) || (ob1 == b1[a] && orow >= 0 && (row < 0 || orow < row));
    const float nb2 = fminf(fminf(b2[a], ob2), take ? b1[a] : ob1);
    const float nb1 = take ? ob1 : b1[a];
    row = take ? orow : row;
    if (half == 0) {
      NnPartial p;
      p.b1 = nb1;
      p.b2 = nb2;
      p.i1 = row;
      p.pad = 0;
      partial[(size_t)(qbase + 32 * a) * gridDim.y + blockIdx.y] = p;
    }
  }
  if (dbg && threadIdx.x == 0) {
    const u32 w0 = (u32)dbg_w0, w1 = (u32)wall_clock64();
    if (blockIdx.x == 1 && blockIdx.y == 1) dbg[12] = (int)((clock64() - dbg_c0) / (t_end - t_begin));  // clk per tile
    atomicMax((u32*)&dbg[13], (w1 - w0));  // longest workgroup life, 10 ns
    atomicAdd((u32*)&dbg[14], (w1 - w0));  // sum of lives
    atomicAdd((u32*)&dbg[15], 1u);
  }
}

// Merge the per-slice partials and decide each query row: if best + 2 eps < second-best the approximate
// winner IS the exact arg-min; otherwise the row goes to the exact re-check list.
// eps bounds |d~ - d_exact-order| for every pair of the row: with u = 2^-24, the 35-term fma chain
// contributes 35u(|b|^2 + 2 a.b) <= 35u(|a|^2 + 2|b|^2), the final addition and the two once-rounded
// norms u(|a|^2 + |b|^2 + d~), and the exact-order evaluation itself 35u d; rounded up to the
// constants below (a.b <= (|a|^2+|b|^2)/2, d <= d~ + eps).
__global__ __launch_bounds__(256) void k_nn_mfma_finish(const NnPartial* __restrict__ partial, int nsplit, int nq,
                                                        const float* __restrict__ qnorm,
                                                        const u32* __restrict__ base_max_norm_bits,
                                                        u64* __restrict__ best, int* __restrict__ recheck_rows,
                                                        float* __restrict__ recheck_thr,
                                                        int* __restrict__ recheck_count) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  float b1 = INFINITY, b2 = INFINITY;
  int i1 = -1;
  for (int sidx = 0; sidx < nsplit; ++sidx) {
    const NnPartial p = partial[(size_t)q * nsplit + sidx];
    const bool take = (p.b1 < b1) || (p.b1 == b1 && p.i1 >= 0 && (i1 < 0 || p.i1 < i1));
    const float nb2 = fminf(fminf(b2, p.b2), take ? b1 : p.b1);
    b1 = take ? p.b1 : b1;
    i1 = take ? p.i1 : i1;
    b2 = nb2;
  }
  const float na = qnorm[q], nbmax = __uint_as_float(*base_max_norm_bits);
  const float u = 5.9604645e-08f;
  const float dmax = fmaxf(na + b1, 0.f) + 1.0f;  // d~ of the leader (|a|^2 is not inside b1)
  // + 32u(|a|^2 + 2|b|^2): the index bits that replace the four low mantissa bits of every candidate
  const float eps = u * (72.0f * na + 140.0f * nbmax + 40.0f * dmax) * 1.01f;
  if (i1 >= 0 && b2 - b1 > 2.0f * eps) {
    best[q] = (u64)(u32)i1;
  } else {
    best[q] = ~0ULL;
    const int slot = atomicAdd(recheck_count, 1);
    recheck_rows[slot] = q;
    // a base row whose approximate distance exceeds this cannot be the exact arg-min (both values are within eps
    // of the exact one); +inf when the slice merge found nothing
    recheck_thr[slot] = (i1 >= 0) ? b1 + 2.0f * eps : INFINITY;
  }
}

// exact re-decision of the listed rows (list length read on the device).  One WAVEFRONT per (listed row, base
// slice): the query row is wave-uniform, so its 33 values and their -2x counterparts (from the k-major query
// table) live in SGPRs and every lane scans base descriptors (34 coalesced loads each from the k-major base
// table) with ~60 VGPRs — many resident waves hide the load latency, there is no LDS staging and no barrier.
// The test is two-staged: a 33-term fma chain gives the approximate distance (within eps of the exact-order
// value, like the MFMA result), and only base rows under the row's threshold (approximate best + 2 eps, from the
// finish kernel) can be the exact arg-min — those few are evaluated with the exact flann::L2 arithmetic.  The
// exact winner always passes the filter, so the tables are unchanged; the VALU work per (row, base) pair drops
// from ~100 to ~35 operations.  A wave arg-min feeds one packed 64-bit atomicMin per (row, slice).
// (Tried and slower: eight rows per workgroup sharing the base loads through LDS-resident queries — 256 VGPRs +
// spills; 8-byte base loads — 34.6 us against 21.4.)
__global__ __launch_bounds__(256) void k_nn_exact_rows(const float* __restrict__ A, const float* __restrict__ QT,
                                                       int nq_pad, const float* __restrict__ BT, int nB, int nb_pad,
                                                       u64* __restrict__ best, const int* __restrict__ rows,
                                                       const float* __restrict__ thr, const int* __restrict__ nrows_p) {
  const int nrows = *nrows_p;
  const int per = (nB + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nB, b0 + per);
  const int lane = qk_lane(), wave = threadIdx.x >> 6;
  for (int ri = blockIdx.x * 4 + wave; ri < nrows; ri += gridDim.x * 4) {
    const int a_idx = __builtin_amdgcn_readfirstlane(rows[ri]);
    const float tr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(thr[ri])));
    float a[33], m2a[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) {
      a[k] = A[(size_t)a_idx * 33 + k];                // uniform address -> scalar loads
      m2a[k] = QT[(size_t)k * nq_pad + a_idx];         // -2 * a[k]
    }
    u64 mine = ~0ULL;
    for (int b = b0 + lane; b < b1; b += 64) {
      float v[34];
#pragma unroll
      for (int k = 0; k < 34; ++k) v[k] = BT[(size_t)k * nb_pad + b];
      float approx = v[33];  // |b|^2
#pragma unroll
      for (int k = 0; k < 33; ++k) approx = fmaf(m2a[k], v[k], approx);
      if (approx <= tr) {  // rare: exact flann::L2 order
        float result = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const float d0 = a[4 * g] - v[4 * g], d1 = a[4 * g + 1] - v[4 * g + 1], d2 = a[4 * g + 2] - v[4 * g + 2],
                      d3 = a[4 * g + 3] - v[4 * g + 3];
          result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        const float dt = a[32] - v[32];
        result += dt * dt;
        const u64 key = ((u64)__float_as_uint(result) << 32) | (u32)b;
        mine = key < mine ? key : mine;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const u64 o = __shfl_xor(mine, off, 64);
      mine = o < mine ? o : mine;
    }
    if (lane == 0 && mine != ~0ULL) atomicMin(&best[a_idx], mine);
  }
}

// one launch instead of six memsets/fills: counters, norm maxima, tuple-test flags, source->target table
__global__ __launch_bounds__(256) void k_match_init(int* __restrict__ mcounts, int swapped, u32* __restrict__ mx0,
                                                    u32* __restrict__ mx1, int* __restrict__ passed, int n_passed,
                                                    int passed_value, int* __restrict__ tgt_of_src, int ns,
                                                    u64* __restrict__ best_small, int n_small,
                                                    u64* __restrict__ best_large, int n_large, int fill_best) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
  if (gid < 16) mcounts[gid] = (gid == MC_SWAPPED) ? swapped : 0;
  if (gid == 16) *mx0 = 0u;
  if (gid == 17) *mx1 = 0u;
  for (int i = gid; i < n_passed; i += gsz) passed[i] = passed_value;
  for (int i = gid; i < ns; i += gsz) tgt_of_src[i] = -1;
  if (fill_best) {
    for (int i = gid; i < n_small; i += gsz) best_small[i] = ~0ULL;
    for (int i = gid; i < n_large; i += gsz) best_large[i] = ~0ULL;
  }
}

// unpack both NN tables and evaluate the mutual-NN test in one pass
__global__ __launch_bounds__(256) void k_cross_flags2(const u64* __restrict__ best_large,
                                                      const u64* __restrict__ best_small, int n_large, int n_small,
                                                      int* __restrict__ nn_of_large, int* __restrict__ nn_of_small,
                                                      int* __restrict__ flags) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
  for (int j = gid; j < n_small; j += gsz) {
    const u64 b = best_small[j];
    nn_of_small[j] = (b == ~0ULL) ? 0 : (int)(u32)b;
  }
  for (int i = gid; i < n_large; i += gsz) {
    const u64 b = best_large[i];
    const int j = (b == ~0ULL) ? 0 : (int)(u32)b;
    nn_of_large[i] = j;
    const u64 bs = best_small[j];
    const int back = (bs == ~0ULL) ? 0 : (int)(u32)bs;
    flags[i] = (back == i) ? 1 : 0;
  }
}

// exclusive scan of the predicate (in[i] >= 0), one workgroup; out has n+1 entries
__global__ __launch_bounds__(1024) void k_scan_nonneg(const int* __restrict__ in, int* __restrict__ out, int n) {
  d_block_scan(in, out, n, [](int x) { return x >= 0 ? 1 : 0; });
}

__global__ void k_corr_compact2(const int* __restrict__ scan, const int* __restrict__ tgt_of_src, int ns,
                                int* __restrict__ corr, int* __restrict__ mcounts, int* __restrict__ mail,
                                const int* __restrict__ counts0, const int* __restrict__ counts1, int seq) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < ns; s += gridDim.x * blockDim.x) {
    const int t = tgt_of_src[s];
    if (t >= 0) {
      corr[2 * scan[s]] = s;
      corr[2 * scan[s] + 1] = t;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) mcounts[MC_NCORR] = scan[ns];
  if (mail && blockIdx.x == 0 && threadIdx.x < 48) {  // last matcher kernel: counters for the host, no copy launches
    const int t = threadIdx.x;
    if (t < 16)
      mail_store_line(mail + MAIL_MATCH, t, (t == MC_NCORR) ? scan[ns] : mcounts[t], seq);
    else if (t < 32)
      mail_store_line(mail + MAIL_CNT0, t - 16, counts0[t - 16], seq);
    else
      mail_store_line(mail + MAIL_CNT1, t - 32, counts1[t - 32], seq);
    __threadfence_system();  // threads 0..47 are one wavefront: the stores above are acknowledged before ...
    if (t == 0) mail[MAIL_SEQ_MATCH] = seq;  // ... the word the host is watching changes
  }
}


__global__ void k_cross_compact(const int* __restrict__ flags, const int* __restrict__ scan,
                                const int* __restrict__ nn_of_large, int n_large, int* __restrict__ cross_i,
                                int* __restrict__ cross_j, int* __restrict__ mcounts) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_large; i += gridDim.x * blockDim.x) {
    if (flags[i]) {
      cross_i[scan[i]] = i;
      cross_j[scan[i]] = nn_of_large[i];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) mcounts[MC_NCROSS] = scan[n_large];
}

// tuple test (reference feature_matcher.cc:187-247); trial t draws qm_rand_u32(seed, 3t+k) % ncorr
__global__ __launch_bounds__(256) void k_tuple(const float4* __restrict__ pts_i, const float* __restrict__ mean_i,
                                               const float4* __restrict__ pts_j, const float* __restrict__ mean_j,
                                               const int* __restrict__ cross_i, const int* __restrict__ cross_j,
                                               const int* __restrict__ mcounts, float scale, u64 seed,
                                               int* __restrict__ passed) {
  const int ncorr = mcounts[MC_NCROSS];
  if (ncorr <= 0) return;
  const long long trials = (long long)ncorr * 100;
  const float mix = mean_i[0], miy = mean_i[1], miz = mean_i[2];
  const float mjx = mean_j[0], mjy = mean_j[1], mjz = mean_j[2];
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < trials;
       t += (long long)gridDim.x * blockDim.x) {
    const int r0 = (int)(qm_rand_u32(seed, 3ULL * (u64)t) % (u32)ncorr);
    const int r1 = (int)(qm_rand_u32(seed, 3ULL * (u64)t + 1ULL) % (u32)ncorr);
    const int r2 = (int)(qm_rand_u32(seed, 3ULL * (u64)t + 2ULL) % (u32)ncorr);
    float pi[3][3], pj[3][3];
    const int rr[3] = {r0, r1, r2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 a = pts_i[cross_i[rr[k]]], b = pts_j[cross_j[rr[k]]];
      pi[k][0] = a.x - mix;
      pi[k][1] = a.y - miy;
      pi[k][2] = a.z - miz;
      pj[k][0] = b.x - mjx;
      pj[k][1] = b.y - mjy;
      pj[k][2] = b.z - mjz;
    }
    float li[3], lj[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int k2 = (k + 1) % 3;
      // edges: (0,1), (1,2), (2,0)
      float dx = pi[k][0] - pi[k2][0], dy = pi[k][1] - pi[k2][1], dz = pi[k][2] - pi[k2][2];
      li[k] = sqrtf(dx * dx + (dy * dy + dz * dz));
      dx = pj[k][0] - pj[k2][0];
      dy = pj[k][1] - pj[k2][1];
      dz = pj[k][2] - pj[k2][2];
      lj[k] = sqrtf(dx * dx + (dy * dy + dz * dz));
    }
    if ((li[0] * scale < lj[0]) && (lj[0] < li[0] / scale) && (li[1] * scale < lj[1]) && (lj[1] < li[1] / scale) &&
        (li[2] * scale < lj[2]) && (lj[2] < li[2] / scale)) {
      passed[r0] = 1;
      passed[r1] = 1;
      passed[r2] = 1;
    }
  }
}

// passed cross pairs -> tgt_of_src (each source index occurs at most once after the cross-check)
__global__ void k_scatter_pairs(const int* __restrict__ cross_i, const int* __restrict__ cross_j,
                                const int* __restrict__ passed, const int* mcounts, int swapped,
                                int* __restrict__ tgt_of_src, int* mc_out) {
  const int nc = mcounts[MC_NCROSS];
  int local = 0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    if (passed[c]) {
      const int i = cross_i[c], j = cross_j[c];
      const int s = swapped ? j : i, t = swapped ? i : j;
      tgt_of_src[s] = t;
      ++local;
    }
  }
  local = wave_sum_i32(local);
  if (qk_lane() == 0 && local) atomicAdd(&mc_out[MC_NTUPLE], local);
}

// ---- fused tails for clouds of up to 16384 points: one workgroup of 1024 threads, every thread owning one
// contiguous run of at most 16 indices, so flag -> exclusive scan -> compaction happens in registers and LDS
// without the three-launch (flags, scan, compact) round trips.
// K6: unpack both NN tables, mutual-NN test, cross pairs in ascending i.
__global__ __launch_bounds__(1024) void k_cross_fused(const u64* __restrict__ best_large,
                                                      const u64* __restrict__ best_small, int n_large, int n_small,
                                                      int* __restrict__ nn_of_large, int* __restrict__ nn_of_small,
                                                      int* __restrict__ cross_i, int* __restrict__ cross_j,
                                                      int* __restrict__ mcounts) {
  extern __shared__ int fl_s[];  // [n_large] nn index | keep flag << 31, staged with coalesced (striped) accesses
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < n_small; j += 1024) {
    const u64 b = best_small[j];
    nn_of_small[j] = (b == ~0ULL) ? 0 : (int)(u32)b;
  }
  for (int i = tid; i < n_large; i += 1024) {
    const u64 b = best_large[i];
    const int j = (b == ~0ULL) ? 0 : (int)(u32)b;
    nn_of_large[i] = j;
    const u64 bs = best_small[j];
    const int back = (bs == ~0ULL) ? 0 : (int)(u32)bs;
    fl_s[i] = j | ((back == i) ? (int)0x80000000 : 0);
  }
  __syncthreads();
  const int K = (n_large + 1023) >> 10, base = tid * K;
  int jj[16];
  u32 keep = 0;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = base + k;
    jj[k] = 0;
    if (k < K && i < n_large) {
      const int v = fl_s[i];
      jj[k] = v & 0x7fffffff;
      if (v < 0) {
        keep |= 1u << k;
        ++cnt;
      }
    }
  }
  int tot;
  const int ex = wave_excl_scan_i32(cnt, &tot);
  if (lane == 63) wsum[wave] = tot;
  __syncthreads();
  int run = ex, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    run += (w < wave) ? wsum[w] : 0;
    total += wsum[w];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if ((keep >> k) & 1u) {
      cross_i[run] = base + k;
      cross_j[run] = jj[k];
      ++run;
    }
  if (tid == 0) mcounts[MC_NCROSS] = total;
}

// K8 + gather: passed cross pairs -> tgt_of_src, compaction in source order, the matched keypoint clouds
// (when asked for) and the counters for the host.
__global__ __launch_bounds__(1024) void k_pairs_fused(const int* __restrict__ cross_i, const int* __restrict__ cross_j,
                                                      const int* __restrict__ passed, int swapped, int ns,
                                                      int* __restrict__ tgt_of_src, int* __restrict__ corr,
                                                      const float4* __restrict__ vs, const float4* __restrict__ vt,
                                                      float4* __restrict__ m_src, float4* __restrict__ m_tgt,
                                                      int m_cap, int* __restrict__ mcounts, int* __restrict__ mail,
                                                      const int* __restrict__ counts0, const int* __restrict__ counts1,
                                                      int seq) {
  __shared__ int wsum[16];
  __shared__ int s_ntuple;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_ntuple = 0;
  __syncthreads();
  const int nc = mcounts[MC_NCROSS];
  int local = 0;
  for (int c = tid; c < nc; c += 1024)
    if (passed[c]) {
      const int i = cross_i[c], j = cross_j[c];
      tgt_of_src[swapped ? j : i] = swapped ? i : j;
      ++local;
    }
  local = wave_sum_i32(local);
  if (lane == 0 && local) atomicAdd(&s_ntuple, local);
  __threadfence_block();
  __syncthreads();  // the scattered targets are visible to the whole workgroup
  extern __shared__ int tg_s[];  // [ns] staged with coalesced (striped) loads
  for (int i = tid; i < ns; i += 1024) tg_s[i] = tgt_of_src[i];
  __syncthreads();
  const int K = (ns + 1023) >> 10, base = tid * K;
  int tt[16];
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int sidx = base + k;
    tt[k] = (k < K && sidx < ns) ? tg_s[sidx] : -1;
    cnt += tt[k] >= 0 ? 1 : 0;
  }
  int tot;
  const int ex = wave_excl_scan_i32(cnt, &tot);
  if (lane == 63) wsum[wave] = tot;
  __syncthreads();
  int run = ex, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    run += (w < wave) ? wsum[w] : 0;
    total += wsum[w];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (tt[k] >= 0) {
      corr[2 * run] = base + k;
      corr[2 * run + 1] = tt[k];
      if (m_src && run < m_cap) {  // the count is still reported: the host raises QTR_ERR_CAPACITY past m_cap
        float4 a = vs[base + k], b = vt[tt[k]];
        a.w = 0.f;
        b.w = 0.f;
        m_src[run] = a;
        m_tgt[run] = b;
      }
      ++run;
    }
  if (tid == 0) {
    mcounts[MC_NCORR] = total;
    mcounts[MC_NTUPLE] = s_ntuple;
  }
  if (mail && tid < 48) {  // last matcher kernel: counters for the host, no copy launches
    const int t = tid;
    if (t < 16)
      mail_store_line(mail + MAIL_MATCH, t, (t == MC_NCORR) ? total : (t == MC_NTUPLE) ? s_ntuple : mcounts[t], seq);
    else if (t < 32)
      mail_store_line(mail + MAIL_CNT0, t - 16, counts0[t - 16], seq);
    else
      mail_store_line(mail + MAIL_CNT1, t - 32, counts1[t - 32], seq);
    __threadfence_system();  // threads 0..47 are one wavefront: the stores above are acknowledged before ...
    if (t == 0) mail[MAIL_SEQ_MATCH] = seq;  // ... the word the host is watching changes
  }
}

__global__ void k_gather_matched(const float4* __restrict__ vs, const float4* __restrict__ vt,
                                 const int* __restrict__ corr, int L, float4* __restrict__ ms, float4* __restrict__ mt) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < L; c += gridDim.x * blockDim.x) {
    float4 a = vs[corr[2 * c]], b = vt[corr[2 * c + 1]];
    a.w = 0.f;
    b.w = 0.f;
    ms[c] = a;
    mt[c] = b;
  }
}

static inline int grid_for(int n) {
  int g = (n + 255) / 256;
  return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}

hipError_t match_enqueue(FrontBufs& F, int ns, int nt, const qtr_frontend_params& fp, hipStream_t st) {
  (void)hipGetLastError();
  // fi = larger cloud, fj = smaller (reference feature_matcher.cc:84-92)
  const int swapped = nt > ns ? 1 : 0;
  CloudBufs& Ci = F.cloud[swapped ? 1 : 0];
  CloudBufs& Cj = F.cloud[swapped ? 0 : 1];
  const int n_large = swapped ? nt : ns, n_small = swapped ? ns : nt;
  const int maxc = n_small;  // cross-checked pairs <= n_small
  const bool tuple = fp.use_tuple_test && fp.tuple_scale != 0;
  hipLaunchKernelGGL(k_match_init, dim3(grid_for(n_large)), dim3(256), 0, st, F.mcounts, swapped, Ci.max_norm,
                     Cj.max_norm, F.passed, maxc, tuple ? 0 : 1, F.tgt_of_src, ns, F.best_small, n_small, F.best_large,
                     n_large, F.nn_engine == 0 ? 1 : 0);
  // K5: NN of every small-cloud descriptor in the large cloud, and of every large-cloud descriptor in
  // the small cloud (the reference queries the latter lazily for hit rows only; the mutual test below
  // only ever reads hit rows, so the result is the same)
  if (F.nn_engine == 0) {
    auto nsplit = [](int nq, int nb) {
      int blocks_x = (nq + 255) / 256;
      int s = (1024 + blocks_x - 1) / blocks_x;
      int maxs = (nb + NN_TILE - 1) / NN_TILE;
      if (s > maxs) s = maxs;
      if (s < 1) s = 1;
      if (s > 256) s = 256;
      return s;
    };
    if (F.ev_nn[0]) (void)hipEventRecord(F.ev_nn[0], st);
    hipLaunchKernelGGL(k_nn_exact, dim3((n_small + 255) / 256, nsplit(n_small, n_large)), dim3(256), 0, st, Cj.fpfh,
                       n_small, Ci.fpfh, n_large, F.best_small, (const int*)nullptr, 0);
    if (F.ev_nn[1]) (void)hipEventRecord(F.ev_nn[1], st);
    if (F.ev_nn[2]) (void)hipEventRecord(F.ev_nn[2], st);
    hipLaunchKernelGGL(k_nn_exact, dim3((n_large + 255) / 256, nsplit(n_large, n_small)), dim3(256), 0, st, Ci.fpfh,
                       n_large, Cj.fpfh, n_small, F.best_large, (const int*)nullptr, 0);
    if (F.ev_nn[3]) (void)hipEventRecord(F.ev_nn[3], st);
  } else {
    const int pad_small = (n_small + NN_QPB - 1) / NN_QPB * NN_QPB, pad_large = (n_large + NN_QPB - 1) / NN_QPB * NN_QPB;
    hipLaunchKernelGGL(k_desc_prep2, dim3(pad_large / 256, 2), dim3(256), 0, st, Ci.fpfh, n_large, pad_large, Ci.baseT,
                       Ci.queryT, Ci.norms, Ci.max_norm, Cj.fpfh, n_small, pad_small, Cj.baseT, Cj.queryT, Cj.norms,
                       Cj.max_norm);
    auto run_dir = [&](CloudBufs& Q, int nq, int nq_pad, CloudBufs& Bc, int nb, int nb_pad, u64* best, int mc_slot,
                       hipEvent_t ev0, hipEvent_t ev1) {
      const int ntiles = nb_pad / 32;
      // Slicing policy: ONE workgroup per compute unit and no more workgroups than compute units.  Measured
      // on MI355X (tests/probe/nn_probe.hip): a lone workgroup runs a tile in ~6.0k clocks (68 MFMAs = 4.35k);
      // two per CU share the matrix pipe AND lose ~40 % to each other's loads, and a grid of 270 workgroups on
      // 256 CUs runs as long as 512 would (the doubled-up CUs finish last).
      const int qblocks = nq_pad / NN_QPB;
      int ns_ = 1, tps = ntiles;
      if (F.nn_target_waves > 0) {  // QTR_NN_WAVES: aim at a wave count (experiments)
        ns_ = (F.nn_target_waves + nq_pad / NN_QPW - 1) / (nq_pad / NN_QPW);
        if (ns_ > 32) ns_ = 32;
        if (ns_ > ntiles) ns_ = ntiles;
        if (ns_ < 1) ns_ = 1;
        tps = (ntiles + ns_ - 1) / ns_;
        ns_ = (ntiles + tps - 1) / tps;
      } else {  // fewest (rounds of n_cu workgroups) x (tiles per workgroup + ~1.5 tiles of prologue / epilogue)
        double best_cost = 1e300;
        for (int cand = 1; cand <= 32 && cand <= ntiles; ++cand) {
          const int ctps = (ntiles + cand - 1) / cand, cns = (ntiles + ctps - 1) / ctps;
          const int rounds = (qblocks * cns + F.n_cu - 1) / F.n_cu;
          const double cost = (double)rounds * (ctps + 1.5);
          if (cost < best_cost) {
            best_cost = cost;
            ns_ = cns;
            tps = ctps;
          }
        }
      }
      if (ev0) (void)hipEventRecord(ev0, st);
      hipLaunchKernelGGL(k_nn_mfma, dim3(nq_pad / NN_QPB, ns_), dim3(256), 0, st, Bc.baseT, nb_pad, Q.queryT, nq_pad, tps,
                         (NnPartial*)F.nn_partial, (F.nn_trace && mc_slot == MC_RECHECK0) ? F.mcounts : (int*)nullptr);
      if (ev1) (void)hipEventRecord(ev1, st);
      hipLaunchKernelGGL(k_nn_mfma_finish, dim3((nq + 255) / 256), dim3(256), 0, st, (const NnPartial*)F.nn_partial, ns_,
                         nq, Q.norms, Bc.max_norm, best, F.recheck_rows, F.recheck_thr, F.mcounts + mc_slot);
      int ey = (nb + 255) / 256;  // ~4 base descriptors per lane and slice
      if (ey > 64) ey = 64;
      if (ey < 1) ey = 1;
      hipLaunchKernelGGL(k_nn_exact_rows, dim3(64, ey), dim3(256), 0, st, Q.fpfh, Q.queryT, nq_pad, Bc.baseT, nb, nb_pad, best,
                         F.recheck_rows, F.recheck_thr, F.mcounts + mc_slot);
    };
    const bool evs = F.nn_events != 0;
    run_dir(Cj, n_small, pad_small, Ci, n_large, pad_large, F.best_small, MC_RECHECK0, evs ? F.ev_nn[0] : nullptr,
            evs ? F.ev_nn[1] : nullptr);
    run_dir(Ci, n_large, pad_large, Cj, n_small, pad_small, F.best_large, MC_RECHECK1, evs ? F.ev_nn[2] : nullptr,
            evs ? F.ev_nn[3] : nullptr);
  }
  // K6 cross-check -> pairs in ascending i
  hipError_t e;
  const bool fused_tail = n_large <= 16384 && ns <= 16384;
  F.gathered = false;
  if (fused_tail) {
    hipLaunchKernelGGL(k_cross_fused, dim3(1), dim3(1024), (size_t)n_large * 4, st, F.best_large, F.best_small, n_large, n_small,
                       F.nn_of_large, F.nn_of_small, F.cross_i, F.cross_j, F.mcounts);
  } else {
    hipLaunchKernelGGL(k_cross_flags2, dim3(grid_for(n_large)), dim3(256), 0, st, F.best_large, F.best_small, n_large,
                       n_small, F.nn_of_large, F.nn_of_small, F.flags);
    if ((e = exclusive_scan_i32(F.flags, F.scan, n_large, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_cross_compact, dim3(grid_for(n_large)), dim3(256), 0, st, F.flags, F.scan, F.nn_of_large,
                       n_large, F.cross_i, F.cross_j, F.mcounts);
  }
  // K7 tuple test
  if (tuple)
    hipLaunchKernelGGL(k_tuple, dim3(2048), dim3(256), 0, st, Ci.vox, Ci.mean, Cj.vox, Cj.mean, F.cross_i, F.cross_j,
                       F.mcounts, fp.tuple_scale, (u64)fp.seed, F.passed);
  // K8 un-swap, sort by (src, tgt), unique  ==  compaction in source-index order
  if (fused_tail) {
    hipLaunchKernelGGL(k_pairs_fused, dim3(1), dim3(1024), (size_t)ns * 4, st, F.cross_i, F.cross_j, F.passed, swapped, ns,
                       F.tgt_of_src, F.corr, F.cloud[0].vox, F.cloud[1].vox, F.m_src, F.m_tgt, F.m_cap, F.mcounts, F.mail,
                       F.cloud[0].counts, F.cloud[1].counts, F.mail_seq);
    F.gathered = F.m_src != nullptr;
  } else {
    hipLaunchKernelGGL(k_scatter_pairs, dim3(grid_for(maxc)), dim3(256), 0, st, F.cross_i, F.cross_j, F.passed,
                       F.mcounts, swapped, F.tgt_of_src, F.mcounts);
    hipLaunchKernelGGL(k_scan_nonneg, dim3(1), dim3(1024), 0, st, F.tgt_of_src, F.scan, ns);
    hipLaunchKernelGGL(k_corr_compact2, dim3(grid_for(ns)), dim3(256), 0, st, F.scan, F.tgt_of_src, ns, F.corr,
                       F.mcounts, F.mail, F.cloud[0].counts, F.cloud[1].counts, F.mail_seq);
  }
  return hipGetLastError();
}

hipError_t gather_matched_enqueue(FrontBufs& F, int L, float4* m_src, float4* m_tgt, hipStream_t st) {
  if (L <= 0 || (F.gathered && m_src == F.m_src && m_tgt == F.m_tgt)) return hipSuccess;  // k_pairs_fused did it
  hipLaunchKernelGGL(k_gather_matched, dim3(grid_for(L)), dim3(256), 0, st, F.cloud[0].vox, F.cloud[1].vox, F.corr, L,
                     m_src, m_tgt);
  return hipGetLastError();
}

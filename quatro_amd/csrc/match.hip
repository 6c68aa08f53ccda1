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
//
// Like the reference (feature_matcher.cc:113-122) the second direction is only asked for the rows of the larger
// cloud that some row of the smaller cloud points at ("hit" rows; ~40 % of the cloud on lidar scans): after the
// first direction is final the hit rows are compacted on the device and their query columns gathered.
//
// Every kernel takes the views of the pairs it serves (MatchArgs) and picks its pair with blockIdx.z: one pair for
// qtr_match / qtr_register_pair, a whole group for qtr_submit_batch.  Sizes that are only known on the device (hit
// count, re-check list length) are read there; grids are sized for the worst case and surplus workgroups exit.
#include <vector>

#include "common.h"
#include "frontend.h"

#define NN_TILE 64
#define CROSS_LDS_BYTES (156 * 1024)  // dynamic LDS k_cross_fused may ask for (160 KB per CU, minus its static part)
#define NN_STRIDE 36  // floats per staged descriptor row (33 + 3 zero pad; 16-byte aligned rows)

// EXT (view in the kernel arguments / in device memory) is a template parameter, not a run-time select: a reference that
// may point into either is a generic pointer, and every load behind it degrades to flat_load.  The pick itself is
// written inline in every kernel — routed through a helper that takes the argument structs by reference, the compiler
// loses the kernel-argument provenance again (see ViewExt in common.h).
#define LAUNCH_MV_K(kern, K, a, grid, block, lds, st, ...)                                                       \
  do {                                                                                                          \
    if ((a).ext)                                                                                                \
      hipLaunchKernelGGL((kern<true, K>), grid, block, lds, st, (ViewExt<MatchView>{(a).ext, {0, 0, 0}}), (a).one, ##__VA_ARGS__); \
    else                                                                                                        \
      hipLaunchKernelGGL((kern<false, K>), grid, block, lds, st, (ViewExt<MatchView>{nullptr, {0, 0, 0}}), (a).one, ##__VA_ARGS__); \
  } while (0)
// the same with a pair of timing events attached to the dispatch itself (hipExtLaunchKernelGGL: the events take the
// kernel's own start / end timestamps) — two hipEventRecord calls around a launch put two barrier packets into the
// queue, ~5.7 us each on this chain; e0 / e1 null: plain launch
#define LAUNCH_MV_EV(kern, a, grid, block, lds, st, e0, e1, ...)                                                 \
  do {                                                                                                          \
    if ((a).ext)                                                                                                \
      hipExtLaunchKernelGGL((kern<true>), grid, block, lds, st, e0, e1, 0, (ViewExt<MatchView>{(a).ext, {0, 0, 0}}), (a).one, ##__VA_ARGS__);  \
    else                                                                                                        \
      hipExtLaunchKernelGGL((kern<false>), grid, block, lds, st, e0, e1, 0, (ViewExt<MatchView>{nullptr, {0, 0, 0}}), (a).one, ##__VA_ARGS__); \
  } while (0)
#define LAUNCH_MV(kern, a, grid, block, lds, st, ...)                                       \
  do {                                                                                      \
    if ((a).ext)                                                                            \
      hipLaunchKernelGGL((kern<true>), grid, block, lds, st, (ViewExt<MatchView>{(a).ext, {0, 0, 0}}), (a).one, ##__VA_ARGS__);             \
    else                                                                                    \
      hipLaunchKernelGGL((kern<false>), grid, block, lds, st, (ViewExt<MatchView>{nullptr, {0, 0, 0}}), (a).one, ##__VA_ARGS__);            \
  } while (0)

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

// All-exact engine (QTR_NN_ENGINE=exact).  grid (ceil(nq_max/256), slices, pairs): thread = one query, blockIdx.y =
// slice of the base cloud.  Direction 1 evaluates the hit rows only (rows = hit list, count on the device).
template <bool EXT>
__global__ __launch_bounds__(256) void k_nn_exact(ViewExt<MatchView> x, MatchView one, int dir) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const NnDir& D = V.d[dir];
  __shared__ __attribute__((aligned(16))) float tile[NN_TILE * NN_STRIDE];
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int nq = V.mcounts[D.nq_slot];
  if (blockIdx.x * 256 >= nq) return;
  const int* rows = dir ? V.hit_rows : nullptr;
  const int a_idx = (q < nq) ? (rows ? rows[q] : q) : -1;
  const float* A = D.A;
  const float* B = dir ? V.fpfh_j : V.fpfh_i;
  const int nB = D.nb;
  float av[33];
#pragma unroll
  for (int t = 0; t < 33; ++t) av[t] = (a_idx >= 0) ? A[(size_t)a_idx * 33 + t] : 0.f;
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
        const float d = l2_flann33(av, tile + r * NN_STRIDE);
        if (d < bd) {
          bd = d;
          bi = base + r;
        }
      }
    }
  }
  if (a_idx >= 0 && bi >= 0) atomicMin(&D.best[a_idx], ((u64)__float_as_uint(bd) << 32) | (u32)bi);
}

// =================================================================================================
// MFMA engine.  d~(a,b) = |a|^2 + (|b|^2 - 2 a.b): the bracket is ONE f32 MFMA chain over K = 34
// (33 descriptor bins + one slot carrying |b|^2 against a constant 1), issued as 17 x
// v_mfma_f32_32x32x2_f32.  The streamed operand is the BASE cloud (M side, 32 rows per tile), the
// stationary operand the QUERIES (N side): in the 32x32 accumulator layout a lane owns ONE query column
// (col = lane & 31) and 16 base rows, so the running best / second-best per query live in that lane's
// registers and no cross-lane reduction is needed until the very end.  Descriptors are pre-transposed
// to k-major ([34][n_pad]) so every fragment load is two coalesced 128-byte segments.
// A query's approximate winner is accepted only when second-best - best exceeds a rigorous rounding bound;
// otherwise the row is re-decided by k_nn_exact_rows (bit-identical tables either way).
//
// Rounding bound.  u = 2^-24.  With nb' the norm row of the base table, the chain computes v = nb' - 2 a.b with
// |error| <= 34u (|b|^2 + 2|a||b|) (34 fused multiply-adds) + 32u (the same) for the accumulator-register index that
// replaces the four low mantissa bits <= 66u |a|^2 + 132u |b|^2.  The norm row is stored SCALED DOWN,
// nb' <= |b|^2 (1 - 141u): every tracked value is then a lower bound of the real distance up to a term that depends on
// the query alone,      d(a,b) >= |a|^2~ + v - 67u |a|^2,
// while for the winner  d(a,b1) <= |a|^2~ + v1 + 67u |a|^2 + 274u |b1|^2,
// and the exact-order float evaluation d_ex obeys |d_ex - d| <= 37u d.  Hence the approximate winner is the exact
// arg-min whenever   v2 - v1 > u (144 |a|^2 + 280 |b1|^2 + 40 (d~1 + d~2))  (k_nn_finish, 1 % slack on top).
// The bound scales with the WINNER's norm, not with the largest norm of the base cloud.
//
// Bit-identical descriptors (degenerate neighbourhoods give thousands of them on a lidar scan) tie exactly and would
// all fail that test: k_desc_dedup hides every base row whose descriptor also sits at a lower row (norm row = 1e30),
// so a tie class is represented by its lowest index — the row FLANN-order evaluation with lowest-index ties returns.
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define NN_K2 17      // k pairs: K = 34
#define NN_QPW 128    // queries per wave (4 accumulators of 32 columns)
#define NN_QPB 512    // queries per workgroup (4 waves)
#define NN_MAXSPLIT 32
#define NN_NORM_SCALE (1.0 - 141.0 * 5.9604644775390625e-08)
#define NNH_NORM_SCALE (1.0 - 56.0 * 5.9604644775390625e-08)  // the f16-split tables' own (round 5, see the f16 engine's bound)

__device__ __forceinline__ u64 desc_hash(const float* d) { return desc_hash33(d); }  // (frontend.hip: k2_fpfh computes the same)

// the 256 rows of a workgroup, row-major table -> LDS: 33 coalesced loads per thread, all in flight before the first LDS
// store (as a plain loop the compiler waits for every load: 33 round trips, 8 us of a 15 us kernel)
template <int ROWS = 256>  // (= threads of the workgroup)
__device__ __forceinline__ void stage_rows(const float* __restrict__ desc, int row0, int n, float* s_rows) {
  const size_t base = (size_t)row0 * 33, lim = (size_t)n * 33;
  float t[33];
#pragma unroll
  for (int k = 0; k < 33; ++k) {
    const size_t e = base + (size_t)(k * ROWS + threadIdx.x);
    t[k] = (e < lim) ? desc[e] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 33; ++k) s_rows[k * ROWS + threadIdx.x] = t[k];
}

// desc[n][33] -> baseT[34][n_pad] (row 33 = scaled |b|^2; pad rows get 1e30 so they never win) and
// queryT[34][n_pad] (-2 * desc, row 33 = 1); the norms (binary64 sum rounded once); and the row's entry in the
// dedup table: slot sequence from the low hash bits, tag = high 32 bits, value = lowest row with that tag.
__device__ __forceinline__ void d_desc_prep(const float* __restrict__ desc, int n, int n_pad,
                                            float* __restrict__ baseT /* null: the f16 engine builds its own tables */,
                                            float* __restrict__ queryT /* null: only the f32 MFMA engine reads it */,
                                            float* __restrict__ norms, u64* __restrict__ hashes, u64* __restrict__ table,
                                            int mask, float* s_rows /* LDS, 256 x 33 */) {
  // the rows of this workgroup, staged through LDS: coalesced loads of the row-major table (a thread reading its own
  // 132-byte row costs one cache-line access per lane and instruction), then conflict-free row reads (stride 33)
  const int row0 = blockIdx.x * 256;
  if (row0 >= n_pad) return;
  QTR_STAMP(STAMP_DESC_PREP, 0)
  stage_rows(desc, row0, n, s_rows);
  __syncthreads();
  QTR_STAMP(STAMP_DESC_PREP, 1)
  const int i = row0 + threadIdx.x;
  if (i >= n_pad) return;
  if (i < n) {
    double acc = 0.0;
    float v[33];
    for (int k = 0; k < 33; ++k) {
      v[k] = s_rows[threadIdx.x * 33 + k];
      acc += (double)v[k] * (double)v[k];
      if (baseT) baseT[(size_t)k * n_pad + i] = v[k];
      if (queryT) queryT[(size_t)k * n_pad + i] = -2.0f * v[k];
    }
    if (baseT) baseT[(size_t)33 * n_pad + i] = __double2float_rd(acc * NN_NORM_SCALE);
    if (queryT) queryT[(size_t)33 * n_pad + i] = 1.0f;
    norms[i] = (float)acc;
    const u64 h = desc_hash(v);
    hashes[i] = h;
    QTR_STAMP(STAMP_DESC_PREP, 2)
    const u64 tag = h & 0xffffffff00000000ULL;
    u32 slot = (u32)h & (u32)mask;
    for (int probe = 0; probe <= mask; ++probe) {
      // (load first: claiming the slot with the CAS straight away saves a round trip on an empty slot but was measured
      // slower — thousands of identical descriptors then queue on one word)
      u64 cur = table[slot];
      if (cur == ~0ULL) {
        const u64 old = atomicCAS(&table[slot], ~0ULL, tag | (u32)i);
        if (old == ~0ULL) break;
        cur = old;
      }
      if ((cur & 0xffffffff00000000ULL) == tag) {
        // entries only ever decrease, so a (possibly stale) value that is already <= i makes the atomic redundant:
        // rows run in roughly ascending order and thousands of identical descriptors would otherwise hammer one word
        if ((u32)cur > (u32)i) atomicMin(&table[slot], tag | (u32)i);
        break;
      }
      slot = (slot + 1) & (u32)mask;
    }
    QTR_STAMP(STAMP_DESC_PREP, 3)
  } else if (baseT) {
    for (int k = 0; k < 33; ++k) {
      baseT[(size_t)k * n_pad + i] = 0.f;
      if (queryT) queryT[(size_t)k * n_pad + i] = 0.f;
    }
    baseT[(size_t)33 * n_pad + i] = 1e30f;
    if (queryT) queryT[(size_t)33 * n_pad + i] = 1.0f;
  }
}
// grid (pad_large_max / 256, 2, pairs): blockIdx.y = cloud (0: larger, 1: smaller)
// tables: 0 none (f16 engine: norms, hashes and the dedup table only), 1 baseT (exact engine), 2 baseT + queryT (f32 MFMA)
template <bool EXT>
__global__ __launch_bounds__(256) void k_desc_prep(ViewExt<MatchView> x, MatchView one, int tables) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  __shared__ float s_rows[256 * 33];
  if (blockIdx.y == 0)
    d_desc_prep(V.fpfh_i, V.n_large, V.pad_large, tables >= 1 ? V.baseT_i : nullptr, tables >= 2 ? V.queryT_i : nullptr, V.norms_i,
                V.hash_i, V.table_i, V.dd_mask, s_rows);
  else
    d_desc_prep(V.fpfh_j, V.n_small, V.pad_small, tables >= 1 ? V.baseT_j : nullptr, tables >= 2 ? V.queryT_j : nullptr, V.norms_j,
                V.hash_j, V.table_j, V.dd_mask, s_rows);
}

// Hides base rows that duplicate a lower row bit for bit (see the header comment).  The table gives the lowest row
// with the same hash tag; the 33 values are compared before a row is hidden, so a hash collision only costs a missed
// merge.
__device__ __forceinline__ void d_desc_dedup(const float* __restrict__ desc, int n, int n_pad, float* __restrict__ baseT,
                                             const u64* __restrict__ hashes, const u64* __restrict__ table, int mask,
                                             int* __restrict__ hidden_count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool hide = false;
  if (i < n) {
    const u64 h = hashes[i];
    const u64 tag = h & 0xffffffff00000000ULL;
    u32 slot = (u32)h & (u32)mask;
    int rep = i;
    for (int probe = 0; probe <= mask; ++probe) {
      const u64 cur = table[slot];
      if (cur == ~0ULL) break;
      if ((cur & 0xffffffff00000000ULL) == tag) {
        rep = (int)(u32)cur;
        break;
      }
      slot = (slot + 1) & (u32)mask;
    }
    if (rep < i) {
      hide = true;
      for (int k = 0; k < 33; ++k)
        hide = hide && (__float_as_uint(desc[(size_t)i * 33 + k]) == __float_as_uint(desc[(size_t)rep * 33 + k]));
      if (hide) baseT[(size_t)33 * n_pad + i] = 1e30f;
    }
  }
  const u64 bal = __ballot(hide);
  if (qk_lane() == 0 && bal) atomicAdd(hidden_count, __popcll(bal));
}
template <bool EXT>
__global__ __launch_bounds__(256) void k_desc_dedup(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  if (blockIdx.y == 0)
    d_desc_dedup(V.fpfh_i, V.n_large, V.pad_large, V.baseT_i, V.hash_i, V.table_i, V.dd_mask, V.mcounts + MC_HIDDEN_I);
  else
    d_desc_dedup(V.fpfh_j, V.n_small, V.pad_small, V.baseT_j, V.hash_j, V.table_j, V.dd_mask, V.mcounts + MC_HIDDEN_J);
}

// Norm-bin order of a cloud's descriptors for the exact re-check.  d(a,b) >= (|a| - |b|)^2, so a base row can only be
// the arg-min of a listed query if sqrt|b|^2 lies within sqrt(d_best + slack) of sqrt|a|^2: with the base table's columns
// sorted by bin of sqrt|b|^2 a listed row scans one contiguous span instead of the whole cloud (~6 % of it on lidar
// scans: most listed rows are near-degenerate descriptors that only compete with each other).
// k_norm_bins: one workgroup per cloud — histogram, scan, scatter (order inside a bin is arbitrary; the arg-min key carries
// the row).  k_norm_gather: the permuted copy of the base table (after k_desc_dedup: hidden rows keep their 1e30).
__device__ __forceinline__ int norm_bin(float nrm) { return min(NORM_BINS - 1, max(0, (int)floorf(sqrtf(fmaxf(nrm, 0.f))))); }
__device__ __forceinline__ void d_norm_bins(const float* __restrict__ norms, int n, int* __restrict__ row_of,
                                            int* __restrict__ start) {
  __shared__ int s_cnt[NORM_BINS], s_cur[NORM_BINS];
  const int tid = threadIdx.x;
  for (int b = tid; b < NORM_BINS; b += 1024) s_cnt[b] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) atomicAdd(&s_cnt[norm_bin(norms[i])], 1);
  __syncthreads();
  if (tid < 64) {  // exclusive scan of 192 counters by one wave (three per lane)
    const int c0 = s_cnt[3 * tid], c1 = s_cnt[3 * tid + 1], c2 = s_cnt[3 * tid + 2];
    int tot;
    const int ex = wave_excl_scan_i32(c0 + c1 + c2, &tot);
    s_cur[3 * tid] = ex;
    s_cur[3 * tid + 1] = ex + c0;
    s_cur[3 * tid + 2] = ex + c0 + c1;
    start[3 * tid] = ex;
    start[3 * tid + 1] = ex + c0;
    start[3 * tid + 2] = ex + c0 + c1;
    if (tid == 0) start[NORM_BINS] = tot;
  }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) row_of[atomicAdd(&s_cur[norm_bin(norms[i])], 1)] = i;
}
template <bool EXT>
__global__ __launch_bounds__(1024) void k_norm_bins(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  static_assert(NORM_BINS == 192, "three counters per lane of one wave");
  if (blockIdx.y == 0)
    d_norm_bins(V.norms_i, V.n_large, V.nb_row_i, V.nb_start_i);
  else
    d_norm_bins(V.norms_j, V.n_small, V.nb_row_j, V.nb_start_j);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k_norm_gather(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y == 0 ? V.n_large : V.n_small, pad = blockIdx.y == 0 ? V.pad_large : V.pad_small;
  if (c >= n) return;
  const float* __restrict__ src = blockIdx.y == 0 ? V.baseT_i : V.baseT_j;
  float* __restrict__ dst = blockIdx.y == 0 ? V.baseTb_i : V.baseTb_j;
  const int r = (blockIdx.y == 0 ? V.nb_row_i : V.nb_row_j)[c];
#pragma unroll
  for (int k = 0; k < 34; ++k) dst[(size_t)k * pad + c] = src[(size_t)k * pad + r];
}

// How the work of one k_nn_mfma launch is cut into items = (pair, block of 512 queries, slice of the base cloud), and in
// which order the X workgroups of the launch take them.
//   * few query blocks (one pair): as many slices as give every workgroup one item and no more — a second round with a
//     handful of items costs as much as a full one;
//   * many (a group of pairs): items of ~1/8 of a workgroup's share, handed out through an atomic counter, so that
//     pairs of different size and the ragged last round cost a few per cent instead of a third of the launch.
// Evaluated by wave 0 of every workgroup (of this kernel and of the f32 engine's k_nn_finish, which has to find the same
// slicing; k_nn_f16 leaves its slicing in the counter line for k_nn_finish_f16) from
// the device-side query counts of all pairs; results in LDS.  Written as a macro on purpose: the kernel-argument
// structs must not travel by reference (see ViewExt).
// a / b for a < 2^22, b >= 1: a float quotient and one correction step each way instead of the ~40-instruction integer
// division (the work-item arithmetic of the nearest-neighbour kernels sits in front of their first load, and a kernel's
// first pass over its code is paced by instruction fetch: every instruction less there is time).  v_rcp_f32 is good to
// 1 ulp: the float quotient is within 0.75 of a / b, its truncation within one of the answer.
__device__ __forceinline__ u32 fast_udiv_small(u32 a, u32 b) {
  u32 q = (u32)((float)a * __builtin_amdgcn_rcpf((float)b));
  if (q * b > a) --q;
  if ((q + 1) * b <= a) ++q;
  return q;
}
__device__ __forceinline__ u32 fast_udiv(u32 a, u32 b) {
  if (a >> 22) return a / b;
  return fast_udiv_small(a, b);
}
// The plan of a single pair's k_nn_f16 launch (what NN_PLAN below computes for a group of pairs): slices per query block
// and base tiles per slice, packed (slices | tiles << 8) as MC_NSPLITx holds them.  qb query blocks, nt base tiles, X
// workgroups.  The host runs it for direction 0 (it knows both clouds' sizes), k_hit_compact for direction 1 (it has
// just counted the hit rows): the search kernel then opens with two scalar loads instead of ~350 instructions.
__host__ __device__ inline int nn_plan_single(int qb, int nt, int X) {
  int sp;
  if (qb <= X) {
    sp = X / (qb > 1 ? qb : 1);
  } else {
    int T = (qb * nt + X * 8 - 1) / (X * 8);
    if (T < 64) T = 64;
    sp = (nt + T - 1) / T;
  }
  if (sp > NN_MAXSPLIT) sp = NN_MAXSPLIT;
  if (sp > nt) sp = nt;
  if (sp < 1) sp = 1;
  const int tps = (nt + sp - 1) / sp;
  sp = (nt + tps - 1) / tps;
  return sp | (tps << 8);
}
#define NN_MAXG 64
#define NN_PLAN_DECL __shared__ int s_off[NN_MAXG + 1], s_ns[NN_MAXG], s_tps[NN_MAXG], s_nq[NN_MAXG];
#define NN_PLAN(G_, dir_, X_) \
  NN_PLAN_DECL                \
  NN_PLAN_BODY(G_, dir_, X_)
#define NN_PLAN_BODY(G_, dir_, X_)                                                                              \
  if (threadIdx.x < 64) {                                                                                       \
    const int g_ = threadIdx.x;                                                                                 \
    int qb_ = 0, nt_ = 1, nq_ = 0;                                                                              \
    if (g_ < (G_)) {                                                                                            \
      const MatchView& P_ = EXT ? x.ext[g_] : one;                                                              \
      nq_ = (dir_) == 0 ? P_.n_small : P_.mcounts[P_.d[dir_].nq_slot]; /* (direction 0: no load to wait for) */ \
      qb_ = (nq_ + NN_QPB - 1) / NN_QPB;                                                                        \
      nt_ = P_.d[dir_].nb_pad / 32;                                                                             \
    }                                                                                                           \
    const int S_ = wave_sum_i32(qb_), W_ = wave_sum_i32(qb_ * nt_);                                             \
    int sp_;                                                                                                    \
    if (S_ <= (X_)) {                                                                                           \
      sp_ = (int)fast_udiv((u32)(X_), (u32)max(S_, 1));                                                         \
    } else {                                                                                                    \
      const int T_ = max(64, (int)fast_udiv((u32)(W_ + (X_) * 8 - 1), (u32)((X_) * 8)));                        \
      sp_ = (int)fast_udiv((u32)(nt_ + T_ - 1), (u32)T_);                                                       \
    }                                                                                                           \
    sp_ = max(1, min(min(sp_, NN_MAXSPLIT), nt_));                                                              \
    const int tps_ = (int)fast_udiv((u32)(nt_ + sp_ - 1), (u32)sp_);                                            \
    sp_ = (int)fast_udiv((u32)(nt_ + tps_ - 1), (u32)tps_);                                                     \
    int tot_;                                                                                                   \
    const int ex_ = wave_excl_scan_i32(qb_ * sp_, &tot_);                                                       \
    if (g_ < (G_)) {                                                                                            \
      s_off[g_] = ex_;                                                                                          \
      s_ns[g_] = sp_;                                                                                           \
      s_tps[g_] = tps_;                                                                                         \
      s_nq[g_] = nq_;                                                                                           \
    }                                                                                                           \
    if (g_ == 0) s_off[(G_)] = tot_;                                                                            \
  }                                                                                                             \
  __syncthreads();

// grid (X): persistent workgroups that take items until the launch's counter runs out.  Each wave keeps 4 x 32 query
// columns stationary (68 VGPRs), streams 32-row base tiles (17 coalesced dword loads per lane, software prefetched one
// tile ahead in a second register set) and issues 68 MFMAs per tile.
template <bool EXT>
__global__ __launch_bounds__(256, 2) void k_nn_mfma(ViewExt<MatchView> x, MatchView one, int dir, int G) {
  NN_PLAN(G, dir, (int)gridDim.x)
  __shared__ int s_item;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  int* counter = (EXT ? x.ext[0].mcounts : one.mcounts) + 14 + dir;  // zeroed by k_match_init
  const int total = s_off[G];
#pragma unroll 1
  while (true) {
    if (threadIdx.x == 0) s_item = atomicAdd(counter, 1);
    __syncthreads();
    const int item = s_item;
    __syncthreads();
    if (item >= total) break;
    int g = 0;
    while (g + 1 < G && item >= s_off[g + 1]) ++g;
    const MatchView& V = EXT ? x.ext[g] : one;  // (inline on purpose: see ViewExt)
    const NnDir& D = V.d[dir];
    const int nb_pad = D.nb_pad, nq_pad = D.nq_pad;
    const int ntiles = nb_pad / 32;
    const int nsplit = s_ns[g], tps = s_tps[g];
    const float* __restrict__ baseT = D.baseT;
    const float* __restrict__ queryT = D.queryT;
    NnPartial* __restrict__ partial = V.partial;
    const int local = item - s_off[g];
    const int qb = local / nsplit, slice = local - qb * nsplit;
    const int qbase = (qb * 4 + wave) * NN_QPW + col;
    float q[4][NN_K2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int kk = 0; kk < NN_K2; ++kk) q[c][kk] = queryT[(size_t)(2 * kk + half) * nq_pad + qbase + 32 * c];
    // running best / second best per query column.  The accumulator register r a value came from rides in the
    // four low mantissa bits of the value itself (v_and_or_b32), so the update is three VALU ops per value —
    // pack, second = med3(best, second, v), best = min(best, v) — and the winning TILE is found once per tile
    // by noticing that best changed.  The <16 ulp perturbation is part of the rounding bound of the finish.
    float b1[4], b2[4];
    int it1[4];  // tile of the best
    float ninf = -INFINITY;
    asm volatile("" : "+v"(ninf));  // opaque: med3(best, v, -inf) stays ONE v_med3 (a literal folds to canonicalise + min)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      b1[c] = b2[c] = INFINITY;
      it1[c] = -1;
    }
    const int t_begin = slice * tps, t_end = min(ntiles, t_begin + tps);
    // uniform row pointers (scalar registers) + ONE 32-bit lane offset: the 17 loads of a tile use the scalar-base
    // addressing form instead of 17 64-bit address pairs in vector registers (the kernel has to fit 256 registers: with
    // more the accumulators move to AGPRs and every value of the epilogue costs an extra v_accvgpr_read)
    const u32 lane_off = (u32)half * (u32)nb_pad + (u32)col;
    float m0[NN_K2], m1[NN_K2];
    auto load_tile = [&](float* m, int t) {
      const u32 off = lane_off + (u32)t * 32u;
#pragma unroll
      for (int kk = 0; kk < NN_K2; ++kk) m[kk] = (baseT + (size_t)(2 * kk) * nb_pad)[off];
    };
    auto compute_tile = [&](const float* m, int t) {
      f32x16 acc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int kk = 0; kk < NN_K2; ++kk)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(m[kk], q[c][kk], acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float before = b1[c];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = __uint_as_float((__float_as_uint(acc[c][r]) & 0xfffffff0u) | (u32)r);
          b2[c] = __builtin_amdgcn_fmed3f(b1[c], b2[c], v);
          b1[c] = __builtin_amdgcn_fmed3f(b1[c], v, ninf);
        }
        it1[c] = (b1[c] != before) ? t : it1[c];
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
    for (int c = 0; c < 4; ++c) {
      // decode the best's row, then merge the two lanes (half 0 / half 1) that own the same query column
      const int r = (int)(__float_as_uint(b1[c]) & 15u);
      int row = (it1[c] < 0) ? -1 : (it1[c] * 32 + 4 * half + (r & 3) + 8 * (r >> 2));
      const float ob1 = __shfl_xor(b1[c], 32, 64), ob2 = __shfl_xor(b2[c], 32, 64);
      const int orow = __shfl_xor(row, 32, 64);
      const bool take = (ob1 < b1[c]) || (ob1 == b1[c] && orow >= 0 && (row < 0 || orow < row));
      const float nb2 = fminf(fminf(b2[c], ob2), take ? b1[c] : ob1);
      const float nb1 = take ? ob1 : b1[c];
      row = take ? orow : row;
      if (half == 0) {
        NnPartial p;
        p.b1 = nb1;
        p.b2 = nb2;
        p.i1 = row;
        p.pad = 0;
        partial[(size_t)(qbase + 32 * c) * NN_MAXSPLIT + slice] = p;  // (fixed stride: the finish reads before it knows nsplit)
      }
    }
  }
}

// =================================================================================================
// f16-split engine (the default).  On gfx950 v_mfma_f32_32x32x2_f32 and the VALU exclude each other — a wave's (and a
// SIMD's) FP32 matrix instructions and vector instructions add up, they do not overlap (tests/probe/gen_probe3.py:
// 4352 clocks of MFMA + 880 of fold = 5230 per tile however the two are interleaved, one or two waves per SIMD) — while
// v_mfma_f32_32x32x16_f16 runs on the matrix cores proper: 32 clocks each, and VALU work interleaved with it is hidden.
// The filter therefore evaluates v = nb' - 2 a.b on the f16 pipe with every f32 operand split in two halves:
//     x^ = S x,  x1 = RN16(x^),  x2 = RN16(x^ - x1),   x^ = x1 + x2 + dx,  |dx| <= max(2^-22 |x^|, 2^-14)
// (the second alternative covers a subnormal x2, whether or not the hardware flushes it), S = 128 for base rows, -2S for
// queries, and a.b ~ sum (b1 q1 + b2 q1 + b1 q2): 3 x 33 products, plus three slots that carry nb' = c1 + c2 + c3 (f16
// pieces) against the constant S^2 = 16384: K = 102, padded to 112 = 7 MFMAs per accumulator instead of 17 at half the
// issue time — 896 clocks of matrix work per tile instead of 4352, with the top-2 fold riding under it (rounds 2-4: three
// vector instructions per value; round 5: ~1.8, see gen_nn_f16_core.py).
// Every product of two f16 numbers is exact in f32; the accumulation inside and across the 7 MFMAs was measured at
// <= 5.5 u sum|terms| on adversarial exponents (3.0 u on histogram-like data; same probe) and is budgeted at 16 u.
//
// Rounding bound, unscaled (u = 2^-24):  |V / S^2 - (nb' - 2 a.b)| <=
//     12.1u (|a|^2 + |b|^2)   dropped products b2 q2, db q, b dq (3 x 2^-22 |b^ q^| each, Cauchy-Schwarz)
//   + 16.1u (|a|^2 + 2|b|^2)  accumulation, sum|terms| / S^2 <= nb' + 2|a||b| (1 + 2^-10)
//   + 32u   (|a|^2 + 2|b|^2)  the accumulator-register index in the four low mantissa bits (rounds 2-4 only, see below)
//   + 8u (33 + |a|^2) + 4u (33 + |b|^2)   subnormal second halves: 2^-14 (sum|q^| + sum|b^|) / S^2, |x| <= (1 + x^2)/2
//   <= 69u |a|^2 + 114u |b|^2 + 396u
// against 66u |a|^2 + 132u |b|^2 of the f32 chain: the same certification inequality holds with + 800u on the right-hand
// side (k_nn_finish; rounds 2-4).
// Round 5: the fold no longer packs anything into the values (gen_nn_f16_core.py), so the third line is gone:
//     E <= 37u |a|^2 + 50u |b|^2 + 396u,
// and the tables' norm row is scaled down by 56u instead of the f32 chain's 141u (NNH_NORM_SCALE; it only has to cover
// the |b|^2 term of E): nb' <= |b|^2 (1 - 56u), nb' >= |b|^2 (1 - 58u).  As in the f32 derivation above, for EVERY base
// row   d(a,b) >= |a|^2~ + v - 38u |a|^2 - 396u   (the |b|^2 part of the error is inside the scaling), for the winner
//     d(a,b1) <= |a|^2~ + v1 + 38u |a|^2 + 108u |b1|^2 + 396u,   and the exact-order float evaluation obeys |d_ex - d| <= 37u d:
// the approximate winner is the exact arg-min whenever
//     v2 - v1 > u (76 |a|^2 + 108 |b1|^2 + 40 (d~1 + d~2) + 800)     (k_nn_finish_f16, 1 % slack on top)
// — 2.3 x tighter than the inequality the f32 chain needs, which is what the listed rows of dense clouds (thousands of
// near-identical descriptors with |b|^2 = 30000) were waiting for.  The re-check's threshold follows: v1 + u (76 |a|^2 +
// 108 |b1|^2 + 80 d~1 + 800).  (Both have an a-posteriori companion in k_nn_finish_f16, which knows the exact distance of
// the candidate it chose: see "charges the leader's error as it is" there.)  Preconditions, checked per pair by k_half_tables: every |descriptor value| <= 255 (f16 range of
// -256 x) and every |b|^2 < 65000 (f16 range of the norm pieces); a pair that violates them (not an FPFH descriptor:
// those are bounded by 100) sets MC_UNSAFE and k_nn_finish_f16 sends all of its rows to the exact re-check.
// Hidden / pad rows carry 3 x 65504 in the norm slots (196512 > any real nb').
//
// Operand layout (both operands): 32 rows form a tile; a row's 112 halves are cut into 14 chunks of 8 (chunk 2m+g is
// what lane group g = lane/32 feeds to MFMA m); a tile stores chunk-major, [tile][14][32 rows] x 16 bytes, so that a
// fragment load is one coalesced 16-byte load per lane.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define NNH_CHUNKS 14
#define NNH_S 128.0f
#define NNH_UNSCALE 6.103515625e-05f  // 1 / S^2 = 2^-14

struct HalfRow {
  _Float16 h1[33], h2[33];
};
__device__ __forceinline__ void half_split(const float* v, float scale, HalfRow& o) {
#pragma unroll
  for (int k = 0; k < 33; ++k) {
    const float xs = v[k] * scale;  // exact: power of two
    const _Float16 a = (_Float16)xs;
    o.h1[k] = a;
    o.h2[k] = (_Float16)(xs - (float)a);
  }
}
// slot s of the base role / the query role (compile-time s after unrolling)
__device__ __forceinline__ _Float16 base_slot(const HalfRow& b, const _Float16* c, int s) {
  return s < 33 ? b.h1[s] : s < 66 ? b.h2[s - 33] : s < 99 ? b.h1[s - 66] : s < 102 ? c[s - 99] : (_Float16)0.f;
}
__device__ __forceinline__ _Float16 query_slot(const HalfRow& q, int s) {
  return s < 33 ? q.h1[s] : s < 66 ? q.h1[s - 33] : s < 99 ? q.h2[s - 66] : s < 102 ? (_Float16)16384.f : (_Float16)0.f;
}
// One launch after k_desc_prep (which filled the dedup table): the rows of this workgroup staged through LDS, a row's
// representative looked up (a row that repeats a lower row bit for bit is hidden: norm slots 3 x 65504), the range
// checks, and both operand tables written.  The engine has no other copy of the descriptors: the k-major f32 tables of
// the other engines are not built.  grid (pad_large_max / 256, 2, pairs)
template <int ROWS>
__device__ __forceinline__ void d_half_tables(const float* __restrict__ desc, int n, int n_pad, const u64* __restrict__ hashes,
                                              const u64* __restrict__ table, int mask, int* __restrict__ hidden_count,
                                              uint4* __restrict__ baseH, uint4* __restrict__ queryH, int* __restrict__ unsafe,
                                              float* s_rows /* LDS, ROWS x 33 */) {
  const int row0 = blockIdx.x * ROWS;
  if (row0 >= n_pad) return;
  QTR_STAMP(STAMP_HALF_TABLES, 0)
  stage_rows<ROWS>(desc, row0, n, s_rows);
  __syncthreads();
  QTR_STAMP(STAMP_HALF_TABLES, 1)
  const int i = row0 + threadIdx.x;
  if (i >= n_pad) return;
  float v[33];
  bool bad = false;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 33; ++k) {
    v[k] = s_rows[threadIdx.x * 33 + k];  // (zeros past the cloud)
    acc += (double)v[k] * (double)v[k];
    bad = bad || !(fabsf(v[k]) <= 255.0f);
  }
  float nbp = 1e30f;  // scaled-down |b|^2 (the lower bound the chain works with); 1e30: hidden duplicate or pad row
  bool hide = false;
  if (i < n) {
    bad = bad || !((float)acc < 65000.0f);
    nbp = __double2float_rd(acc * NNH_NORM_SCALE);
    const u64 h = hashes[i];
    const u64 tag = h & 0xffffffff00000000ULL;
    u32 slot = (u32)h & (u32)mask;
    int rep = i;
    for (int probe = 0; probe <= mask; ++probe) {
      const u64 cur = table[slot];
      if (cur == ~0ULL) break;
      if ((cur & 0xffffffff00000000ULL) == tag) {
        rep = (int)(u32)cur;
        break;
      }
      slot = (slot + 1) & (u32)mask;
    }
    QTR_STAMP(STAMP_HALF_TABLES, 2)
    if (rep < i) {  // the 33 values are compared before a row is hidden: a hash collision only costs a missed merge
      u32 diff = 0;  // (no early exit: 33 independent loads instead of a chain of 33 round trips)
#pragma unroll
      for (int k = 0; k < 33; ++k) diff |= __float_as_uint(v[k]) ^ __float_as_uint(desc[(size_t)rep * 33 + k]);
      hide = diff == 0;
      if (hide) nbp = 1e30f;
    }
    QTR_STAMP(STAMP_HALF_TABLES, 3)
  }
  if (bad) *unsafe = 1;
  const u64 bal = __ballot(hide);
  if (qk_lane() == 0 && bal) atomicAdd(hidden_count, __popcll(bal));
  _Float16 c[3];
  if (!(nbp < 65000.0f)) {
    c[0] = c[1] = c[2] = (_Float16)65504.f;
  } else {
    c[0] = (_Float16)nbp;
    const float r1 = nbp - (float)c[0];
    c[1] = (_Float16)r1;
    c[2] = (_Float16)(r1 - (float)c[1]);
  }
  HalfRow hb, hq;
  half_split(v, NNH_S, hb);
  half_split(v, -2.0f * NNH_S, hq);
  const size_t t0 = ((size_t)(i >> 5) * NNH_CHUNKS) * 32 + (i & 31);
#pragma unroll
  for (int ch = 0; ch < NNH_CHUNKS; ++ch) {
    h8 b8, q8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      b8[e] = base_slot(hb, c, 8 * ch + e);
      q8[e] = query_slot(hq, 8 * ch + e);
    }
    ((h8*)baseH)[t0 + (size_t)ch * 32] = b8;
    ((h8*)queryH)[t0 + (size_t)ch * 32] = q8;
  }
}
#define HT_ROWS_SINGLE 64
template <bool EXT>
__global__ __launch_bounds__(256) void k_half_tables(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  // a thread per row: one pair at a time 256-row workgroups would fill 144 of the 256 compute units, so it runs as
  // one-wave workgroups there (the host launches HT_ROWS_SINGLE threads); the batched path has rows enough
  constexpr int ROWS = EXT ? 256 : HT_ROWS_SINGLE;
  __shared__ float s_rows[ROWS * 33];
  if (blockIdx.y == 0)
    d_half_tables<ROWS>(V.fpfh_i, V.n_large, V.pad_large, V.hash_i, V.table_i, V.dd_mask, V.mcounts + MC_HIDDEN_I, V.baseH_i, V.queryH_i,
                  V.mcounts + MC_UNSAFE, s_rows);
  else
    d_half_tables<ROWS>(V.fpfh_j, V.n_small, V.pad_small, V.hash_j, V.table_j, V.dd_mask, V.mcounts + MC_HIDDEN_J, V.baseH_j, V.queryH_j,
                  V.mcounts + MC_UNSAFE, s_rows);
}

// Same items, queue and partial records as k_nn_mfma.  Per wave: 4 x 32 stationary query columns (4 x 7 fragments of
// 16 bytes, in AGPRs), TWO accumulator sets — the 28 MFMAs of a tile go to one set while the 64 values the previous
// tile left in the other set are folded into the running top-2 in the shadow of the matrix instructions.  The inner
// loop is the hand-scheduled block of nn_f16_core.inc (generated by gen_nn_f16_core.py: register map, schedule and the
// wait states it has to respect are documented there).  ~250 VGPRs + 112 AGPRs: one workgroup per compute unit.
#include "nn_f16_core.inc"
#ifdef QTR_NN_LDS_STAGE  // experiment build (tests/gpu_r6_nnlds.sh): the base tile staged through LDS, see gen_nn_f16_core.py --lds
#include "nn_f16_core_lds.inc"
#endif
// -DQTR_NN_TIMING (diagnostic build, tests/probe/nn_stamps.py): thread 0 of every workgroup records the shader clock and
// the 100 MHz wall clock at five points of its first item
#ifdef QTR_NN_TIMING
__device__ unsigned long long g_nn_stamp[2][256][12];
#define NN_STAMP(i)                                                          \
  if (threadIdx.x == 0 && blockIdx.x < 256) {                                \
    g_nn_stamp[dir][blockIdx.x][2 * (i)] = clock64();                        \
    g_nn_stamp[dir][blockIdx.x][2 * (i) + 1] = wall_clock64();               \
  }
extern "C" int qtr_debug_nn_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_stamp), sizeof(g_nn_stamp));
}
extern "C" int qtr_debug_stamps(unsigned long long* out) {  // [QTR_STAMP_KERNELS][32][QTR_STAMP_POINTS][2]
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(g_stamp));
}
#else
#define NN_STAMP(i)
#endif
template <bool EXT>
__global__ __launch_bounds__(256, 1) void k_nn_f16(ViewExt<MatchView> x, MatchView one, int dir, int G, int la, int plan) {
  NN_STAMP(0)
  NN_PLAN_DECL
  // A group of pairs plans in the kernel (NN_PLAN).  A single pair's plan is one word (see nn_plan_single): `plan` from
  // the host (direction 0), or, plan < 0, the word k_hit_compact left beside its count of hit rows (direction 1).
  int u_ns = 1, u_tps = 1, u_nq = 0, u_qb = 0;
  if constexpr (EXT) {
    NN_PLAN_BODY(G, dir, (int)gridDim.x)
    // the slicing the plan decided, for k_nn_finish_f16 (which then needs neither the plan nor its barrier)
    if (blockIdx.x == 0 && threadIdx.x < G) x.ext[threadIdx.x].mcounts[MC_NSPLIT0 + dir] = s_ns[threadIdx.x] | (s_tps[threadIdx.x] << 8);
  } else {
    int w = plan;
    u_nq = one.n_small;
    if (plan < 0) {
      w = one.mcounts[MC_NSPLIT0 + dir];
      u_nq = one.mcounts[one.d[dir].nq_slot];
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
      one.mcounts[MC_NSPLIT0 + dir] = w;
    }
    u_ns = w & 0xff;
    u_tps = w >> 8;
    u_qb = (u_nq + NN_QPB - 1) / NN_QPB;
  }
  NN_STAMP(1)
  __shared__ int s_item;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  int* counter = (EXT ? x.ext[0].mcounts : one.mcounts) + 14 + dir;  // zeroed by k_match_init
  const int total = EXT ? s_off[G] : u_qb * u_ns;
  // the first item of a workgroup is its own index, later ones come from the counter (which therefore counts from
  // gridDim.x on): a launch with no more items than workgroups — every single-pair launch — never touches it, and nobody
  // waits for an atomic's round trip before the first load
  int item = blockIdx.x;
#pragma unroll 1
  while (true) {
    // A single pair's one round of items: consecutive workgroup ids go to the eight XCDs in turn, and with item =
    // (query block, slice) numbered slice-fastest every XCD would stream ALL query blocks against one eighth of the
    // base (8 x (Q + B / 8) of L2 fill: 33 MB for direction 0's 7.6 MB of tables).  Instead the (query block, slice)
    // grid is cut into 2 x 4 rectangles, enumerated rectangle by rectangle, and XCD x takes the x-th eighth of that
    // enumeration (workgroup 8 j + x its j-th cell): every XCD then meets about half of the query blocks and a quarter
    // of the base, 8 x (Q / 2 + B / 4).  (Round 5: 4 x 2 — a quarter of the query blocks, half of the base — measured a
    // microsecond faster per launch: the launch opens with every workgroup fetching its 114 KB of query fragments at
    // once, and four instead of two of the eight workgroups that share a query block then share an L2.)
    const bool dealt = !EXT && total <= (int)gridDim.x && (gridDim.x & 7) == 0;
    int cell = item;
    if (dealt) {
      const int per = (total + 7) >> 3, j = item >> 3;
      cell = (j < per) ? (item & 7) * per + j : total;
    }
    if (cell >= total) break;
    int g = 0;
    if constexpr (EXT)
      while (g + 1 < G && cell >= s_off[g + 1]) ++g;
    const MatchView& V = EXT ? x.ext[g] : one;  // (inline on purpose: see ViewExt)
    const NnDir& D = V.d[dir];
    const int ntiles = D.nb_pad / 32;
    const int nsplit = EXT ? s_ns[g] : u_ns, tps = EXT ? s_tps[g] : u_tps;
    NnPartial* __restrict__ partial = V.partial;
    const int local = EXT ? cell - s_off[g] : cell;
    int qb, slice;
    if (dealt) {
      // (which rectangle: subtractions; ONE division — the eight of the unrolled form were a microsecond of every launch)
      // (2^la x 2^(3-la) rectangles; la = 1 is the 2 x 4 cut described above)
      const int lb = 3 - la;
      const int nqb = u_qb;
      const int qh = (nqb + (1 << la) - 1) >> la, sq = (nsplit + (1 << lb) - 1) >> lb;
      int rem = cell, q0 = 0, s0 = 0, cols = 1;
      bool found = false;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int A = r >> lb, B = r & ((1 << lb) - 1);
        const int rows = min(qh, max(0, nqb - A * qh));
        const int cl = min(sq, max(0, nsplit - B * sq)), cells = rows * cl;
        if (!found && rem < cells) {
          q0 = A * qh;
          s0 = B * sq;
          cols = cl;
          found = true;
        }
        if (!found) rem -= cells;
      }
      const int qd = (int)fast_udiv_small((u32)rem, (u32)cols);  // (rem < total <= gridDim.x)
      qb = q0 + qd;
      slice = s0 + rem - qd * cols;
    } else {
      qb = (int)fast_udiv((u32)local, (u32)nsplit);
      slice = local - qb * nsplit;
    }
    const int qbase = (qb * 4 + wave) * NN_QPW + col;
    const int t_begin = slice * tps, t_end = min(ntiles, t_begin + tps);
    float b1[4], b2[4];
    int it1[4];  // tile of the best
    NN_STAMP(2)
    if (t_begin < t_end) {
      // where this lane's four query rows start in the query table (direction 1: the hit rows, looked up in place —
      // a list of rows needs no gathered copy of the table); columns past the list take row 0, nobody reads their result
      u32 qoff[4];
      {
        const int nq = EXT ? s_nq[g] : u_nq;  // (the plan read it)
        const int* __restrict__ qmap = D.qmap;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int qi = qbase + 32 * c;
          const int row = qmap ? ((qi < nq) ? qmap[qi] : 0) : qi;
          qoff[c] = ((u32)((row >> 5) * NNH_CHUNKS + half) * 32u + (u32)(row & 31)) * 16u;
        }
      }
#ifdef QTR_NN_LDS_STAGE
      __shared__ __attribute__((aligned(16))) uint4 s_nn_tiles[2 * 512];  // two 8 KB buffers
      nn_f16_core_lds(D.queryH, qoff, D.baseH + (size_t)t_begin * (NNH_CHUNKS * 32), t_end - t_begin, t_begin,
                      ((u32)half * 32u + (u32)col) * 16u, (u32)(uintptr_t)s_nn_tiles, wave, b1, b2, it1);
#else
      nn_f16_core(D.queryH, qoff, D.baseH + (size_t)t_begin * (NNH_CHUNKS * 32), t_end - t_begin, t_begin,
                  ((u32)half * 32u + (u32)col) * 16u, b1, b2, it1);
#endif
    NN_STAMP(3)
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        b1[c] = b2[c] = INFINITY;
        it1[c] = -1;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // merge the two lanes (half 0 / half 1) that own the same query column.  No row index travels with a value: a
      // lane knows the tile and the quad of accumulator registers its best came from, i.e. 4 candidate rows — rows
      // 8 quad + 4 half + {0..3} of that tile — and k_nn_finish_f16 picks the candidate with the smallest exact
      // distance.  i1 = 2 * (4 tile + quad) + half.
      int cand = (it1[c] < 0) ? -1 : (it1[c] * 2 + half);
      const float ob1 = __shfl_xor(b1[c], 32, 64), ob2 = __shfl_xor(b2[c], 32, 64);
      const int ocand = __shfl_xor(cand, 32, 64);
      const bool take = (ob1 < b1[c]) || (cand < 0 && ocand >= 0);  // (a tie leaves second == best: never certified)
      const float nb2 = fminf(fminf(b2[c], ob2), take ? b1[c] : ob1);
      const float nb1 = take ? ob1 : b1[c];
      cand = take ? ocand : cand;
      if (half == 0) {
        NnPartial p;
        p.b1 = nb1 * NNH_UNSCALE;  // exact (power of two): the finish works in unscaled units
        p.b2 = nb2 * NNH_UNSCALE;
        p.i1 = cand;
        p.pad = 0;
        partial[(size_t)(qbase + 32 * c) * NN_MAXSPLIT + slice] = p;  // (fixed stride: the finish reads before it knows nsplit)
      }
    }
    NN_STAMP(4)
#ifdef QTR_NN_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 256) g_nn_stamp[dir][blockIdx.x][10] = (unsigned long long)(t_end - t_begin);
#endif
    if (total <= (int)gridDim.x) break;  // (uniform) nothing beyond the first round
    __syncthreads();
    if (threadIdx.x == 0) s_item = (int)gridDim.x + atomicAdd(counter, 1);
    __syncthreads();
    item = s_item;
  }
}

// The f32 MFMA engine's finish (test build: the product's f16 engine has k_nn_finish_f16 below).  Merge the per-slice
// partials and decide each query: certified (see the header comment) or listed for the exact re-check.  X = gridDim.x of
// the k_nn_mfma launch it follows.  grid (ceil(nq_max/512), 1, pairs).
#define NN_FIN_THREADS 512
template <bool EXT>
__global__ __launch_bounds__(NN_FIN_THREADS) void k_nn_finish(ViewExt<MatchView> x, MatchView one, int dir, int X, int G,
                                                              float cadd, int f16) {
  NN_PLAN(G, dir, X)
  __shared__ int s_w[16], s_base;
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const NnDir& D = V.d[dir];
  const int nq = V.mcounts[D.nq_slot];
  if ((int)blockIdx.x * NN_FIN_THREADS >= nq) return;
  const int t = blockIdx.x * NN_FIN_THREADS + threadIdx.x;
  const bool valid = t < nq;
  QTR_STAMP(STAMP_NN_FINISH, 0)
  // the queries are visited in norm-bin order (direction 0: qorder; direction 1: the hit list is built in that order)
  // and the re-check list is appended in visiting order, one block of 512 queries at a time: the eight rows a re-check
  // workgroup shares then have overlapping spans of the base cloud
  const int q = valid ? ((D.qorder && !f16) ? D.qorder[t] : t) : 0;  // (the f16 engine's re-check has no spans: plain order)
  const int nsplit = s_ns[blockIdx.z];
  float b1 = INFINITY, b2 = INFINITY;
  int i1 = -1;
  const NnPartial* __restrict__ partial = V.partial;
  if (valid)
    for (int s0 = 0; s0 < nsplit; s0 += 8) {  // the records of eight slices per round trip
      NnPartial pp[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) pp[k] = partial[(size_t)q * NN_MAXSPLIT + min(s0 + k, nsplit - 1)];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (s0 + k < nsplit) {
          const NnPartial p = pp[k];
          const bool take = (p.b1 < b1) || (p.b1 == b1 && p.i1 >= 0 && (i1 < 0 || p.i1 < i1));
          const float nb2 = fminf(fminf(b2, p.b2), take ? b1 : p.b1);
          b1 = take ? p.b1 : b1;
          i1 = take ? p.i1 : i1;
          b2 = nb2;
        }
    }
  QTR_STAMP(STAMP_NN_FINISH, 1)
  const int row = valid ? (D.qmap ? D.qmap[q] : q) : 0;
  const float na = valid ? D.qnorm[f16 ? row : q] : 0.f;  // (the f16 engine's query tables are the cloud's own: by row)
  const float nb1 = (i1 >= 0) ? D.bnorm[i1] : 0.f;
  const float u = 5.9604645e-08f;
  const float d1 = fmaxf(na + b1, 0.f) + 1.0f;  // d~ of the leader (|a|^2 is not inside b1)
  const float d2 = (b2 < INFINITY) ? fmaxf(na + b2, 0.f) + 1.0f : d1;
  // cadd: the engine's constant term (f16-split engine: 800; see its header comment)
  const float gap = u * (144.0f * na + 280.0f * nb1 + 40.0f * (d1 + d2) + cadd) * 1.01f;
  const bool unsafe = V.mcounts[MC_UNSAFE] != 0;  // descriptor values outside the f16 engine's range: everything is re-checked
  if (unsafe) i1 = -1;
  const bool certified = i1 >= 0 && (b2 == INFINITY || b2 - b1 > gap);
  const bool listed = valid && !certified;
  if (valid) D.best[row] = certified ? (u64)(u32)i1 : ~0ULL;
  // ordered append: ranks inside the workgroup from ballots, one atomic per workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u64 bal = __ballot(listed);
  if (lane == 0) s_w[wave] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < NN_FIN_THREADS / 64; ++w) {
      const int c = s_w[w];
      s_w[w] = tot;
      tot += c;
    }
    s_base = tot ? atomicAdd(V.mcounts + D.rc_slot, tot) : 0;
  }
  __syncthreads();
  if (listed) {
    const int slot = s_base + s_w[wave] + __popcll(bal & lanemask_lt());
    V.recheck_rows[slot] = row;
    // a base row whose approximate (lower-bound) value exceeds this cannot be the exact arg-min; +inf when the slice
    // merge found nothing
    const float thr = (i1 >= 0) ? b1 + u * (144.0f * na + 280.0f * nb1 + 80.0f * d1 + cadd) * 1.02f + 1e-30f : INFINITY;
    V.recheck_thr[slot] = thr;
    if (f16) {
      V.recheck_q[slot] = row;  // where k_recheck_filter finds the row's f16 query fragments (row of D.queryH)
    } else {
      // columns of the norm-bin order that can hold the arg-min: |sqrt|b|^2 - sqrt|a|^2| <= sqrt(d) and d <= |a|^2~ + thr
      // (+ the rounding slack of the bound above, norms rounded to float: 0.02 and 0.1 % cover both generously)
      int lo = 0, hi = NORM_BINS - 1;
      if (i1 >= 0) {
        const float sa = sqrtf(fmaxf(na, 0.f)), R = sqrtf(fmaxf(na + thr, 0.f) + 1.0f) * 1.001f + 0.02f;
        lo = max(0, (int)floorf(sa - R));
        hi = min(NORM_BINS - 1, (int)floorf(sa + R));
      }
      V.recheck_span[slot] = make_int2(D.bstart[lo], D.bstart[hi + 1]);
    }
  }
  QTR_STAMP(STAMP_NN_FINISH, 2)
}

__device__ __forceinline__ float recheck_exact_dist(const float* __restrict__ a, const float* __restrict__ b);  // (below)
// The f16 engine's finish (round 5).
// k_nn_f16 no longer carries a row index with a value (its fold is ~1.8 instructions per value instead of 3.3:
// gen_nn_f16_core.py): a partial record names the tile, the QUAD of accumulator registers and the lane half its best came
// from, i.e. 4 candidate rows — rows 8 quad + 4 half + {0..3} of that tile: ONE aligned run of 528 bytes of the row-major
// descriptor table.  Per query: (1) the slices' records are merged (best, second, candidate code of the best); (2) the 4
// candidates get the EXACT flann::L2 distance — 4 lanes per query, the run staged through LDS with 16-byte loads — and
// the smallest (lowest row on a tie) is the query's row i1; (3) the certification of the header comment, with the LARGEST
// |b|^2 among the candidates in the bound.  Why that is sound: let r* be the row whose filter value is the best (unknown
// here, one of the candidates).  The bound grows with |b|^2, so passing it with the candidates' maximum implies passing it
// with |b(r*)|^2, hence r* is the exact arg-min of the whole cloud (lowest index on an exact tie: a tie partner's filter
// value would lie inside the gap) and therefore of the candidates: i1 = r*.  A query that does not pass goes to
// k_recheck_filter as before, its threshold computed with the same maximum (a larger threshold only lets more pairs
// through to the exact evaluation).  Pad rows of the last tile are not candidates; a hidden duplicate among them has the
// distance of its lower-indexed original and loses the tie.
// (First form of the round: the TILE only, 16 candidates and 2.1 KB of LDS per query — 64 queries filled a compute unit,
// 11 us per launch single, half a millisecond for a group of sixteen pairs: 5138 against 5407 batched registrations/s
// for the packed-index loop of rounds 2-4.  With four candidates the batched path gains too: 5620 against 5560.)
// 64 queries per workgroup of 256 threads, four lanes per query; grid (ceil(nq_max / 64), 1, pairs).
#define NN_FINH_Q 64
#define NN_FINH_PITCH 132  // floats per query of the candidate stage: 4 rows x 33
template <bool EXT>
__global__ __launch_bounds__(256) void k_nn_finish_f16(ViewExt<MatchView> x, MatchView one, int dir, float cadd) {
  __shared__ __attribute__((aligned(16))) float s_rows[4][16][NN_FINH_PITCH];
  __shared__ float s_qv[4][16][36];
  __shared__ float s_b1[NN_FINH_Q], s_b2[NN_FINH_Q], s_nbmax[NN_FINH_Q], s_na[NN_FINH_Q], s_e1[NN_FINH_Q];
  __shared__ int s_row[NN_FINH_Q], s_i1[NN_FINH_Q];
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const NnDir& D = V.d[dir];
  // (the records sit at a fixed stride of NN_MAXSPLIT per query — the slots past nsplit hold whatever an earlier launch
  // left and are not looked at — so their loads need neither counter and leave together with the counters' loads)
  const NnPartial* __restrict__ pr =
      V.partial + (size_t)(blockIdx.x * NN_FINH_Q + (threadIdx.x >> 6) * 16 + ((threadIdx.x & 63) >> 2)) * NN_MAXSPLIT;
  const int nq = dir == 0 ? V.n_small : V.mcounts[D.nq_slot];  // (direction 0: every row of the smaller cloud)
  if ((int)blockIdx.x * NN_FINH_Q >= nq) return;
  QTR_STAMP(STAMP_NN_FINISH, 0)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nsplit = V.mcounts[MC_NSPLIT0 + dir] & 0xff;  // (the plan word of the k_nn_f16 launch: slices | tiles per slice << 8)
  const bool unsafe = V.mcounts[MC_UNSAFE] != 0;   // descriptor values outside the f16 engine's range: everything is re-checked
  const float* __restrict__ A = D.A;
  const float* __restrict__ B = dir ? V.fpfh_j : V.fpfh_i;
  const int nb = D.nb;
  const int qj = lane >> 2, k = lane & 3;   // the lane's query of the wave's sixteen, and its candidate
  const int ql = wave * 16 + qj;            // ... of the workgroup's 64
  const int q = blockIdx.x * NN_FINH_Q + ql;
  // ---- (1) four lanes per query merge the slices' records (at most NN_MAXSPLIT = 32: eight per lane, all in flight)
  float b1 = INFINITY, b2 = INFINITY;
  int cand = -1, row = 0;
  float na = 0.f;
  NnPartial pp[8];  // (unconditional: the slots of a query past the count are inside the arena, and nobody looks at them)
#pragma unroll
  for (int i = 0; i < 8; ++i) pp[i] = pr[k + 4 * i];
  if (q < nq) {
    if (k == 0) {
      row = D.qmap ? D.qmap[q] : q;
      na = D.qnorm[row];  // (the f16 engine's query tables are the cloud's own: by row)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (k + 4 * i < nsplit) {
        const NnPartial p = pp[i];
        const bool take = (p.b1 < b1) || (p.b1 == b1 && p.i1 > cand) || (cand < 0 && p.i1 >= 0);
        b2 = fminf(fminf(b2, p.b2), take ? b1 : p.b1);
        b1 = take ? p.b1 : b1;
        cand = take ? p.i1 : cand;
      }
  }
#pragma unroll
  for (int m = 1; m <= 2; m <<= 1) {  // (a butterfly: all four lanes end up with the merged record)
    const float ob1 = __shfl_xor(b1, m, 64), ob2 = __shfl_xor(b2, m, 64);
    const int oc = __shfl_xor(cand, m, 64);
    const bool take = (ob1 < b1) || (ob1 == b1 && oc > cand) || (cand < 0 && oc >= 0);
    b2 = fminf(fminf(b2, ob2), take ? b1 : ob1);
    b1 = take ? ob1 : b1;
    cand = take ? oc : cand;
  }
  row = __shfl(row, lane & ~3, 64);
  QTR_STAMP(STAMP_NN_FINISH, 1)
  // ---- (2) exact distances of the query's 4 candidates: cand = (4 tile + quad) * 2 + half
  const int run0 = (cand >> 3) * 32 + 8 * ((cand >> 1) & 3) + 4 * (cand & 1);
  const int brow = run0 + k;
  const bool ok = cand >= 0 && brow < nb;
  float nbm = ok ? D.bnorm[brow] : 0.f;  // (issued with the staging loads: one round trip for both)
  float* rows = &s_rows[wave][qj][0];
  float* qv = &s_qv[wave][qj][0];
  const bool aligned = (((uintptr_t)B) & 15) == 0;
  if (cand >= 0) {  // the four lanes of a query stage ITS run of 33 sixteen-byte pieces
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int pc = k + 4 * i;
      if (pc < 33) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (aligned && run0 + 4 <= nb) {
          v = *(const float4*)(B + (size_t)run0 * 33 + 4 * pc);
        } else {  // the cloud's last rows, or a caller's table that is not 16-byte aligned
          const size_t f0 = (size_t)run0 * 33 + 4 * pc, fend = (size_t)nb * 33;
          v.x = f0 < fend ? B[f0] : 0.f;
          v.y = f0 + 1 < fend ? B[f0 + 1] : 0.f;
          v.z = f0 + 2 < fend ? B[f0 + 2] : 0.f;
          v.w = f0 + 3 < fend ? B[f0 + 3] : 0.f;
        }
        *(float4*)(rows + 4 * pc) = v;
      }
    }
  }
  {  // the query's own row
    const float* __restrict__ ar = A + (size_t)row * 33;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int e = k + 4 * i;
      if (e < 33) qv[e] = ar[e];
    }
  }
  // (the stage is the wave's own: LDS operations of one wave complete in order, nothing to wait for but the compiler)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  {
    const float result = recheck_exact_dist(qv, rows + k * 33);  // flann::L2 accumulation order
    u64 key = (ok && result == result) ? (((u64)__float_as_uint(result) << 32) | (u32)brow) : ~0ULL;
#pragma unroll
    for (int m = 2; m >= 1; m >>= 1) {
      const u64 ok2 = ((u64)(u32)__shfl_xor((int)(key >> 32), m, 64) << 32) | (u32)__shfl_xor((int)(u32)key, m, 64);
      key = ok2 < key ? ok2 : key;
      nbm = fmaxf(nbm, __shfl_xor(nbm, m, 64));
    }
    if (k == 0) {
      s_i1[ql] = (key == ~0ULL) ? -1 : (int)(u32)key;
      s_nbmax[ql] = nbm;
      s_e1[ql] = __uint_as_float((u32)(key >> 32));  // the chosen candidate's exact distance (NaN pattern when there is none)
      s_b1[ql] = b1;
      s_b2[ql] = b2;
      s_row[ql] = row;
      s_na[ql] = na;
    }
  }
  __syncthreads();
  QTR_STAMP(STAMP_NN_FINISH, 2)
  // ---- (3) certification and the ordered append of the listed rows: one thread per query (wave 0)
  if (wave != 0) return;
  {
    const int t = blockIdx.x * NN_FINH_Q + lane;
    const bool valid = t < nq;
    const float c1 = s_b1[lane], c2 = s_b2[lane];
    int i1 = s_i1[lane];
    const int crow = s_row[lane];
    const float cna = valid ? s_na[lane] : 0.f;
    const float nb1 = (i1 >= 0) ? s_nbmax[lane] : 0.f;
    const float u = 5.9604645e-08f;
    const float d1 = fmaxf(cna + c1, 0.f) + 1.0f;  // d~ of the leader (|a|^2 is not inside b1)
    const float d2 = (c2 < INFINITY) ? fmaxf(cna + c2, 0.f) + 1.0f : d1;
    const float gap = u * (76.0f * cna + 108.0f * nb1 + 40.0f * (d1 + d2) + cadd) * 1.01f;  // (the f16 engine's own bound, round 5)
    if (unsafe) i1 = -1;
    // The chosen candidate's EXACT distance e1 is known here, and with it a bound that charges the leader's error as it is
    // instead of as it could be.  A base row b with d_ex(a,b) <= e1 has d(a,b) <= e1 (1 + 38u), and every base row obeys
    // d(a,b) >= |a|^2~ + v - 38u |a|^2 - 396u (header comment), so its filter value is at most
    //     T = e1 - |a|^2~ + u (38 |a|^2 + 38 e1 + 396)
    // (42 / 42 / 400 and 2 % on top below: the three float roundings of the expression lose <= 2u (e1 + |a|^2)).  Hence:
    // the second-best value of the whole cloud ABOVE T means every row but the leader r* is strictly farther than e1 in the
    // exact evaluation — also the other candidates, so the chosen one IS r* and is the exact arg-min: certified, whatever
    // the a-priori gap says.  And a listed row's re-check needs no base row above T.  With |b|^2 = 30000 the a-priori
    // window above the leader's value is u (76 |a|^2 + 108 |b|^2 + ...), this one u (42 |a|^2 + ...) + the leader's actual
    // error: what the near-identical descriptors of dense clouds were listed for, and evaluated for, shrinks accordingly.
    const float e1 = s_e1[lane];
    const float thr_e = (i1 >= 0) ? (e1 - cna) + u * (42.0f * cna + 42.0f * e1 + 0.5f * cadd) * 1.02f + 1e-30f : INFINITY;
    const bool certified = i1 >= 0 && (c2 == INFINITY || c2 - c1 > gap || c2 > thr_e);
    const bool listed = valid && !certified;
    if (valid) D.best[crow] = certified ? (u64)(u32)i1 : ~0ULL;
    const u64 bal = __ballot(listed);
    int base = 0;
    if (lane == 0 && bal) base = atomicAdd(V.mcounts + D.rc_slot, __popcll(bal));  // one atomic per workgroup
    base = __builtin_amdgcn_readfirstlane(base);
    if (listed) {
      const int slot = base + __popcll(bal & lanemask_lt());
      V.recheck_rows[slot] = crow;
      // a base row whose approximate (lower-bound) value exceeds this cannot be the exact arg-min; +inf when the slice
      // merge found nothing
      const float thr = (i1 >= 0) ? fminf(c1 + u * (76.0f * cna + 108.0f * nb1 + 80.0f * d1 + cadd) * 1.02f + 1e-30f, thr_e) : INFINITY;
      V.recheck_thr[slot] = thr;
      V.recheck_q[slot] = crow;  // where k_recheck_filter finds the row's f16 query fragments (row of D.queryH)
    }
  }
}

// Re-check of the listed rows (f16 engine): the matrix pipe again.  A listed row's exact arg-min is among the base rows
// whose filter value is at most the row's threshold (k_nn_finish_f16: the leader's value plus twice the rounding bound) —
// two or three near-ties, typically — so the listed rows go through the same 7-MFMA chain once more, this time
// comparing every entry with the row's threshold; the (row, base row) pairs that are not ABOVE it (a NaN entry — possible
// when the descriptors are outside the filter's range — passes) get the exact flann::L2 evaluation.
// Round 6: the unit of work is a WAVE with 128 listed rows — four column blocks of 32 stationary in registers, like the
// main kernel's waves — and a slice of the base tiles; the waves of the launch are dealt over (row group, slice) items on
// their own (nothing in here is shared between the waves of a workgroup).  Until then a wave held 32 rows: in-kernel
// stamps (tests/probe/recheck_stamps.py) put a dense-mode launch at 1455 clocks per 32 x 32 tile — one dependent chain
// of seven MFMAs per tile (224 clocks of matrix pipe) behind a base tile's L2 round trip, 4.7 % of the time in the exact
// evaluation — 366 us per item with 16 445 listed rows; four independent chains per base tile amortise the tile's loads
// and hide one another's latency.
// Round 5: every wave collects its passing pairs in a list of its own (LDS) and evaluates the list DENSELY — 64 pairs at
// a time, one per lane, straight from the descriptor tables (the running minimum of every listed row in LDS, ONE global
// atomic per row and item at the end) — whenever it fills, and when the slice is through.  Dense un-voxelised clouds are
// what this is for: a third of the rows of a 50 000-point scan of flat surfaces are listed (near-identical descriptors),
// each with hundreds of base rows under its threshold.  The round-4 form evaluated a pair where it turned up (a sparse
// loop over the 16 entries of every lane: a tenth of the lanes busy) with a global 64-bit atomic each; the voxelised
// scans (a thousand listed rows, two or three pairs each) never fill a list.  A tile that ALONE holds more pairs than a
// list (an infinite threshold: no leader, MC_UNSAFE) is evaluated where it stands, so the kernel is complete whatever
// passes — in the limit an exact scan of every pair.   grid (x, y, pairs): x * y workgroups of four waves per pair.
#define RCW_CAP 1024
#define RC_QB 4            // column blocks of 32 listed rows per wave
#define RC_ROWS (32 * RC_QB)
__device__ __forceinline__ float recheck_exact_dist(const float* __restrict__ a, const float* __restrict__ b) {
  float result = 0.f;  // flann::L2's accumulation order
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const float d0 = a[4 * g] - b[4 * g], d1 = a[4 * g + 1] - b[4 * g + 1], d2 = a[4 * g + 2] - b[4 * g + 2],
                d3 = a[4 * g + 3] - b[4 * g + 3];
    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  const float dt = a[32] - b[32];
  result += dt * dt;
  return result;
}
__device__ __forceinline__ void recheck_exact_pair(const float* __restrict__ a, const float* __restrict__ b, u64* best, int base_row) {
  const float result = recheck_exact_dist(a, b);
  if (result == result) atomicMin(best, ((u64)__float_as_uint(result) << 32) | (u32)base_row);  // (NaN never wins)
}
#define RC_STAMP(pt) if (dir == 0) { QTR_STAMP(STAMP_RECHECK, pt) } else { QTR_STAMP(STAMP_RECHECK1, pt) }
#define RC_SMALL_ROWS 2048  // up to here the one-block form
// (the kernel's body for QB column blocks of 32 listed rows per wave: 4 is the throughput form — dense clouds, thousands of
// listed rows —, 1 the latency form a scan pair's thousand rows take: with four blocks a wave of theirs has three tile visits
// behind 28 KB of staged fragments and a launch took 12.4 us, with one block eleven visits behind 7 KB: 10.8)
template <int QB>
__device__ __forceinline__ void recheck_items(const MatchView& V, const NnDir& D, const int dir, const int nrows, u32 (&s_wc)[4][RCW_CAP],
                                              u64 (&s_wbest)[4][32 * RC_QB], int (&s_wrow)[4][32 * RC_QB], int (&s_wctl)[4][4],
                                              h8* __restrict__ s_q /* [column block][chunk pair][lane]: the group's fragments */) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), col = lane & 31, half = lane >> 5;
  const u32 frag = (u32)half * 32u + (u32)col;
  const int ntiles = D.nb_pad / 32;
  const h8* __restrict__ baseH = (const h8*)D.baseH;
  const h8* __restrict__ queryH = (const h8*)D.queryH;
  const int* __restrict__ qcol = V.recheck_q;  // listed row -> its row of the query table (k_nn_finish_f16)
  const float* __restrict__ A = D.A;
  const float* __restrict__ B = dir ? V.fpfh_j : V.fpfh_i;
  const int nb = D.nb;
  const int qgroups = (nrows + (32 * QB) - 1) / (32 * QB);
  // work items = (group of 128 listed rows) x (slice of the base tiles), dealt over the launch's workgroups with as many
  // slices per group as there are workgroups for it; the four waves of a workgroup take a quarter of the slice each and
  // share the group's 28 KB of query fragments through LDS — fetched ONCE per workgroup: with every wave fetching its own
  // (the first round-6 form) a thousand listed rows cost 58 MB of fragment traffic per launch, 6 us of the headline's
  // launch before its first MFMA.  A thousand listed rows (direction 1 of a scan pair: nine groups) then take three tiles
  // per wave — one round trip — and the 16 k rows of a dense cloud a hundred.
  const int nwg = gridDim.x * gridDim.y, wid = blockIdx.y * gridDim.x + blockIdx.x;
  const int nsl = max(1, min((ntiles + 3) / 4, nwg / qgroups)), per_wg = (ntiles + nsl - 1) / nsl, per = (per_wg + 3) / 4;
  u32* __restrict__ wc = &s_wc[wave][0];
  u64* __restrict__ wbest = &s_wbest[wave][0];
  int* __restrict__ wrow = &s_wrow[wave][0];
  int* __restrict__ wctl = &s_wctl[wave][0];  // [0] places reserved, [1] first tile that did not fit, [2] entries that did
  for (int item = wid; item < qgroups * nsl; item += nwg) {
    const int qg = item / nsl, T0 = (item - qg * nsl) * per_wg, T1 = min(ntiles, T0 + per_wg);
    const int t0 = T0 + wave * per, t1 = min(T1, t0 + per);
    // everything that does not wait for the fragments goes out first: the wave's first base tile, its thresholds (one
    // round trip beside the two of the fragment staging instead of behind them: a launch on a thousand listed rows is
    // five dependent round trips and three tiles)
    h8 m0[7], m1[7];
    float thr[QB];
#pragma unroll
    for (int j = 0; j < 7; ++j) m0[j] = baseH[((size_t)min(t0, ntiles + 1) * NNH_CHUNKS + 2 * j) * 32 + frag];
#pragma unroll
    for (int cb = 0; cb < QB; ++cb) {
      const int slot = qg * (32 * QB) + cb * 32 + col;
      thr[cb] = (slot < nrows) ? V.recheck_thr[slot] * (NNH_S * NNH_S) + 0.0f : 0.f;  // (+ 0: never -0, see tile_mask)
    }
    __syncthreads();  // (the previous item's fragments have been read)
    for (int e = threadIdx.x; e < QB * 7 * 64; e += 256) {
      const int cb = e / (7 * 64), m = (e - cb * 7 * 64) >> 6, l = e & 63;
      const int slot = qg * (32 * QB) + cb * 32 + (l & 31);
      const int qc = (slot < nrows) ? qcol[slot] : 0;  // (rows past the list: any column; nothing of theirs is listed)
      s_q[e] = queryH[((size_t)(qc >> 5) * NNH_CHUNKS + 2 * m + (l >> 5)) * 32 + (qc & 31)];
    }
    __syncthreads();
    if (t0 < t1) {  // (uniform per wave: the last waves of a short slice have nothing)
    h8 q[QB][7];
    bool live[QB];
#pragma unroll
    for (int cb = 0; cb < QB; ++cb) {
      const int slot = qg * (32 * QB) + cb * 32 + col;
      live[cb] = slot < nrows;
#pragma unroll
      for (int m = 0; m < 7; ++m) q[cb][m] = s_q[(cb * 7 + m) * 64 + lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the previous item's reads of the wave's LDS are through)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = lane; e < (32 * QB); e += 64) {
      const int slot = qg * (32 * QB) + e;
      wrow[e] = (slot < nrows) ? V.recheck_rows[slot] : -1;
      wbest[e] = ~0ULL;
    }
    if (lane == 0) {
      wctl[0] = 0;
      wctl[1] = 0x7fffffff;
      wctl[2] = 0x7fffffff;
    }
    int wn = 0;  // (uniform) pairs in the wave's list
#ifdef QTR_NN_TIMING
    long long rc_drain_clk = 0, rc_drain_pairs = 0, rc_drains = 0;  // (tests/probe/recheck_stamps.py: the exact evaluation's share)
#define RC_DRAIN_T0 const long long rc_t0_ = clock64(); rc_drain_pairs += wn; ++rc_drains;
#define RC_DRAIN_T1 rc_drain_clk += clock64() - rc_t0_;
#else
#define RC_DRAIN_T0
#define RC_DRAIN_T1
#endif
    // the wave's list, densely: lane e takes pair e, both rows straight from the descriptor tables (a listed row's 132
    // bytes are read by every pair it takes part in: they stay in the unit's cache)
    auto drain = [&]() __attribute__((always_inline)) {
      RC_DRAIN_T0
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the list is the wave's own: LDS operations of one wave
      __builtin_amdgcn_wave_barrier();                        // complete in order)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int e = lane; e < wn; e += 64) {
        const u32 en = wc[e];
        const int c = (int)(en >> 20), brow = (int)(en & 0xfffffu);
        if (brow < nb) {  // (not a pad row of the last tile)
          const float result = recheck_exact_dist(A + (size_t)wrow[c] * 33, B + (size_t)brow * 33);
          if (result == result) atomicMin(&wbest[c], ((u64)__float_as_uint(result) << 32) | (u32)brow);  // (NaN never wins)
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      wn = 0;
      RC_DRAIN_T1
    };
    auto load = [&](h8 (&m)[7], int t) __attribute__((always_inline)) {
      t = min(t, ntiles + 1);  // (the table is padded by two tiles: running ahead of the slice is harmless)
#pragma unroll
      for (int j = 0; j < 7; ++j) m[j] = baseH[((size_t)t * NNH_CHUNKS + 2 * j) * 32 + frag];
    };
    // which of the lane's 4 x 16 entries of tile t pass: bit 16 cb + r.  Two column blocks' chains at a time (independent
    // accumulators: the matrix pipe works on one while the other's result is still on its way)
    auto tile_mask = [&](const h8 (&m)[7]) __attribute__((always_inline)) -> u64 {
      u64 pass = 0;
      const f32x16 zero16 = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      // "entry r is above the threshold" is the SIGN of thr - acc[r], shifted into the mask by one v_alignbit: two vector
      // instructions per entry and no compare whose mask has to travel through a scalar register (the first form — compare,
      // select, or — was 3.5 with its wait states and kept the sweep at 3280 clocks per visit against 896 of matrix pipe).
      // acc = thr gives +0 (thr is never -0): not above.  A threshold of +inf (no leader; descriptors outside the filter's
      // range, where an entry may be inf or NaN and inf - inf has no sign to speak of) passes everything without looking.
      auto fails = [&](const f32x16& acc, float th) __attribute__((always_inline)) -> u32 {
        u32 f = 0;  // bit 15 - r: entry r FAILS
#pragma unroll
        for (int r = 0; r < 16; ++r) f = __builtin_amdgcn_alignbit(f, __float_as_uint(th - acc[r]), 31);
        return (th == INFINITY) ? 0xffffu : (__brev(~f) >> 16);  // bit r: entry r passes
      };
      if constexpr (QB == 1) {
        f32x16 acc = zero16;
#pragma unroll
        for (int j = 0; j < 7; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(m[j], q[0][j], acc, 0, 0, 0);
        pass = live[0] ? (u64)fails(acc, thr[0]) : 0ULL;
      } else {
#pragma unroll
        for (int cb = 0; cb < QB; cb += 2) {
          f32x16 acc0 = zero16, acc1 = zero16;
#pragma unroll
          for (int j = 0; j < 7; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(m[j], q[cb][j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(m[j], q[cb + 1][j], acc1, 0, 0, 0);
          }
          const u32 p0 = live[cb] ? fails(acc0, thr[cb]) : 0u, p1 = live[cb + 1] ? fails(acc1, thr[cb + 1]) : 0u;
          pass |= (u64)(p0 | (p1 << 16)) << (16 * cb);
        }
      }
      return pass;
    };
    auto append = [&](u64 pass, int t, int at) __attribute__((always_inline)) {
      while (pass) {
        const int e = __ffsll((unsigned long long)pass) - 1, cb = e >> 4, r = e & 15;
        pass &= pass - 1;
        wc[at++] = ((u32)(cb * 32 + col) << 20) | (u32)(t * 32 + 8 * (r >> 2) + 4 * half + (r & 3));  // (row < 2^20: QTR_NN_MAX_ROWS)
      }
    };
    RC_STAMP(1)
    // ---- the sweep as the voxelised scans need it: the few passing pairs appended to the list; a list that would overflow
    // ends it (a lane with pairs reserves their places with ONE LDS atomic — no vote, no scan on the path of the tiles that
    // hold nothing; the first reservation that does not fit marks its tile, and what was reserved before it is a prefix of
    // the list: reservations only grow)
    auto hot = [&](const h8 (&m)[7], int t) __attribute__((always_inline)) {
      const u64 pass = tile_mask(m);
      if (pass) {
        const int cnt = __popcll(pass), at = atomicAdd(&wctl[0], cnt);
        if (at + cnt <= RCW_CAP) {
          append(pass, t, at);
        } else {
          atomicMin(&wctl[1], t);
          atomicMin(&wctl[2], at);
        }
      }
    };
    for (int t = t0; t < t1; t += 2) {  // (m0 holds tile t0 since the top of the item)
      load(m1, t + 1);
      hot(m0, t);
      if (t + 1 < t1) {
        load(m0, t + 2);
        hot(m1, t + 1);
      }
      if (__builtin_amdgcn_readfirstlane(wctl[1]) != 0x7fffffff) break;  // a list overflowed: the dense loop takes over
    }
    RC_STAMP(2)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int ovf = (wctl[1] != 0x7fffffff) ? __builtin_amdgcn_readfirstlane(wctl[1]) : -1;  // (uniform) first tile whose pairs did not fit
    wn = __builtin_amdgcn_readfirstlane(min(min(wctl[0], wctl[2]), RCW_CAP));
    if (wn > 0) drain();
    // ---- ... and as the dense clouds need it, from the tile that overflowed on: one tile ahead, the list evaluated
    // whenever the next tile's pairs would not fit (and when the slice is through)
    if (ovf >= 0) {
      load(m0, ovf);
      for (int t = ovf; t <= t1; ++t) {
        u64 pass = 0;
        int tot = 0, at = 0;
        if (t < t1) {
          load(m1, t + 1);
          pass = tile_mask(m0);
          at = wave_excl_scan_i32(__popcll(pass), &tot);
        }
        if (t == t1 || wn + tot > RCW_CAP) {
          if (wn > 0) drain();
          if (t == t1) break;
        }
        if (tot > RCW_CAP) {  // more than a list holds in ONE tile (infinite thresholds): evaluated where they stand
          while (pass) {
            const int e = __ffsll((unsigned long long)pass) - 1, cb = e >> 4, r = e & 15;
            pass &= pass - 1;
            const int brow = t * 32 + 8 * (r >> 2) + 4 * half + (r & 3), row = wrow[cb * 32 + col];
            if (brow < nb) recheck_exact_pair(A + (size_t)row * 33, B + (size_t)brow * 33, &D.best[row], brow);
          }
        } else if (tot > 0) {
          append(pass, t, wn + at);
          wn += tot;
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) m0[j] = m1[j];
      }
    }
    RC_STAMP(3)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int e = lane; e < (32 * QB); e += 64) {
      const u64 b = wbest[e];
      const int row = wrow[e];
      if (row >= 0 && b != ~0ULL) atomicMin(&D.best[row], b);
    }
    RC_STAMP(4)
#ifdef QTR_NN_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 32 && blockIdx.y == 0 && blockIdx.z == 0) {
      unsigned long long(*g)[2] = g_stamp[dir == 0 ? STAMP_RECHECK : STAMP_RECHECK1][blockIdx.x];
      g[5][0] = (unsigned long long)rc_drain_clk;
      g[5][1] = (unsigned long long)rc_drain_pairs;
      g[6][0] = (unsigned long long)rc_drains;
      g[6][1] = (unsigned long long)(t1 - t0);
      g[7][0] = (unsigned long long)nrows;
      g[7][1] = (unsigned long long)(qgroups * nsl);
    }
#endif
    }  // t0 < t1
  }
#undef RC_DRAIN_T0
#undef RC_DRAIN_T1
}

template <bool EXT>
__global__ __launch_bounds__(256, 2) void k_recheck_filter(ViewExt<MatchView> x, MatchView one, int dir) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const NnDir& D = V.d[dir];
  const int nrows = V.mcounts[D.rc_slot];
  if (nrows <= 0) return;
  RC_STAMP(0)
  __shared__ u32 s_wc[4][RCW_CAP];      // per wave: passing pairs, (listed row of the wave's 128) << 20 | base row
  __shared__ u64 s_wbest[4][RC_ROWS];   // per wave: packed (exact distance bits << 32 | base row) minimum of each listed row
  __shared__ int s_wrow[4][RC_ROWS];    // per wave: its listed rows (source / target row ids; -1 past the end of the list)
  __shared__ int s_wctl[4][4];          // per wave: the list's reservation counter and overflow marks (see the sweep)
  __shared__ __attribute__((aligned(16))) h8 s_q[RC_QB * 7 * 64];
  if (nrows <= RC_SMALL_ROWS)
    recheck_items<1>(V, D, dir, nrows, s_wc, s_wbest, s_wrow, s_wctl, s_q);
  else
    recheck_items<RC_QB>(V, D, dir, nrows, s_wc, s_wbest, s_wrow, s_wctl, s_q);
}

#undef RC_STAMP

// exact re-decision of the listed rows (list length read on the device).  A workgroup takes EIGHT listed rows at a
// time: their 33 values (and the -2x copies the approximate chain needs) sit in LDS, where every lane reads them with
// broadcast 16-byte loads, and each of the four waves streams a quarter of the base slice — 34 coalesced loads per
// base row serve all eight queries, so the re-check moves an eighth of the L2 traffic of a row-at-a-time scan (it was
// L2-bound: one pass over the k-major base table per listed row).
// The test is two-staged: a 33-term fma chain gives the approximate (lower-bound, like the MFMA result) value, and
// only base rows under the row's threshold can be the exact arg-min — those few are evaluated with the exact
// flann::L2 arithmetic.  The exact winner always passes the filter, so the tables are unchanged.  Hidden duplicate
// rows carry 1e30 and are skipped: their lower-indexed twin has the same exact distance and wins the tie.
// Candidates fold into an LDS minimum per row; one packed 64-bit atomicMin per (row, slice) publishes it.
// grid (groups, slices, pairs).
#define XR 8
template <bool EXT>
__global__ __launch_bounds__(256, 2) void k_nn_exact_rows(ViewExt<MatchView> x, MatchView one, int dir) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const NnDir& D = V.d[dir];
  const int nrows = V.mcounts[D.rc_slot];
  if (nrows <= 0) return;
  __shared__ __attribute__((aligned(16))) float s_m2a[33][XR];  // [k][row]: -2 * a
  __shared__ __attribute__((aligned(16))) float s_a[XR][36];    // [row][k]
  __shared__ float s_thr[XR];
  __shared__ int s_row[XR];
  __shared__ u64 s_best[XR];  // packed (exact distance bits << 32 | base row) minima of the group's rows
  __shared__ int s_span[2];
  const float* __restrict__ A = D.A;
  const float* __restrict__ BT = D.baseTb;   // columns in norm-bin order (k_norm_bins): a listed row scans one span
  const int* __restrict__ brow = D.brow;
  const int nb_pad = D.nb_pad;
  const int* __restrict__ rows = V.recheck_rows;
  const float* __restrict__ thr = V.recheck_thr;
  const int2* __restrict__ span = V.recheck_span;
  u64* __restrict__ best = D.best;
  const int lane = qk_lane(), wave = threadIdx.x >> 6;
  const int ngroups = (nrows + XR - 1) / XR;
  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    __syncthreads();
    for (int e = threadIdx.x; e < XR * 33; e += 256) {
      const int r = e / 33, k = e - r * 33;
      const int ri = g * XR + r;
      const float val = (ri < nrows) ? A[(size_t)rows[ri] * 33 + k] : 0.f;
      s_a[r][k] = val;
      s_m2a[k][r] = -2.0f * val;
    }
    if (threadIdx.x < XR) {
      const int ri = g * XR + threadIdx.x;
      s_row[threadIdx.x] = (ri < nrows) ? rows[ri] : -1;
      s_thr[threadIdx.x] = (ri < nrows) ? thr[ri] : -INFINITY;
      s_best[threadIdx.x] = ~0ULL;
    }
    if (threadIdx.x == 64) {  // union of the group's spans (the list is nearly sorted by norm: they overlap)
      int lo = 0x7fffffff, hi = 0;
      for (int r = 0; r < XR; ++r) {
        const int ri = g * XR + r;
        if (ri < nrows) {
          const int2 sp = span[ri];
          lo = min(lo, sp.x);
          hi = max(hi, sp.y);
        }
      }
      s_span[0] = lo;
      s_span[1] = hi;
    }
    __syncthreads();
    const int per = (max(s_span[1] - s_span[0], 0) + gridDim.y - 1) / gridDim.y;
    const int b0 = s_span[0] + blockIdx.y * per, b1 = min(s_span[1], b0 + per);
    for (int b = b0 + wave * 64 + lane; b < b1; b += 256) {
      // (compiler barrier: without it the 66 loop-invariant 16-byte LDS reads below are hoisted out of the loop and
      // held in 264 registers — spills; they are meant to be re-read, broadcast, every iteration)
      asm volatile("" ::: "memory");
      float v[34];
      const u32 bu = (u32)b;  // uniform row pointer + 32-bit lane offset: scalar-base loads, one offset register
#pragma unroll
      for (int k = 0; k < 34; ++k) v[k] = (BT + (size_t)k * nb_pad)[bu];
      float ap[XR];
#pragma unroll
      for (int r = 0; r < XR; ++r) ap[r] = v[33];  // scaled |b|^2
#pragma unroll
      for (int k = 0; k < 33; ++k) {
        if (k % 4 == 0) asm volatile("" ::: "memory");  // bounds how many LDS reads are in flight (registers)
        const float4 ma = *(const float4*)&s_m2a[k][0], mb = *(const float4*)&s_m2a[k][4];
        ap[0] = fmaf(ma.x, v[k], ap[0]);
        ap[1] = fmaf(ma.y, v[k], ap[1]);
        ap[2] = fmaf(ma.z, v[k], ap[2]);
        ap[3] = fmaf(ma.w, v[k], ap[3]);
        ap[4] = fmaf(mb.x, v[k], ap[4]);
        ap[5] = fmaf(mb.y, v[k], ap[5]);
        ap[6] = fmaf(mb.z, v[k], ap[6]);
        ap[7] = fmaf(mb.w, v[k], ap[7]);
      }
      u32 pass = 0;
#pragma unroll
      for (int r = 0; r < XR; ++r) pass |= (ap[r] <= s_thr[r]) ? (1u << r) : 0u;
      // rare (except among degenerate near-identical descriptors): exact flann::L2 order, one row at a time (a loop, not
      // eight predicated copies: those would keep eight LDS rows in registers)
      while (pass) {
        const int r = __ffs((int)pass) - 1;
        pass &= pass - 1;
        const float* av = s_a[r];
        float result = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) {
          const float4 a4 = *(const float4*)(av + 4 * q4);
          const float d0 = a4.x - v[4 * q4], d1 = a4.y - v[4 * q4 + 1], d2 = a4.z - v[4 * q4 + 2],
                      d3 = a4.w - v[4 * q4 + 3];
          result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        const float dt = av[32] - v[32];
        result += dt * dt;
        atomicMin(&s_best[r], ((u64)__float_as_uint(result) << 32) | (u32)brow[b]);
      }
    }
    __syncthreads();
    if (threadIdx.x < XR && s_row[threadIdx.x] >= 0 && s_best[threadIdx.x] != ~0ULL)
      atomicMin(&best[s_row[threadIdx.x]], s_best[threadIdx.x]);
  }
}

// one launch instead of a handful of memsets/fills: counters, tuple-test flags, source->target table, NN tables,
// dedup tables.  grid (g, 1, pairs)
// (tail_lookback — the look-back over per-workgroup counts the multi-workgroup compactions of the tail use — lives in common.h)
template <bool EXT>
__global__ __launch_bounds__(256) void k_match_init(ViewExt<MatchView> x, MatchView one, int clear_tables, int* zero_words,
                                                    int n_zero) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
  // (a caller's own clean slate rides along: the whole-path driver's solver state — a 4.6 us launch of its own until round 5)
  if (zero_words && blockIdx.x == gridDim.x - 1 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < n_zero; i += blockDim.x) zero_words[i] = 0;
  if (gid < 16) V.mcounts[gid] = (gid == MC_SWAPPED) ? V.swapped : (gid == MC_NQ0) ? V.n_small : 0;
  const int pv = V.tuple ? 0 : 1;
  const int np = V.crosscheck ? V.n_small : V.n_small + V.n_large;
  for (int i = gid; i < np; i += gsz) V.passed[i] = pv;
  if (!V.crosscheck)
    for (int i = gid; i < V.ns; i += gsz) V.nc_cnt[i] = V.nc_fill[i] = 0;
  for (int i = gid; i < V.ns; i += gsz) V.tgt_of_src[i] = -1;
  for (int i = gid; i < V.n_small; i += gsz) V.best_small[i] = ~0ULL;
  for (int i = gid; i < V.n_large; i += gsz) V.best_large[i] = ~0ULL;
  if (clear_tables)  // (0: the FPFH chain cleared and filled them — frontend.hip, desc_prep)
    for (int i = gid; i <= V.dd_mask; i += gsz) {
      V.table_i[i] = ~0ULL;
      V.table_j[i] = ~0ULL;
    }
  for (int i = gid; i < 2 * TAIL_MAXWG; i += gsz) V.scan[i] = 0;  // look-back words of k_cross_multi / k_pairs_multi
}

// Rows of the larger cloud that the (final) first direction points at, in the cloud's NORM-BIN order (k_norm_bins): a
// bit set in LDS (one workgroup per pair), then a scan over the bin-ordered rows.  (The order only matters to the
// span re-check of direction 1 (k_nn_exact_rows), whose listed rows then share spans; the cross-check reads the NN
// tables by row.  bin_order = 0 — the f16 engine, whose re-check sweeps every tile — lists them by row.)
// grid (1, 1, pairs), 1024 threads.
template <bool EXT>
__global__ __launch_bounds__(1024) void k_hit_compact(ViewExt<MatchView> x, MatchView one, int bin_order, int nn_X) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  extern __shared__ u32 hit_bits[];  // ceil(n_large / 32) words
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = V.n_large, nw = (n + 31) / 32, ns = V.n_small;
  QTR_STAMP(STAMP_HIT_COMPACT, 0)
  for (int w = tid; w < nw; w += 1024) hit_bits[w] = 0u;
  __syncthreads();
  for (int j0 = tid; j0 < ns; j0 += 16 * 1024) {  // one workgroup, a latency chain: sixteen loads in flight per thread
    u64 b[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) b[q] = (j0 + q * 1024 < ns) ? V.best_small[j0 + q * 1024] : 0ULL;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (j0 + q * 1024 < ns) {
        const int i = (b[q] == ~0ULL) ? 0 : (int)(u32)b[q];
        atomicOr(&hit_bits[i >> 5], 1u << (i & 31));
      }
  }
  __syncthreads();
  QTR_STAMP(STAMP_HIT_COMPACT, 1)
  // positions (of the bin order, or plain rows: then whole words of the bit set) are dealt in contiguous runs so that
  // the output keeps that order
  const int items = bin_order ? n : nw;
  const int per = (items + 1023) / 1024;
  const int p0 = min(items, tid * per), p1 = min(items, p0 + per);
  int cnt = 0;
  if (bin_order) {
    for (int p = p0; p < p1; ++p) {
      const int r = V.nb_row_i[p];
      cnt += (hit_bits[r >> 5] >> (r & 31)) & 1u;
    }
  } else {
    for (int w = p0; w < p1; ++w) cnt += __popc(hit_bits[w]);
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
  if (bin_order) {
    for (int p = p0; p < p1; ++p) {
      const int r = V.nb_row_i[p];
      if ((hit_bits[r >> 5] >> (r & 31)) & 1u) V.hit_rows[run++] = r;
    }
  } else {
    for (int w = p0; w < p1; ++w) {
      u32 bits = hit_bits[w];
      while (bits) {
        V.hit_rows[run++] = 32 * w + (__ffs((int)bits) - 1);
        bits &= bits - 1;
      }
    }
  }
  if (tid == 0) {
    V.mcounts[MC_NHIT] = total;
    // (single pair, nn_X = workgroups of the k_nn_f16 launch that asks for these rows: its plan, see nn_plan_single)
    if (nn_X > 0) V.mcounts[MC_NSPLIT1] = nn_plan_single((total + NN_QPB - 1) / NN_QPB, V.d[1].nb_pad / 32, nn_X);
  }
  QTR_STAMP(STAMP_HIT_COMPACT, 2)
}

// query columns of the hit rows, gathered into a compact k-major table (pad columns: zeros with the constant-1 row,
// never read back).  grid (pad_large_max / 256, 1, pairs)
template <bool EXT>
__global__ __launch_bounds__(256) void k_hit_gather(ViewExt<MatchView> x, MatchView one, int f32_tables) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int nhit = V.mcounts[MC_NHIT];
  const int padded = (nhit + NN_QPB - 1) / NN_QPB * NN_QPB;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= padded) return;
  const int pad = V.pad_large;
  if (c < nhit) {
    const int r = V.hit_rows[c];
    V.norms_c[c] = V.norms_i[r];
    if (f32_tables) {
#pragma unroll
      for (int k = 0; k < 34; ++k) V.queryT_c[(size_t)k * pad + c] = V.queryT_i[(size_t)k * pad + r];
    } else {
#pragma unroll
      for (int ch = 0; ch < NNH_CHUNKS; ++ch)
        V.queryH_c[((size_t)(c >> 5) * NNH_CHUNKS + ch) * 32 + (c & 31)] = V.queryH_i[((size_t)(r >> 5) * NNH_CHUNKS + ch) * 32 + (r & 31)];
    }
  } else {
    V.norms_c[c] = 0.f;
    if (f32_tables) {
#pragma unroll
      for (int k = 0; k < 33; ++k) V.queryT_c[(size_t)k * pad + c] = 0.f;
      V.queryT_c[(size_t)33 * pad + c] = 1.0f;
    } else {
#pragma unroll
      for (int ch = 0; ch < NNH_CHUNKS; ++ch) V.queryH_c[((size_t)(c >> 5) * NNH_CHUNKS + ch) * 32 + (c & 31)] = make_uint4(0, 0, 0, 0);
    }
  }
}

// unpack both NN tables and evaluate the mutual-NN test in one pass (rows of the larger cloud that were not asked
// hold ~0 and decode to -1: nobody points at them)
template <bool EXT>
__global__ __launch_bounds__(256) void k_cross_flags2(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
  for (int j = gid; j < V.n_small; j += gsz) {
    const u64 b = V.best_small[j];
    V.nn_of_small[j] = (b == ~0ULL) ? 0 : (int)(u32)b;
  }
  for (int i = gid; i < V.n_large; i += gsz) {
    const u64 b = V.best_large[i];
    const int j = (b == ~0ULL) ? 0 : (int)(u32)b;
    V.nn_of_large[i] = (b == ~0ULL) ? -1 : j;
    const u64 bs = V.best_small[j];
    const int back = (bs == ~0ULL) ? 0 : (int)(u32)bs;
    V.flags[i] = V.crosscheck ? ((b != ~0ULL && back == i) ? 1 : 0) : ((b != ~0ULL) ? 1 : 0);  // mutual pair / asked row
  }
}

// exclusive scans (one workgroup per pair); out has n+1 entries
template <bool EXT>
__global__ __launch_bounds__(1024) void k_scan_flags(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  d_block_scan(V.flags, V.scan, V.n_large, [](int x) { return x; });
}
template <bool EXT>
__global__ __launch_bounds__(1024) void k_scan_nonneg(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  d_block_scan(V.tgt_of_src, V.scan, V.ns, [](int x) { return x >= 0 ? 1 : 0; });
}

// last matcher kernel: counters for the host, no copy launches.  Called by threads 0..47 (one wavefront).
__device__ __forceinline__ void match_mail(const MatchView& V, int t, int total, int ntuple) {
  int* mail = V.mail;
  if (t < 16)
    mail_store_line(mail + MAIL_MATCH, t, (t == MC_NCORR) ? total : (t == MC_NTUPLE) ? ntuple : V.mcounts[t], V.seq);
  else if (t < 32)
    mail_store_line(mail + MAIL_CNT0, t - 16, V.counts0[t - 16], V.seq);
  else
    mail_store_line(mail + MAIL_CNT1, t - 32, V.counts1[t - 32], V.seq);
  __threadfence_system();  // threads 0..47 are one wavefront: the stores above are acknowledged before ...
  if (t == 0) mail[MAIL_SEQ_MATCH] = V.seq;  // ... the word the host is watching changes
}

template <bool EXT>
__global__ void k_corr_compact2(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int ns = V.ns;
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < ns; s += gridDim.x * blockDim.x) {
    const int t = V.tgt_of_src[s];
    if (t >= 0) {
      V.corr[2 * V.scan[s]] = s;
      V.corr[2 * V.scan[s] + 1] = t;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) V.mcounts[MC_NCORR] = V.scan[ns];
  if (V.mail && blockIdx.x == 0 && threadIdx.x < 48) match_mail(V, threadIdx.x, V.scan[ns], V.mcounts[MC_NTUPLE]);
}

template <bool EXT>
__global__ void k_cross_compact(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V.n_large; i += gridDim.x * blockDim.x) {
    if (V.flags[i]) {
      V.cross_i[V.scan[i]] = i;
      V.cross_j[V.scan[i]] = V.nn_of_large[i];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) V.mcounts[MC_NCROSS] = V.scan[V.n_large];
}

// tuple test (reference feature_matcher.cc:187-247); trial t draws qm_rand_u32(seed, 3t+k) % ncorr
template <bool EXT>
__global__ __launch_bounds__(256) void k_tuple(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  if (!V.tuple) return;
  const int ncorr = V.mcounts[MC_NCROSS];
  if (ncorr <= 0) return;
  const float4* __restrict__ pts_i = V.vox_i;
  const float4* __restrict__ pts_j = V.vox_j;
  const int* __restrict__ cross_i = V.cross_i;
  const int* __restrict__ cross_j = V.cross_j;
  const float scale = V.tuple_scale;
  const u64 seed = V.seed;
  const long long trials = (long long)ncorr * 100;
  const float mix = V.mean_i[0], miy = V.mean_i[1], miz = V.mean_i[2];
  const float mjx = V.mean_j[0], mjy = V.mean_j[1], mjz = V.mean_j[2];
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < trials;
       t += (long long)gridDim.x * blockDim.x) {
    const int r0 = (int)(qm_rand_u32(seed, 3ULL * (u64)t) % (u32)ncorr);
    const int r1 = (int)(qm_rand_u32(seed, 3ULL * (u64)t + 1ULL) % (u32)ncorr);
    const int r2 = (int)(qm_rand_u32(seed, 3ULL * (u64)t + 2ULL) % (u32)ncorr);
    float pi[3][3], pj[3][3];
    const int rr[3] = {r0, r1, r2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 pa = pts_i[cross_i[rr[k]]], pb = pts_j[cross_j[rr[k]]];
      pi[k][0] = pa.x - mix;
      pi[k][1] = pa.y - miy;
      pi[k][2] = pa.z - miz;
      pj[k][0] = pb.x - mjx;
      pj[k][1] = pb.y - mjy;
      pj[k][2] = pb.z - mjz;
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
      V.passed[r0] = 1;
      V.passed[r1] = 1;
      V.passed[r2] = 1;
    }
  }
}

// passed cross pairs -> tgt_of_src (each source index occurs at most once after the cross-check)
template <bool EXT>
__global__ void k_scatter_pairs(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int nc = V.mcounts[MC_NCROSS];
  const int swapped = V.swapped;
  int local = 0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    if (V.passed[c]) {
      const int i = V.cross_i[c], j = V.cross_j[c];
      const int s = swapped ? j : i, t = swapped ? i : j;
      V.tgt_of_src[s] = t;
      ++local;
    }
  }
  local = wave_sum_i32(local);
  if (qk_lane() == 0 && local) atomicAdd(&V.mcounts[MC_NTUPLE], local);
}

// ---- use_crosscheck = 0 (reference feature_matcher.cc:124-181: "Skipping Cross Check"): the list handed to the tuple
// test is corres_ij (every hit row i of the larger cloud with ITS nearest neighbour, ascending i) followed by corres_ji
// (every row j of the smaller cloud with its nearest neighbour, ascending j); the survivors are un-swapped, sorted and
// made unique.  A source index can now carry several targets, so the tail builds per-source target lists instead of
// the one-target-per-source table of the cross-checked path.  Single pair only (the batched path keeps the cross-check).
template <bool EXT>
__global__ __launch_bounds__(256) void k_nc_list(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  // corres_ij in ascending i: flags (row asked in direction 1 = hit) and their scan come from k_cross_flags2 / k_scan_flags
  const int nhit = V.scan[V.n_large], n = nhit + V.n_small;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < V.n_large + V.n_small; e += gridDim.x * blockDim.x) {
    if (e < V.n_large) {
      if (V.flags[e]) {
        V.cross_i[V.scan[e]] = e;
        V.cross_j[V.scan[e]] = V.nn_of_large[e];
      }
    } else {
      const int j = e - V.n_large;
      V.cross_i[nhit + j] = V.nn_of_small[j];
      V.cross_j[nhit + j] = j;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) V.mcounts[MC_NCROSS] = n;
}
// pass 0: count the survivors of every source index; pass 1: drop their targets into the source's slice
template <bool EXT>
__global__ __launch_bounds__(256) void k_nc_scatter(ViewExt<MatchView> x, MatchView one, int pass) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int nc = V.mcounts[MC_NCROSS], swapped = V.swapped;
  int local = 0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    if (!V.passed[c]) continue;
    const int i = V.cross_i[c], j = V.cross_j[c];
    const int s = swapped ? j : i, t = swapped ? i : j;
    if (pass == 0) {
      atomicAdd(&V.nc_cnt[s], 1);
      ++local;
    } else {
      V.nc_list[V.nc_off[s] + atomicAdd(&V.nc_fill[s], 1)] = t;
    }
  }
  if (pass == 0) {
    local = wave_sum_i32(local);
    if (qk_lane() == 0 && local) atomicAdd(&V.mcounts[MC_NTUPLE], local);
  }
}
template <bool EXT>
__global__ __launch_bounds__(1024) void k_nc_scan(ViewExt<MatchView> x, MatchView one, int which) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  d_block_scan(which == 0 ? V.nc_cnt : V.nc_fill, which == 0 ? V.nc_off : V.scan, V.ns, [](int v) { return v; });
}
// Every source sorts its target list, drops repeats and leaves the distinct count in nc_fill.  Almost every list is a
// handful of entries (one thread, insertion sort) — but a list's length is the source's in-degree in the other cloud's
// nearest-neighbour table, and a descriptor that many points of the other cloud are nearest to (flat ground: one pair of
// the bench pool has a source with thousands of targets; identical descriptors: ALL of them) made that thread's
// quadratic loop the registration's longest kernel by a factor of ten (13 ms, profiles/r6_ab.txt section 15).  Lists above
// NC_SHORT entries are therefore left to the whole workgroup: the targets are indices below nt, so the list is laid down
// as a bit set in LDS (NC_RANGE targets at a time) and read back in order — sorted and unique in O(k + nt / 32) — via a
// scratch slice (cross_i: dead since k_nc_scatter) because a list longer than one range is still being read while its
// head is written.
#define NC_SHORT 32
#define NC_RANGE 16384
__device__ __forceinline__ int nc_unique_long(int* __restrict__ l, int* __restrict__ out, int k, int n_other,
                                              u32* __restrict__ s_bits, int* __restrict__ s_wt) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int u = 0;  // (workgroup-uniform)
  for (int lo = 0; lo < n_other; lo += NC_RANGE) {
    const int span = min(NC_RANGE, n_other - lo), words = (span + 31) >> 5;
    for (int w = tid; w < words; w += 256) s_bits[w] = 0u;
    __syncthreads();
    for (int a = tid; a < k; a += 256) {
      const int v = l[a] - lo;
      if ((unsigned)v < (unsigned)span) atomicOr(&s_bits[v >> 5], 1u << (v & 31));
    }
    __syncthreads();
    for (int w0 = 0; w0 < words; w0 += 256) {
      const int w = w0 + tid;
      u32 bits = (w < words) ? s_bits[w] : 0u;
      int tot;
      const int ex = wave_excl_scan_i32(__popc(bits), &tot);
      if (lane == 0) s_wt[wave] = tot;
      __syncthreads();
      int pos = u + ex;
      for (int q = 0; q < wave; ++q) pos += s_wt[q];
      while (bits) {
        out[pos++] = lo + (w << 5) + __ffs(bits) - 1;
        bits &= bits - 1u;
      }
      u += (s_wt[0] + s_wt[1]) + (s_wt[2] + s_wt[3]);
      __syncthreads();
    }
  }
  for (int a = tid; a < u; a += 256) l[a] = out[a];  // (out was written by other threads: the loop's last barrier orders it)
  return u;
}
template <bool EXT>
__global__ __launch_bounds__(256) void k_nc_unique(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  __shared__ u32 s_bits[NC_RANGE / 32];
  __shared__ int s_long[256], s_nlong, s_wt[4];
  const int ns = V.ns, nt = V.nt;
  for (int base = blockIdx.x * 256; base < ns; base += gridDim.x * 256) {
    if (threadIdx.x == 0) s_nlong = 0;
    __syncthreads();
    const int s = base + (int)threadIdx.x;
    if (s < ns) {
      int* l = V.nc_list + V.nc_off[s];
      const int k = V.nc_cnt[s];
      if (k > NC_SHORT) {
        s_long[atomicAdd(&s_nlong, 1)] = s;
      } else {
        for (int a = 1; a < k; ++a) {  // insertion sort: a handful of entries
          const int v = l[a];
          int b = a - 1;
          while (b >= 0 && l[b] > v) {
            l[b + 1] = l[b];
            --b;
          }
          l[b + 1] = v;
        }
        int u = 0;
        for (int a = 0; a < k; ++a)
          if (a == 0 || l[a] != l[a - 1]) l[u++] = l[a];
        V.nc_fill[s] = u;
      }
    }
    __syncthreads();
    const int nlong = s_nlong;
    for (int q = 0; q < nlong; ++q) {
      const int sl = s_long[q], off = V.nc_off[sl];
      const int u = nc_unique_long(V.nc_list + off, V.cross_i + off, V.nc_cnt[sl], nt, s_bits, s_wt);
      if (threadIdx.x == 0) V.nc_fill[sl] = u;
    }
    __syncthreads();
  }
}
template <bool EXT>
__global__ __launch_bounds__(256) void k_nc_emit(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  __shared__ int s_long[256], s_nlong;
  const int ns = V.ns;
  for (int base = blockIdx.x * 256; base < ns; base += gridDim.x * 256) {
    if (threadIdx.x == 0) s_nlong = 0;
    __syncthreads();
    const int s = base + (int)threadIdx.x;
    if (s < ns) {
      const int u = V.nc_fill[s], o = V.scan[s];
      if (u > NC_SHORT) {
        s_long[atomicAdd(&s_nlong, 1)] = s;  // (the workgroup writes a long list together)
      } else {
        const int* l = V.nc_list + V.nc_off[s];
        for (int a = 0; a < u; ++a) {
          V.corr[2 * (o + a)] = s;
          V.corr[2 * (o + a) + 1] = l[a];
        }
      }
    }
    __syncthreads();
    const int nlong = s_nlong;
    for (int q = 0; q < nlong; ++q) {
      const int sl = s_long[q], u = V.nc_fill[sl], o = V.scan[sl];
      const int* l = V.nc_list + V.nc_off[sl];
      for (int a = threadIdx.x; a < u; a += 256) {
        V.corr[2 * (o + a)] = sl;
        V.corr[2 * (o + a) + 1] = l[a];
      }
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) V.mcounts[MC_NCORR] = V.scan[ns];
  if (V.mail && blockIdx.x == 0 && threadIdx.x < 48) match_mail(V, threadIdx.x, V.scan[ns], V.mcounts[MC_NTUPLE]);
}

// ---- fused tails for clouds of up to 32768 points: one workgroup of 1024 threads per pair, every thread owning one
// contiguous run of at most KMAX (16 or 32) indices, so flag -> exclusive scan -> compaction happens in registers and LDS
// without the three-launch (flags, scan, compact) round trips.
// K6: unpack both NN tables, mutual-NN test, cross pairs in ascending i.
template <bool EXT, int KMAX>
__global__ __launch_bounds__(1024) void k_cross_fused(ViewExt<MatchView> x, MatchView one, int lds_gather) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  extern __shared__ int fl_s[];  // [n_large] nn index | keep flag << 31, staged with coalesced (striped) accesses
                                 // (+ [n_small] with lds_gather: the smaller cloud's table, see below)
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_large = V.n_large, n_small = V.n_small;
  QTR_STAMP(STAMP_CROSS, 0)
  // (one workgroup walks both clouds: sixteen rows per thread and round trip, or the loop is a chain of ~2 x n / 1024
  // dependent memory latencies)
  // The mutual test looks the smaller cloud's answer up at a random row for every row of the larger one: ~18 k gathers
  // of 8 bytes issued by ONE compute unit cost 64 address cycles per wave instruction (8 us).  When both tables fit, the
  // smaller cloud's is therefore staged in LDS with coalesced loads and the gather runs there.
  int* nn_s = lds_gather ? fl_s + n_large : nullptr;
  for (int j0 = tid; j0 < n_small; j0 += 16 * 1024) {
    u64 b[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) b[q] = (j0 + q * 1024 < n_small) ? V.best_small[j0 + q * 1024] : ~0ULL;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (j0 + q * 1024 < n_small) {
        const int v = (b[q] == ~0ULL) ? 0 : (int)(u32)b[q];
        V.nn_of_small[j0 + q * 1024] = v;
        if (nn_s) nn_s[j0 + q * 1024] = v;
      }
  }
  if (nn_s) __syncthreads();  // (uniform)
  for (int i0 = tid; i0 < n_large; i0 += 16 * 1024) {
    u64 b[16];
    int back[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) b[q] = (i0 + q * 1024 < n_large) ? V.best_large[i0 + q * 1024] : ~0ULL;
    if (nn_s) {
#pragma unroll
      for (int q = 0; q < 16; ++q) back[q] = nn_s[(b[q] == ~0ULL) ? 0 : (int)(u32)b[q]];
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const u64 bs = V.best_small[(b[q] == ~0ULL) ? 0 : (int)(u32)b[q]];
        back[q] = (bs == ~0ULL) ? 0 : (int)(u32)bs;
      }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = i0 + q * 1024;
      if (i < n_large) {
        const int j = (b[q] == ~0ULL) ? 0 : (int)(u32)b[q];
        V.nn_of_large[i] = (b[q] == ~0ULL) ? -1 : j;
        fl_s[i] = j | ((b[q] != ~0ULL && back[q] == i) ? (int)0x80000000 : 0);
      }
    }
  }
  __syncthreads();
  QTR_STAMP(STAMP_CROSS, 1)
  const int K = (n_large + 1023) >> 10, base = tid * K;
  int jj[KMAX];
  u32 keep = 0;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
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
  for (int k = 0; k < KMAX; ++k)
    if ((keep >> k) & 1u) {
      V.cross_i[run] = base + k;
      V.cross_j[run] = jj[k];
      ++run;
    }
  if (tid == 0) V.mcounts[MC_NCROSS] = total;
  QTR_STAMP(STAMP_CROSS, 2)
}

// K8 + gather: passed cross pairs -> tgt_of_src, compaction in source order, the matched keypoint clouds
// (when asked for) and the counters for the host.
template <bool EXT, int KMAX>
__global__ __launch_bounds__(1024) void k_pairs_fused(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  __shared__ int wsum[16];
  __shared__ int s_ntuple;
  extern __shared__ int tg_s[];  // [ns]: target of every source index, or -1 (the map lives in LDS only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swapped = V.swapped, ns = V.ns;
  if (tid == 0) s_ntuple = 0;
  for (int i = tid; i < ns; i += 1024) tg_s[i] = -1;
  __syncthreads();
  const int nc = V.mcounts[MC_NCROSS];
  int local = 0;
  for (int c0 = tid; c0 < nc; c0 += 8 * 1024) {  // (eight per thread and round trip, as in k_cross_fused)
    int pf[8], ci[8], cj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = min(c0 + q * 1024, nc - 1);
      pf[q] = (c0 + q * 1024 < nc) ? (int)V.passed[c] : 0;
      ci[q] = V.cross_i[c];
      cj[q] = V.cross_j[c];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (pf[q]) {
        tg_s[swapped ? cj[q] : ci[q]] = swapped ? ci[q] : cj[q];  // (a source index occurs at most once after the cross-check)
        ++local;
      }
  }
  local = wave_sum_i32(local);
  if (lane == 0 && local) atomicAdd(&s_ntuple, local);
  __syncthreads();
  const int K = (ns + 1023) >> 10, base = tid * K;
  int tt[KMAX];
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
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
  // the counters go to the host as soon as they are known: it enqueues the solver's launches (stream-ordered behind
  // this kernel) while the list below is still being written
  if (V.mail && tid < 48) match_mail(V, tid, total, s_ntuple);
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (tt[k] >= 0) {
      V.corr[2 * run] = base + k;
      V.corr[2 * run + 1] = tt[k];
      if (V.m_src && run < V.m_cap) {  // the count is still reported: the host raises QTR_ERR_CAPACITY past m_cap
        float4 pa = V.vox_s[base + k], pb = V.vox_t[tt[k]];
        pa.w = 0.f;
        pb.w = 0.f;
        V.m_src[run] = pa;
        V.m_tgt[run] = pb;
      }
      ++run;
    }
  if (tid == 0) {
    V.mcounts[MC_NCORR] = total;
    V.mcounts[MC_NTUPLE] = s_ntuple;
  }
}

// K6 on many compute units (see TAIL_MAXWG above): a workgroup takes CM_ROWS consecutive rows of the larger cloud — their
// answers, the answers' answers (a gather of 8 bytes per row, spread over all workgroups instead of funnelled through one
// unit's address pipeline), the mutual test — and a slice of the smaller cloud's table; the cross pairs land in
// ascending i behind the pairs of the workgroups before it.
#define CM_ROWS 512
template <bool EXT>
__global__ __launch_bounds__(256) void k_cross_multi(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int n_large = V.n_large, n_small = V.n_small;
  const int nwg = (n_large + CM_ROWS - 1) / CM_ROWS, w = blockIdx.x;
  if (w >= nwg) return;
  __shared__ int s_w[4], s_red[5];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const int per = (n_small + nwg - 1) / nwg, j0 = w * per, j1 = min(n_small, j0 + per);
    for (int j = j0 + tid; j < j1; j += 256) {
      const u64 bsm = V.best_small[j];
      V.nn_of_small[j] = (bsm == ~0ULL) ? 0 : (int)(u32)bsm;
    }
  }
  const int base = w * CM_ROWS + 2 * tid;  // two consecutive rows per thread: thread order = row order
  u64 b[2], bs[2];
  int jj[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) b[k] = (base + k < n_large) ? V.best_large[base + k] : ~0ULL;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    jj[k] = (b[k] == ~0ULL) ? 0 : (int)(u32)b[k];
    bs[k] = V.best_small[jj[k]];
  }
  bool keep[2];
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = base + k;
    const int back = (bs[k] == ~0ULL) ? 0 : (int)(u32)bs[k];
    keep[k] = i < n_large && b[k] != ~0ULL && back == i;
    if (i < n_large) V.nn_of_large[i] = (b[k] == ~0ULL) ? -1 : jj[k];
    cnt += keep[k] ? 1 : 0;
  }
  int tot;
  const int ex = wave_excl_scan_i32(cnt, &tot);
  if (lane == 0) s_w[wave] = tot;
  __syncthreads();
  int run = ex, total = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    run += (q < wave) ? s_w[q] : 0;
    total += s_w[q];
  }
  const int before = tail_lookback(V.scan, w, total, s_red);
  if (before < 0) {  // look-back timed out: no writes at unknown offsets; the next kernel reports the failure
    if (tid == 0) {
      V.mcounts[MC_TAILERR] = 1;
      if (w == nwg - 1) V.mcounts[MC_NCROSS] = 0;
    }
    return;
  }
  run += before;
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (keep[k]) {
      V.cross_i[run] = base + k;
      V.cross_j[run] = jj[k];
      ++run;
    }
  if (w == nwg - 1 && tid == 0) V.mcounts[MC_NCROSS] = before + total;
}

// K8 on many compute units: a workgroup owns PM_SRC consecutive SOURCE indices; it walks the whole list of cross pairs
// (a few thousand entries) for the tuple-test survivors whose source falls into its range, and writes them — in source
// order — behind the correspondences of the workgroups before it.  The last workgroup knows the totals and tells the host.
#define PM_SRC 512
template <bool EXT>
__global__ __launch_bounds__(256) void k_pairs_multi(ViewExt<MatchView> x, MatchView one) {
  const MatchView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int ns = V.ns, swapped = V.swapped;
  const int nwg = max(1, (ns + PM_SRC - 1) / PM_SRC), w = blockIdx.x;
  if (w >= nwg) return;
  __shared__ int s_tg[PM_SRC];
  __shared__ int s_w[4], s_red[5], s_ntuple;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s0 = w * PM_SRC;
  s_tg[2 * tid] = -1;
  s_tg[2 * tid + 1] = -1;
  if (tid == 0) s_ntuple = 0;
  __syncthreads();
  const int nc = V.mcounts[MC_NCROSS];
  int local = 0;
  for (int c0 = tid; c0 < nc; c0 += 8 * 256) {  // (eight per thread and round trip)
    int pf[8], ci[8], cj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = min(c0 + q * 256, nc - 1);
      pf[q] = (c0 + q * 256 < nc) ? (int)V.passed[c] : 0;
      ci[q] = V.cross_i[c];
      cj[q] = V.cross_j[c];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (pf[q]) {
        const int src = (swapped ? cj[q] : ci[q]) - s0;  // (a source index occurs at most once after the cross-check)
        if (src >= 0 && src < PM_SRC) s_tg[src] = swapped ? ci[q] : cj[q];
        ++local;
      }
  }
  local = wave_sum_i32(local);
  if (lane == 0 && local) atomicAdd(&s_ntuple, local);
  __syncthreads();
  int tt[2], cnt = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    tt[k] = (s0 + 2 * tid + k < ns) ? s_tg[2 * tid + k] : -1;
    cnt += tt[k] >= 0 ? 1 : 0;
  }
  int tot;
  const int ex = wave_excl_scan_i32(cnt, &tot);
  if (lane == 0) s_w[wave] = tot;
  __syncthreads();
  int run = ex, total = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    run += (q < wave) ? s_w[q] : 0;
    total += s_w[q];
  }
  const int before = tail_lookback(V.scan + TAIL_MAXWG, w, total, s_red);
  const bool tail_err = before < 0 || V.mcounts[MC_TAILERR] != 0;  // (this look-back or k_cross_multi's timed out)
  run += before;
  // the counters go to the host as soon as they are known: it enqueues the solver's launches (stream-ordered behind this
  // kernel) while the lists are still being written.  A failed look-back reports -1 correspondences: QTR_ERR_HIP.
  if (w == nwg - 1) {
    if (V.mail && tid < 48) match_mail(V, tid, tail_err ? -1 : before + total, s_ntuple);
    if (tid == 0) {
      V.mcounts[MC_NCORR] = tail_err ? -1 : before + total;
      V.mcounts[MC_NTUPLE] = s_ntuple;
    }
  }
  if (tail_err) return;
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (tt[k] >= 0) {
      const int src = s0 + 2 * tid + k;
      V.corr[2 * run] = src;
      V.corr[2 * run + 1] = tt[k];
      if (V.m_src && run < V.m_cap) {  // the count is still reported: the host raises QTR_ERR_CAPACITY past m_cap
        float4 pa = V.vox_s[src], pb = V.vox_t[tt[k]];
        pa.w = 0.f;
        pb.w = 0.f;
        V.m_src[run] = pa;
        V.m_tgt[run] = pb;
      }
      ++run;
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

// Which tail follows the two searches when the mutual test is on: the fused one (k_cross_multi, k_tuple, k_pairs_multi —
// three launches; their look-backs have TAIL_MAXWG words, one per 512 rows) or the six-launch chain with its
// single-workgroup scans (54 us each at 50 000 rows).  Until the end of round 5 the fused tail stopped at 32768 rows, the
// limit of the one-workgroup kernels it had replaced (QTR_MATCH_TAIL=single, test build: still theirs).
static_assert(CM_ROWS == 512 && PM_SRC == 512, "TAIL_MAXWG look-back words cover TAIL_MAXWG * 512 rows");
static bool tail_is_fused(bool crosscheck, int max_large, int max_ns) {
  static const bool tail_single = [] {
    const char* e = QTR_ENGINE_ENV("QTR_MATCH_TAIL");
    return e && strcmp(e, "single") == 0;
  }();
  const int lim = tail_single ? 32768 : TAIL_MAXWG * 512;
  return crosscheck && max_large <= lim && max_ns <= lim;
}
static inline int grid_for(int n) {
  int g = (n + 255) / 256;
  return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}

static MatchView make_match_view(FrontBufs& F, int ns, int nt, const qtr_frontend_params& fp, unsigned long long seed) {
  MatchView V;
  memset(&V, 0, sizeof(V));
  // fi = larger cloud, fj = smaller (reference feature_matcher.cc:84-92)
  const int swapped = nt > ns ? 1 : 0;
  CloudBufs& Ci = F.cloud[swapped ? 1 : 0];
  CloudBufs& Cj = F.cloud[swapped ? 0 : 1];
  V.swapped = swapped;
  V.ns = ns;
  V.nt = nt;
  V.n_large = swapped ? nt : ns;
  V.n_small = swapped ? ns : nt;
  V.pad_large = (V.n_large + NN_QPB - 1) / NN_QPB * NN_QPB;
  V.pad_small = (V.n_small + NN_QPB - 1) / NN_QPB * NN_QPB;
  V.vox_i = Ci.vox;
  V.vox_j = Cj.vox;
  V.mean_i = Ci.mean;
  V.mean_j = Cj.mean;
  V.fpfh_i = Ci.fpfh;
  V.fpfh_j = Cj.fpfh;
  V.baseT_i = Ci.baseT;
  V.queryT_i = Ci.queryT;
  V.norms_i = Ci.norms;
  V.baseT_j = Cj.baseT;
  V.queryT_j = Cj.queryT;
  V.norms_j = Cj.norms;
  V.baseTb_i = Ci.baseTb;
  V.baseTb_j = Cj.baseTb;
  V.baseH_i = Ci.baseH;
  V.baseH_j = Cj.baseH;
  V.queryH_i = Ci.queryH;
  V.queryH_j = Cj.queryH;
  V.queryH_c = F.queryH_c;
  V.nb_row_i = Ci.nb_row;
  V.nb_row_j = Cj.nb_row;
  V.nb_start_i = Ci.nb_start;
  V.nb_start_j = Cj.nb_start;
  V.hash_i = Ci.dd_hash;
  V.hash_j = Cj.dd_hash;
  V.table_i = Ci.dd_table;
  V.table_j = Cj.dd_table;
  V.dd_mask = F.dd_slots - 1;
  V.best_small = F.best_small;
  V.best_large = F.best_large;
  V.nn_of_small = F.nn_of_small;
  V.nn_of_large = F.nn_of_large;
  V.cross_i = F.cross_i;
  V.cross_j = F.cross_j;
  V.flags = F.flags;
  V.scan = F.scan;
  V.passed = F.passed;
  V.tgt_of_src = F.tgt_of_src;
  V.corr = F.corr;
  V.mcounts = F.mcounts;
  V.partial = (NnPartial*)F.nn_partial;
  V.recheck_rows = F.recheck_rows;
  V.recheck_thr = F.recheck_thr;
  V.recheck_span = F.recheck_span;
  V.recheck_q = F.recheck_q;
  V.hit_rows = F.hit_rows;
  V.queryT_c = F.queryT_c;
  V.norms_c = F.norms_c;
  V.vox_s = F.cloud[0].vox;
  V.vox_t = F.cloud[1].vox;
  V.m_src = F.m_src;
  V.m_tgt = F.m_tgt;
  V.m_cap = F.m_cap;
  V.mail = F.mail;
  V.counts0 = F.cloud[0].counts;
  V.counts1 = F.cloud[1].counts;
  V.seq = F.mail_seq;
  V.nc_cnt = F.nc_cnt;
  V.nc_fill = F.nc_fill;
  V.nc_off = F.nc_off;
  V.nc_list = F.nc_list;
  V.crosscheck = fp.use_crosscheck ? 1 : 0;
  V.tuple = (fp.use_tuple_test && fp.tuple_scale != 0) ? 1 : 0;
  V.tuple_scale = fp.tuple_scale;
  V.seed = seed;
  NnDir& d0 = V.d[0];  // rows of the smaller cloud ask the larger one
  d0.baseT = Ci.baseT;
  d0.bnorm = Ci.norms;
  d0.nb = V.n_large;
  d0.nb_pad = V.pad_large;
  d0.queryT = Cj.queryT;
  d0.baseH = Ci.baseH;
  d0.queryH = Cj.queryH;
  d0.qnorm = Cj.norms;
  d0.qmap = nullptr;
  d0.nq_pad = V.pad_small;
  d0.A = Cj.fpfh;
  d0.QT = Cj.queryT;
  d0.qt_pad = V.pad_small;
  d0.baseTb = Ci.baseTb;
  d0.brow = Ci.nb_row;
  d0.bstart = Ci.nb_start;
  d0.qorder = Cj.nb_row;
  d0.best = F.best_small;
  d0.nq_slot = MC_NQ0;
  d0.rc_slot = MC_RECHECK0;
  NnDir& d1 = V.d[1];  // hit rows of the larger cloud ask the smaller one
  d1.baseT = Cj.baseT;
  d1.bnorm = Cj.norms;
  d1.nb = V.n_small;
  d1.nb_pad = V.pad_small;
  d1.queryT = F.queryT_c;
  d1.baseH = Cj.baseH;
  // (the f16 engine reads the hit rows' fragments and norms in place, by row; the f32 MFMA engine a gathered copy)
  d1.queryH = (F.nn_engine == 2) ? Ci.queryH : F.queryH_c;
  d1.qnorm = (F.nn_engine == 2) ? Ci.norms : F.norms_c;
  d1.qmap = F.hit_rows;
  d1.nq_pad = V.pad_large;
  d1.A = Ci.fpfh;
  d1.QT = Ci.queryT;
  d1.qt_pad = V.pad_large;
  d1.baseTb = Cj.baseTb;
  d1.brow = Cj.nb_row;
  d1.bstart = Cj.nb_start;
  d1.qorder = nullptr;
  d1.best = F.best_large;
  d1.nq_slot = MC_NHIT;
  d1.rc_slot = MC_RECHECK1;
  return V;
}

// Enqueues the matcher for the pairs of `views` (G of them; one travels in the kernel arguments).  ev: optional
// brackets of the two k_nn_mfma launches (single pair only).
static hipError_t match_launch(const MatchView* views, int G, int nn_engine, int n_cu, ViewStage* stage,
                               hipStream_t st, hipEvent_t const* ev, bool init_done = false, bool prep_done = false) {
  MatchArgs a;
  a.one = views[0];
  a.ext = nullptr;
  if (G > 1) {
    a.ext = (const MatchView*)stage_push(stage, views, sizeof(MatchView) * (size_t)G, st);
    if (!a.ext) return hipErrorOutOfMemory;
  }
  int max_large = 1, max_small = 1, max_ns = 1, max_pad = NN_QPB;
  bool any_tuple = false;
  for (int g = 0; g < G; ++g) {
    max_large = max(max_large, views[g].n_large);
    max_small = max(max_small, views[g].n_small);
    max_ns = max(max_ns, views[g].ns);
    max_pad = max(max_pad, views[g].pad_large);
    any_tuple = any_tuple || views[g].tuple;
  }
  const dim3 B256(256);
  if (!init_done)
    LAUNCH_MV(k_match_init, a, dim3(grid_for(max(max_large, views[0].dd_mask + 1)), 1, G), B256, 0, st, prep_done ? 0 : 1, (int*)nullptr, 0);
  // K5: NN of every small-cloud descriptor in the large cloud, then of the HIT rows of the large cloud in the small
  // one (the reference asks the latter lazily, feature_matcher.cc:113-122; the mutual test only reads hit rows)
#ifdef QTR_TEST_ENGINES
  if (nn_engine == 0) {
    auto nsplit = [](int nq, int nb) {
      int blocks_x = (nq + 255) / 256;
      int s = (1024 + blocks_x - 1) / blocks_x;
      int maxs = (nb + NN_TILE - 1) / NN_TILE;
      if (s > maxs) s = maxs;
      if (s < 1) s = 1;
      if (s > 256) s = 256;
      return s;
    };
    LAUNCH_MV(k_desc_prep, a, dim3(max_pad / 256, 2, G), B256, 0, st, 1);  // norms for the bin order k_hit_compact follows
    LAUNCH_MV(k_norm_bins, a, dim3(1, 2, G), dim3(1024), 0, st);
    if (ev && ev[0]) (void)hipEventRecord(ev[0], st);
    LAUNCH_MV(k_nn_exact, a, dim3((max_small + 255) / 256, nsplit(max_small, max_large), G), B256, 0, st, 0);
    if (ev && ev[1]) (void)hipEventRecord(ev[1], st);
    LAUNCH_MV(k_hit_compact, a, dim3(1, 1, G), dim3(1024), (size_t)((max_large + 31) / 32) * 4, st, 1, 0);
    if (ev && ev[2]) (void)hipEventRecord(ev[2], st);
    LAUNCH_MV(k_nn_exact, a, dim3((max_large + 255) / 256, nsplit(max_large, max_small), G), B256, 0, st, 1);
    if (ev && ev[3]) (void)hipEventRecord(ev[3], st);
  } else
#endif
  {
#ifdef QTR_TEST_ENGINES
    const bool f16 = nn_engine == 2;
#else
    (void)nn_engine;
    constexpr bool f16 = true;
#endif
    // (the f16 engine needs nothing of k_desc_prep but the norms, hashes and duplicate table: k2_fpfh left them)
    if (!(prep_done && f16)) LAUNCH_MV(k_desc_prep, a, dim3(max_pad / 256, 2, G), B256, 0, st, f16 ? 0 : 2);
    if (f16) {  // operand tables with the duplicates hidden, straight from the descriptors
      if (a.ext) LAUNCH_MV(k_half_tables, a, dim3(max_pad / 256, 2, G), B256, 0, st);
      else LAUNCH_MV(k_half_tables, a, dim3(max_pad / HT_ROWS_SINGLE, 2, G), dim3(HT_ROWS_SINGLE), 0, st);
    }
#ifdef QTR_TEST_ENGINES
    else {  // the norm-bin order serves the span re-check (k_nn_exact_rows); the f16 engine's filter sweeps every tile
      LAUNCH_MV(k_desc_dedup, a, dim3((max_large + 255) / 256, 2, G), B256, 0, st);
      LAUNCH_MV(k_norm_bins, a, dim3(1, 2, G), dim3(1024), 0, st);
      LAUNCH_MV(k_norm_gather, a, dim3((max_large + 255) / 256, 2, G), B256, 0, st);
    }
#endif
    // persistent workgroups: one per compute unit (two fit; the other lane's launch may be the second).
    // QTR_NN_WGS_PER_CU = 2 (experiment knob) launches both from this chain.
    static const int wgs_per_cu = [] {
      const char* e = QTR_ENGINE_ENV("QTR_NN_WGS_PER_CU");
      return (e && atoi(e) == 2) ? 2 : 1;
    }();
    const int X = n_cu * wgs_per_cu;
    // how a single pair's (query block, slice) grid is cut into the eight XCDs' rectangles: 2^la x 2^(3-la)
    static const int nn_deal = [] {
      const char* e = QTR_ENGINE_ENV("QTR_NN_DEAL");
      return (e && atoi(e) >= 0 && atoi(e) <= 3) ? atoi(e) : 2;  // (4 x 2: same-box sweep, gpurun_out/r5e/deal.txt)
    }();
    // QTR_NN_EVENTS=record: bracket the launch with two hipEventRecord calls instead of attaching the events to it
    static const bool attach_events = [] {
      const char* e = QTR_ENGINE_ENV("QTR_NN_EVENTS");
      return !(e && strcmp(e, "record") == 0);
    }();
    auto run_dir = [&](int dir, int nq_max, int nb_max, hipEvent_t e0, hipEvent_t e1) {
      // a single pair's plan: direction 0 here, direction 1 by k_hit_compact (see nn_plan_single)
      const int plan = (G == 1 && dir == 0) ? nn_plan_single((views[0].n_small + NN_QPB - 1) / NN_QPB, views[0].d[0].nb_pad / 32, X) : -1;
      if (e0 && e1 && attach_events) {
        if (f16) LAUNCH_MV_EV(k_nn_f16, a, dim3(X, 1, 1), B256, 0, st, e0, e1, dir, G, nn_deal, plan);
#ifdef QTR_TEST_ENGINES
        else LAUNCH_MV_EV(k_nn_mfma, a, dim3(X, 1, 1), B256, 0, st, e0, e1, dir, G);
#endif
      } else {
        if (e0) (void)hipEventRecord(e0, st);
        if (f16) LAUNCH_MV(k_nn_f16, a, dim3(X, 1, 1), B256, 0, st, dir, G, nn_deal, plan);
#ifdef QTR_TEST_ENGINES
        else LAUNCH_MV(k_nn_mfma, a, dim3(X, 1, 1), B256, 0, st, dir, G);
#endif
        if (e1) (void)hipEventRecord(e1, st);
      }
      if (f16) LAUNCH_MV(k_nn_finish_f16, a, dim3((nq_max + NN_FINH_Q - 1) / NN_FINH_Q, 1, G), B256, 0, st, dir, 800.0f);
#ifdef QTR_TEST_ENGINES
      else LAUNCH_MV(k_nn_finish, a, dim3((nq_max + NN_FIN_THREADS - 1) / NN_FIN_THREADS, 1, G), dim3(NN_FIN_THREADS), 0, st, dir,
                     X, G, 0.0f, 0);
#endif
      // (single pair: 8 x 64 workgroups = the 512 the device holds at two per compute unit — one round)
      if (f16) LAUNCH_MV(k_recheck_filter, a, dim3(8, G > 1 ? 16 : 64, G), B256, 0, st, dir);
      // (row group, span slice) workgroups: a group's span is a few per cent of the base cloud
      (void)nb_max;
      // (most spans are a fraction of a per cent of the cloud, a few cover half of it: enough slices that the widest
      // span is shared by many workgroups)
      const int ey = G > 1 ? 4 : 16;
      int ex = 128;
      if (G > 1) ex = max(8, 384 / G);  // a group of pairs shares the device
#ifdef QTR_TEST_ENGINES
      if (!f16) LAUNCH_MV(k_nn_exact_rows, a, dim3(ex, ey, G), B256, 0, st, dir);  // (the f16 engine's filter is complete)
#else
      (void)ex;
      (void)ey;
#endif
    };
    run_dir(0, max_small, max_large, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr);
    LAUNCH_MV(k_hit_compact, a, dim3(1, 1, G), dim3(1024), (size_t)((max_large + 31) / 32) * 4, st, f16 ? 0 : 1,
              (f16 && G == 1) ? X : 0);
#ifdef QTR_TEST_ENGINES
    if (!f16) LAUNCH_MV(k_hit_gather, a, dim3(max_pad / 256, 1, G), B256, 0, st, 1);  // (f16: the rows are read in place)
#endif
    run_dir(1, max_large, max_small, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr);
  }
  // K6 cross-check -> pairs in ascending i
  const bool crosscheck = views[0].crosscheck != 0;
  const bool fused_tail = tail_is_fused(crosscheck, max_large, max_ns);
  const bool fused16 = max_large <= 16384 && max_ns <= 16384;
  // the multi-workgroup compactions (QTR_MATCH_TAIL=single: the one-workgroup kernels of round 2, kept for comparison)
  static const bool tail_single = [] {
    const char* e = QTR_ENGINE_ENV("QTR_MATCH_TAIL");
    return e && strcmp(e, "single") == 0;
  }();
  const bool tail_multi = fused_tail && !tail_single;
  if (!crosscheck) {  // the unfiltered list, see k_nc_list
    LAUNCH_MV(k_cross_flags2, a, dim3(grid_for(max_large), 1, G), B256, 0, st);
    LAUNCH_MV(k_scan_flags, a, dim3(1, 1, G), dim3(1024), 0, st);
    LAUNCH_MV(k_nc_list, a, dim3(grid_for(max_large + max_small), 1, G), B256, 0, st);
  } else if (fused_tail) {
    // (both nearest-neighbour tables in LDS when they fit: see the kernel)
    const int lds_gather = ((size_t)max_large + (size_t)max_small) * 4 <= (size_t)CROSS_LDS_BYTES ? 1 : 0;
    const size_t cross_lds = (size_t)max_large * 4 + (lds_gather ? (size_t)max_small * 4 : 0);
    if (tail_multi) LAUNCH_MV(k_cross_multi, a, dim3((max_large + CM_ROWS - 1) / CM_ROWS, 1, G), B256, 0, st);
#ifdef QTR_TEST_ENGINES
    else if (fused16) LAUNCH_MV_K(k_cross_fused, 16, a, dim3(1, 1, G), dim3(1024), cross_lds, st, lds_gather);
    else LAUNCH_MV_K(k_cross_fused, 32, a, dim3(1, 1, G), dim3(1024), cross_lds, st, lds_gather);
#else
    (void)cross_lds;
    (void)fused16;
#endif
  } else {
    LAUNCH_MV(k_cross_flags2, a, dim3(grid_for(max_large), 1, G), B256, 0, st);
    LAUNCH_MV(k_scan_flags, a, dim3(1, 1, G), dim3(1024), 0, st);
    LAUNCH_MV(k_cross_compact, a, dim3(grid_for(max_large), 1, G), B256, 0, st);
  }
  // K7 tuple test
  if (any_tuple) LAUNCH_MV(k_tuple, a, dim3(G > 1 ? 256 : 2048, 1, G), B256, 0, st);
  // K8 un-swap, sort by (src, tgt), unique  ==  compaction in source-index order
  if (!crosscheck) {
    const int gl = grid_for(max_large + max_small), gs = grid_for(max_ns);
    LAUNCH_MV(k_nc_scatter, a, dim3(gl, 1, G), B256, 0, st, 0);
    LAUNCH_MV(k_nc_scan, a, dim3(1, 1, G), dim3(1024), 0, st, 0);
    LAUNCH_MV(k_nc_scatter, a, dim3(gl, 1, G), B256, 0, st, 1);
    LAUNCH_MV(k_nc_unique, a, dim3(gs, 1, G), B256, 0, st);
    LAUNCH_MV(k_nc_scan, a, dim3(1, 1, G), dim3(1024), 0, st, 1);
    LAUNCH_MV(k_nc_emit, a, dim3(gs, 1, G), B256, 0, st);
  } else if (fused_tail) {
    if (tail_multi) LAUNCH_MV(k_pairs_multi, a, dim3(max(1, (max_ns + PM_SRC - 1) / PM_SRC), 1, G), B256, 0, st);
#ifdef QTR_TEST_ENGINES
    else if (fused16) LAUNCH_MV_K(k_pairs_fused, 16, a, dim3(1, 1, G), dim3(1024), (size_t)max_ns * 4, st);
    else LAUNCH_MV_K(k_pairs_fused, 32, a, dim3(1, 1, G), dim3(1024), (size_t)max_ns * 4, st);
#endif
  } else {
    LAUNCH_MV(k_scatter_pairs, a, dim3(grid_for(max_small), 1, G), B256, 0, st);
    LAUNCH_MV(k_scan_nonneg, a, dim3(1, 1, G), dim3(1024), 0, st);
    LAUNCH_MV(k_corr_compact2, a, dim3(grid_for(max_ns), 1, G), B256, 0, st);
  }
  return hipGetLastError();
}

hipError_t match_init_enqueue(FrontBufs& F, int ns, int nt, const qtr_frontend_params& fp, hipStream_t st, bool clear_tables,
                              int* zero_words, int n_zero) {
  (void)hipGetLastError();
  MatchArgs a;
  a.one = make_match_view(F, ns, nt, fp, (unsigned long long)fp.seed);
  a.ext = nullptr;
  LAUNCH_MV(k_match_init, a, dim3(grid_for(max(a.one.n_large, a.one.dd_mask + 1)), 1, 1), dim3(256), 0, st, clear_tables ? 1 : 0, zero_words, n_zero);
  return hipGetLastError();
}
hipError_t match_enqueue(FrontBufs& F, int ns, int nt, const qtr_frontend_params& fp, hipStream_t st, bool init_done,
                         bool prep_done) {
  (void)hipGetLastError();
  const MatchView V = make_match_view(F, ns, nt, fp, (unsigned long long)fp.seed);
  const bool fused_tail = tail_is_fused(V.crosscheck != 0, V.n_large, ns);
  F.gathered = fused_tail && F.m_src != nullptr;
  const bool evs = F.nn_events != 0;
  return match_launch(&V, 1, F.nn_engine, F.n_cu, nullptr, st, evs ? F.ev_nn : nullptr, init_done, prep_done);
}

hipError_t match_enqueue_group(FrontBufs* const* F, int G, const int* n, const qtr_frontend_params* fp,
                               const unsigned long long* seeds, ViewStage* stage, hipStream_t st, bool prep_done) {
  (void)hipGetLastError();
  std::vector<MatchView> v((size_t)G);
  int max_large = 1, max_ns = 1;
  for (int g = 0; g < G; ++g) {
    v[g] = make_match_view(*F[g], n[2 * g], n[2 * g + 1], *fp, seeds[g]);
    max_large = max(max_large, v[g].n_large);
    max_ns = max(max_ns, v[g].ns);
  }
  const bool fused_tail = tail_is_fused(fp->use_crosscheck != 0, max_large, max_ns);  // (as match_launch decides)
  for (int g = 0; g < G; ++g) F[g]->gathered = fused_tail && F[g]->m_src != nullptr;
  return match_launch(v.data(), G, F[0]->nn_engine, F[0]->n_cu, stage, st, nullptr, false, prep_done);
}

hipError_t gather_matched_enqueue(FrontBufs& F, int L, float4* m_src, float4* m_tgt, hipStream_t st) {
  if (L <= 0 || (F.gathered && m_src == F.m_src && m_tgt == F.m_tgt)) return hipSuccess;  // k_pairs_fused did it
  hipLaunchKernelGGL(k_gather_matched, dim3(grid_for(L)), dim3(256), 0, st, F.cloud[0].vox, F.cloud[1].vox, F.corr, L,
                     m_src, m_tgt);
  return hipGetLastError();
}

// the 32-per-thread fused tails stage up to 32768 ints (128 KB) in dynamic LDS; k_cross_fused (either variant) up to
// CROSS_LDS_BYTES when both nearest-neighbour tables fit
hipError_t match_init_attributes() {
  hipError_t e;
  (void)e;
#ifdef QTR_TEST_ENGINES  // (the one-workgroup tails: comparison engines)
  if ((e = hipFuncSetAttribute((const void*)k_cross_fused<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, CROSS_LDS_BYTES)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)k_cross_fused<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, CROSS_LDS_BYTES)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)k_cross_fused<false, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, CROSS_LDS_BYTES)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)k_cross_fused<true, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, CROSS_LDS_BYTES)) != hipSuccess) return e;
#define SET_LDS2(kern)                                                                                                   \
  if ((e = hipFuncSetAttribute((const void*)kern<false, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024)) != hipSuccess) \
    return e;                                                                                                            \
  if ((e = hipFuncSetAttribute((const void*)kern<true, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024)) != hipSuccess)  \
    return e;
  SET_LDS2(k_pairs_fused)
#undef SET_LDS2
#endif
  return hipSuccess;
}

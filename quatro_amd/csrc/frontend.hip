// frontend.hip — voxel-grid down-sampling, radius-neighbour lists, normals, SPFH and FPFH on gfx950.
//
// Replaces, at the reference's call sites, the un-vendored PCL 1.8.1 classes the reference drives:
//   voxelize<T>()                      reference include/quatro.hpp:49-68      (pcl::VoxelGrid)
//   FPFHEstimation::computeFPFHFeatures reference src/teaser_utils/fpfh.cc:44-75 (pcl::NormalEstimation,
//                                       pcl::FPFHEstimationOMP, pcl::search::KdTree radius search)
// All of this is integer/gather work bounded by HBM/L2 traffic, not matrix work: the kernels are
// organised around coalesced 16-byte point loads, LDS staging and wave-level (ballot / shuffle)
// compaction.  kd-trees are replaced by a sort-based uniform grid (cell = search radius): points are
// sorted by packed cell key and a query scans 9 contiguous key ranges.
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "common.h"
#include "frontend.h"

// order-preserving float <-> u32 encoding for atomicMin/atomicMax
__device__ __forceinline__ u32 enc_f32(float f) {
  const u32 b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(u32 e) {
  const u32 b = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
  return __uint_as_float(b);
}


__global__ void k_set_count(int* counts, int which, int value) {
  if (threadIdx.x == 0 && blockIdx.x == 0) counts[which] = value;
}

// Bounding box of a cloud, first half: workgroup b leaves the order-preserving encodings of its min x,y,z / max x,y,z in
// part[8 b .. 8 b + 5].  The consumer (k2_keys_hist) folds the gridDim.x records: nothing has to be initialised first
// — with atomics on one record a launch of its own had to reset it before every cloud.
__device__ __forceinline__ void d_minmax(const float4* __restrict__ pts, int n, u32* __restrict__ part) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = pts[i];
    mn[0] = fminf(mn[0], p.x);
    mn[1] = fminf(mn[1], p.y);
    mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x);
    mx[1] = fmaxf(mx[1], p.y);
    mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
    }
  }
  __shared__ float red[4][6];
  const int wave = threadIdx.x >> 6;
  if (qk_lane() == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[wave][a] = mn[a];
      red[wave][3 + a] = mx[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    float v = red[0][a];
    for (int w = 1; w < 4; ++w) v = (a < 3) ? fminf(v, red[w][a]) : fmaxf(v, red[w][a]);
    part[8 * blockIdx.x + a] = enc_f32(v);
  }
}
// second half, by every workgroup of the consumer: fold `parts` records into s_mm[6] (LDS); ends with a barrier
#define MM_MAX_PARTS 128
__device__ __forceinline__ void mm_fold(const u32* __restrict__ part, int parts, u32* s_mm) {
  if (threadIdx.x < 64) {
    u32 lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    for (int p = threadIdx.x; p < parts; p += 64) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = min(lo[a], part[8 * p + a]);
        hi[a] = max(hi[a], part[8 * p + 3 + a]);
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        lo[a] = min(lo[a], (u32)__shfl_xor((int)lo[a], off, 64));
        hi[a] = max(hi[a], (u32)__shfl_xor((int)hi[a], off, 64));
      }
    }
    if (threadIdx.x == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        s_mm[a] = lo[a];
        s_mm[3 + a] = hi[a];
      }
    }
  }
  __syncthreads();
}

// =================================================================================================
// Stable LSD radix sort of 64-bit keys by bits [32, 32+key_bits), 8 bits per pass.  One wavefront per
// 1024-element tile: stability needs ranks in element order, which a single wave gets for free from
// ballot match masks and in-order LDS updates (no barriers).  Used with key = (cell index << 32) |
// point index on inputs that are already in ascending point-index order, so equal cells keep ascending
// point order — the accumulation order the oracle defines for pcl::VoxelGrid centroids.
// The three histogram buffers of a sort rotate: pass p reads H[p % 3] (tile-major rows of 256 counters), adds the NEXT
// pass's histogram into H[(p + 1) % 3] while it scatters (each key's destination tile and next digit are known at that
// point: one fire-and-forget global atomic per key instead of a histogram launch per pass) and clears its tile's row of
// H[(p + 2) % 3].  Only the first pass has a histogram kernel of its own.
// key_of(i) yields key i: a load for a plain histogram, or the key computed from its point (then also stored to `store`,
// which fuses the key kernel of a sort into its first histogram: one launch fewer on a chain of ~8 us launches)
template <int BITS, int TILE, typename KeyOf>
__device__ __forceinline__ void d_radix_hist(KeyOf key_of, u64* __restrict__ store, int n, int shift, u32* __restrict__ hist,
                                             int nblk) {
  constexpr int NB = 1 << BITS;
  static_assert(NB == 256 && TILE == 1024, "256 threads: one digit and four keys each");
  __shared__ u32 cnt[NB];
  const int tid = threadIdx.x, blk = blockIdx.x;
  if (blk >= nblk) return;
  hist[((size_t)nblk + blk) * NB + tid] = 0;      // H[1], H[2]: rows of this tile
  hist[((size_t)2 * nblk + blk) * NB + tid] = 0;
  cnt[tid] = 0;
  __syncthreads();
  const int base = blk * TILE;
#pragma unroll
  for (int s = 0; s < TILE / 256; ++s) {
    const int i = base + s * 256 + tid;
    if (i < n) {
      const u64 key = key_of(i);
      if (store) store[i] = key;
      atomicAdd(&cnt[(u32)(key >> shift) & (u32)(NB - 1)], 1u);
    }
  }
  __syncthreads();
  hist[(size_t)blk * NB + tid] = cnt[tid];  // hist[blk][digit]: one coalesced 1 KB row per tile
}

// One workgroup of four wavefronts per 1024-key tile.  Stability needs ranks in element order: wave w owns the w-th
// quarter of the tile, so after a counting phase (per-wave digit counts in LDS) every wave knows where its keys of each
// digit start, and ranks its own 256 keys in order with ballot match masks (four steps of 64 instead of the sixteen a
// single wave needed: the pass is latency-bound, the chip has 45 tiles to chew on).
template <int BITS, int TILE>
__device__ __forceinline__ void d_radix_scatter(const u64* __restrict__ in, u64* __restrict__ out, int n,
                                                      int shift, const u32* __restrict__ hist, int nblk,
                                                      u32* __restrict__ hist_next /* or null: last pass */,
                                                      u32* __restrict__ hist_clear /* or null */) {
  constexpr int NB = 1 << BITS;
  static_assert(NB == 256 && TILE == 1024, "256 threads: one digit and four keys each");
  __shared__ u32 s_tot[4][NB], s_bef[4][NB];  // partial column sums of the tile histograms, one slice per wave
  __shared__ u32 cntw[4][NB];                 // digit counts of each wave's quarter, then its running bases
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x;
  if (blk >= nblk) return;
#define SCATTER_STAMP(pt) if (hist_next && n > 30000) { QTR_STAMP(STAMP_SCATTER, pt) }
  SCATTER_STAMP(0)
  if (hist_clear) hist_clear[(size_t)blk * NB + tid] = 0;
  const int tbase = blk * TILE + wave * 256;
  u64 keys[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int i = tbase + s * 64 + lane;
    keys[s] = (i < n) ? in[i] : 0ULL;
  }
  // offsets from the RAW per-tile histograms (no separate scan launch): digit d of this tile starts at
  //   sum_{d'<d} total[d'] + sum_{b<blk} hist[b][d];   wave w adds up the rows b = w, w+4, ... (4 digits per lane)
  {
    u32 tot[4] = {0, 0, 0, 0}, before[4] = {0, 0, 0, 0};
    // eight rows in flight per round trip (32, i.e. one round for a raw cloud's ~120 tiles, was measured: no faster)
    constexpr int RIF = 8;
    for (int b0 = wave; b0 < nblk; b0 += 4 * RIF) {
      uint4 h[RIF];
#pragma unroll
      for (int q = 0; q < RIF; ++q) h[q] = ((const uint4*)(hist + (size_t)min(b0 + 4 * q, nblk - 1) * NB))[lane];
#pragma unroll
      for (int q = 0; q < RIF; ++q) {
        const int b = b0 + 4 * q;
        const u32 live = (b < nblk) ? 0xffffffffu : 0u, m = (b < blk) ? 0xffffffffu : 0u;
        tot[0] += h[q].x & live;
        tot[1] += h[q].y & live;
        tot[2] += h[q].z & live;
        tot[3] += h[q].w & live;
        before[0] += h[q].x & m;
        before[1] += h[q].y & m;
        before[2] += h[q].z & m;
        before[3] += h[q].w & m;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s_tot[wave][lane * 4 + q] = tot[q];
      s_bef[wave][lane * 4 + q] = before[q];
      cntw[wave][lane * 4 + q] = 0;
    }
  }
  __syncthreads();
  SCATTER_STAMP(1)
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (tbase + s * 64 + lane < n) atomicAdd(&cntw[wave][(u32)(keys[s] >> shift) & (u32)(NB - 1)], 1u);
  // thread = digit: global start of the digit, then the four waves' starts inside it
  const u32 mytot = s_tot[0][tid] + s_tot[1][tid] + s_tot[2][tid] + s_tot[3][tid];
  const u32 mybef = s_bef[0][tid] + s_bef[1][tid] + s_bef[2][tid] + s_bef[3][tid];
  __shared__ u32 s_wsum[4];
  int wtot;
  const int ex = wave_excl_scan_i32((int)mytot, &wtot);
  if (lane == 63) s_wsum[wave] = (u32)wtot;
  __syncthreads();
  {
    u32 run = (u32)ex + mybef;
    for (int w = 0; w < wave; ++w) run += s_wsum[w];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const u32 c = cntw[w][tid];
      cntw[w][tid] = run;
      run += c;
    }
  }
  __syncthreads();
  SCATTER_STAMP(2)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int i = tbase + s * 64 + lane;
    const bool valid = i < n;
    const u64 key = keys[s];
    const u32 d = (u32)(key >> shift) & (u32)(NB - 1);
    u64 m = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < BITS; ++bit) {
      const bool one = (d >> bit) & 1u;
      const u64 b = __ballot(valid && one);
      m &= one ? b : ~b;
    }
    const u32 pos = valid ? cntw[wave][d] + (u32)__popcll(m & lanemask_lt()) : 0u;
    if (valid) out[pos] = key;
    if (hist_next) {
      // one atomic per group of lanes that agree on this digit, the next digit and the destination tile instead of one
      // per key: the upper digits of a grid key take few values, and hundreds of device-scope atomics on one counter
      // were what made the middle pass of a sort twice as long as the last (which has none)
      const u32 d2 = (u32)(key >> (shift + BITS)) & (u32)(NB - 1);
      u64 m2 = m;
#pragma unroll
      for (int bit = 0; bit < BITS; ++bit) {
        const bool one = (d2 >> bit) & 1u;
        const u64 b = __ballot(valid && one);
        m2 &= one ? b : ~b;
      }
      // positions grow with the lane inside a group, so its tiles are those of its first and last lane
      const u32 tile = pos / TILE;
      const int first = valid ? __ffsll((unsigned long long)m2) - 1 : 0, lastl = valid ? 63 - __clzll((long long)m2) : 0;
      const u32 tile_first = (u32)__shfl((int)tile, first, 64), tile_last = (u32)__shfl((int)tile, lastl, 64);
      if (valid) {
        if (tile_first == tile_last) {
          if (lane == first) atomicAdd(&hist_next[(size_t)tile * NB + d2], (u32)__popcll(m2));
        } else {
          atomicAdd(&hist_next[(size_t)tile * NB + d2], 1u);
        }
      }
    }
    __syncthreads();  // orders the LDS reads above before the updates below (each wave only touches its own row)
    if (valid && (m & lanemask_lt()) == 0) cntw[wave][d] += (u32)__popcll(m);
    __syncthreads();
  }
  SCATTER_STAMP(3)
#undef SCATTER_STAMP
}

// out[0..n] = exclusive scan of f(in[0..n)); single workgroup of 1024 threads.  Up to 16384 elements each
// thread owns one contiguous run of K <= 16 values held in registers (one round of loads, two barriers);
// longer inputs take the chunked path (three barriers and one load round trip per 1024 elements).
template <typename F>
__device__ __forceinline__ void d_block_scan(const int* __restrict__ in, int* __restrict__ out, int n, F f) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = (n + 1023) >> 10;
  if (K <= 16) {
    const int base = tid * K;
    int v[16];
    int s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      v[k] = (k < K && base + k < n) ? f(in[base + k]) : 0;
      s += v[k];
    }
    int tot;
    const int ex = wave_excl_scan_i32(s, &tot);
    if (lane == 63) wsum[wave] = tot;
    __syncthreads();
    int run = ex, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      run += (w < wave) ? wsum[w] : 0;
      total += wsum[w];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < K && base + k < n) out[base + k] = run;
      run += v[k];
    }
    if (tid == 0) out[n] = total;
    return;
  }
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int v = i < n ? f(in[i]) : 0;
    int tot;
    const int ex = wave_excl_scan_i32(v, &tot);
    if (lane == 63) wsum[wave] = tot;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    if (i < n) out[i] = carry_s + woff + ex;
    __syncthreads();
    if (tid == 1023) carry_s = carry_s + woff + ex + v;
    __syncthreads();
  }
  if (tid == 0) out[n] = carry_s;
}
__device__ __forceinline__ void d_scan_i32_copy(const int* __restrict__ in, int* __restrict__ out, int n) {
  d_block_scan(in, out, n, [](int x) { return x; });
}

// =================================================================================================
// K1  pcl::VoxelGrid::applyFilter restated
struct VoxGrid {
  float inv;
  int minb[3], divb[3];
  int overflow;
};
__device__ __forceinline__ VoxGrid vox_grid(const u32* mm, float leaf) {
  VoxGrid g;
  g.inv = 1.0f / leaf;
  float mn[3], mx[3];
  long long dprod = 1;
  for (int a = 0; a < 3; ++a) {
    mn[a] = dec_f32(mm[a]);
    mx[a] = dec_f32(mm[3 + a]);
    const long long d = (long long)((mx[a] - mn[a]) * g.inv) + 1;
    dprod *= d;
    g.minb[a] = (int)floorf(mn[a] * g.inv);
    const int maxb = (int)floorf(mx[a] * g.inv);
    g.divb[a] = maxb - g.minb[a] + 1;
  }
  g.overflow = dprod > 2147483647LL;
  return g;
}

__device__ __forceinline__ u64 vox_key(const VoxGrid& g, const float4& p, int i) {
  const int i0 = (int)(floorf(p.x * g.inv) - (float)g.minb[0]);
  const int i1 = (int)(floorf(p.y * g.inv) - (float)g.minb[1]);
  const int i2 = (int)(floorf(p.z * g.inv) - (float)g.minb[2]);
  const int idx = i0 + i1 * g.divb[0] + i2 * (g.divb[0] * g.divb[1]);
  return ((u64)(u32)idx << 32) | (u32)i;
}
// the flags a voxel grid hands to the rest of the chain (one thread of the key launch)
__device__ __forceinline__ void vox_grid_counts(const VoxGrid& g, int* __restrict__ counts) {
  if (g.overflow) counts[CNT_VOX_OVERFLOW] = 1;
  const long long cells = (long long)g.divb[0] * g.divb[1] * g.divb[2];  // every key is below this
  counts[CNT_SORT_BITS] = (g.overflow || cells <= 1) ? (g.overflow ? 32 : 1) : 64 - __clzll(cells - 1);
}

// heads per 1024-element block
// Points per workgroup of the centroid kernel.  One pair at a time the launch is a chain of latencies (keys -> gathered points ->
// LDS -> look-back -> the runs' sums): 256-point tiles — 1024 workgroups on the two clouds of a scan pair instead of 256, four
// per compute unit — take 7 us off it (profiles/r6_ab.txt section 16).  A tile reads its halo again (768 points for 256), which
// the batched path, bound by instruction issue, pays for: it keeps 1024-point tiles.
#define VOX_TILE_SINGLE 256
#define VOX_TILE_BATCH 1024
#define VOX_HALO 512
template <int VOX_TILE>
__device__ __forceinline__ void d_vox_centroids(const u64* __restrict__ keys, const float4* __restrict__ pts,
                                                       int P, int* __restrict__ look, float4* __restrict__ out,
                                                       int cap, int nblk, int* __restrict__ counts,
                                                       int* __restrict__ mail, int* __restrict__ mail_seq_slot,
                                                       int seq) {
  // element t of the window lives in s_p[t + 1] as (x, y, z, cell id bits): one 16-byte LDS read per element
  __shared__ float4 s_p[VOX_TILE + VOX_HALO + 1];
  __shared__ int wtot[4];
  if (blockIdx.x >= nblk) return;
  const int lane = qk_lane(), wave = threadIdx.x >> 6;
  const int base = blockIdx.x * VOX_TILE;
  QTR_STAMP(STAMP_CENTROIDS, 0)
  {
    // two rounds of independent loads (keys, then the gathered points) instead of six dependent pairs
    constexpr int NLD = (VOX_TILE + VOX_HALO) / 256;
    u64 kk[NLD];
    float4 pp[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int i = base + q * 256 + threadIdx.x;
      kk[q] = (i < P) ? keys[i] : ~0ULL;
    }
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int i = base + q * 256 + threadIdx.x;
      pp[q] = (i < P) ? pts[(u32)kk[q]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int t = q * 256 + threadIdx.x;
      pp[q].w = __uint_as_float((u32)(kk[q] >> 32));  // 0xffffffff past the end of the cloud
      s_p[t + 1] = pp[q];
    }
  }
  QTR_STAMP(STAMP_CENTROIDS, 1)
  if (threadIdx.x == 0) s_p[0].w = __uint_as_float((base > 0) ? (u32)(keys[base - 1] >> 32) : 0xffffffffu);
  __syncthreads();
  // first output slot of this tile and the voxel count of the cloud: the tile's own run heads are counted from the window
  // that has just landed in LDS, published, and the counts of the tiles in front are added up as they appear
  // (tail_lookback; until round 5 a launch of its own — k2_vox_headcount — counted the heads first and every workgroup
  // of this kernel added up its table: 4.7 us + a dispatch on the chain).  The LAST tile knows the total and hands the
  // counters to the host.
  __shared__ int s_hc[4], s_red[5];
  int running, total = 0;
  {
    int c = 0;
#pragma unroll
    for (int t0 = 0; t0 < VOX_TILE; t0 += 256) {
      const int t = t0 + threadIdx.x, i = base + t;
      c += ((i < P) && (i == 0 || __float_as_uint(s_p[t + 1].w) != __float_as_uint(s_p[t].w))) ? 1 : 0;
    }
    c = wave_sum_i32(c);
    if (lane == 0) s_hc[wave] = c;
    __syncthreads();
    const int mine = (s_hc[0] + s_hc[1]) + (s_hc[2] + s_hc[3]);
    running = tail_lookback(look, (int)blockIdx.x, mine, s_red);
    total = running + mine;
  }
  if ((int)blockIdx.x == nblk - 1) {  // this is the last voxelise kernel: the last tile hands the counters to the host
    if (running < 0) total = -1;      // (a tile in front never published its count: the host refuses the cloud)
    if (threadIdx.x == 0) counts[CNT_NVOX] = total;
    if (mail && threadIdx.x < 16) {
      mail_store_line(mail, threadIdx.x, (threadIdx.x == CNT_NVOX) ? total : counts[threadIdx.x], seq);
      __threadfence_system();
      if (threadIdx.x == 0) *mail_seq_slot = seq;
    }
  }
  if (running < 0) {
    // no writes at unknown offsets — and the failure is STICKY: the last tile's own look-back may still come out whole
    // (a slow predecessor can appear after a middle tile has given up on it but before the last tile does), so every tile
    // that gives up raises a word of the counter line; it reaches the host with the counters the matcher's tail mails
    // (after this kernel in stream order) and in qtr_voxelize's read-back — as the matcher's MC_TAILERR does
    if (threadIdx.x == 0) __hip_atomic_store(counts + CNT_VOX_TAILERR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  QTR_STAMP(STAMP_CENTROIDS, 2)
  for (int t0 = 0; t0 < VOX_TILE; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const int i = base + t;
    const u32 cell = __float_as_uint(s_p[t + 1].w);
    const bool head = (i < P) && (i == 0 || cell != __float_as_uint(s_p[t].w));
    const u64 bal = __ballot(head);
    if (lane == 0) wtot[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wtot[w];
    const int tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (head) {
      const int slot = running + woff + __popcll(bal & lanemask_lt());
      float cx = 0.f, cy = 0.f, cz = 0.f;
      int e = t;
      // the additions of a run are sequential by definition (float, sorted order); what can overlap is the LDS
      // traffic: fetch eight candidates at a time, then add the ones that still belong to the run
      bool more = true;
      while (more && e + 8 <= VOX_TILE + VOX_HALO) {
        float4 c8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) c8[q] = s_p[e + 1 + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (more && __float_as_uint(c8[q].w) == cell) {
            cx += c8[q].x;
            cy += c8[q].y;
            cz += c8[q].z;
            ++e;
          } else {
            more = false;
          }
        }
      }
      while (more && e < VOX_TILE + VOX_HALO && __float_as_uint(s_p[e + 1].w) == cell) {
        cx += s_p[e + 1].x;
        cy += s_p[e + 1].y;
        cz += s_p[e + 1].z;
        ++e;
      }
      int g = base + e;
      if (e == VOX_TILE + VOX_HALO) {  // run longer than the halo: continue from global memory
        bool more = true;
        while (more) {  // eight keys, then eight gathered points per round trip; additions stay in order
          u64 k8[8];
          float4 p8[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) k8[q] = (g + q < P) ? keys[g + q] : ~0ULL;
#pragma unroll
          for (int q = 0; q < 8; ++q)
            p8[q] = ((u32)(k8[q] >> 32) == cell) ? pts[(u32)k8[q]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (more && (u32)(k8[q] >> 32) == cell) {
              cx += p8[q].x;
              cy += p8[q].y;
              cz += p8[q].z;
              ++g;
            } else {
              more = false;
            }
          }
        }
      }
      const float cnt = (float)(g - i);
      if (slot < cap) out[slot] = make_float4(cx / cnt, cy / cnt, cz / cnt, 0.f);
    }
    running += tot;
    __syncthreads();
  }
  QTR_STAMP(STAMP_CENTROIDS, 3)
}

// =================================================================================================
// Radius-neighbour lists.  Semantics of pcl::search::KdTree::radiusSearch (FLANN RadiusResultSet,
// sorted): d2 = ((dx^2)+dy^2)+dz^2 in float with the query first, kept iff d2 < float(r*r), sorted by
// (d2, index), query included.
struct CellGrid {
  float mn[3];
  float cell;
};
__device__ __forceinline__ void cell_of(const CellGrid& g, const float4& p, int* c) {
  c[0] = min(255, max(0, (int)floorf((p.x - g.mn[0]) / g.cell)));
  c[1] = min(255, max(0, (int)floorf((p.y - g.mn[1]) / g.cell)));
  c[2] = min(255, max(0, (int)floorf((p.z - g.mn[2]) / g.cell)));
}
__device__ __forceinline__ u32 cell_key(int cx, int cy, int cz) { return ((u32)cz << 16) | ((u32)cy << 8) | (u32)cx; }

__device__ __forceinline__ CellGrid cell_grid(const u32* __restrict__ mm, float cell) {
  CellGrid g;
  g.mn[0] = dec_f32(mm[0]);
  g.mn[1] = dec_f32(mm[1]);
  g.mn[2] = dec_f32(mm[2]);
  g.cell = cell;
  return g;
}
__device__ __forceinline__ u64 cell_sort_key(const CellGrid& g, const float4& p, int i) {
  int c[3];
  cell_of(g, p, c);
  return ((u64)cell_key(c[0], c[1], c[2]) << 32) | (u32)i;
}

__device__ __forceinline__ int lower_bound_hi(const u64* keys, int n, u32 k) {  // first i with hi(keys[i]) >= k
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((u32)(keys[mid] >> 32) < k)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// the same for two keys at once, k0 <= k1: the two chains of dependent loads advance together (a fixed number of
// halvings, no branch), so the pair costs the round trips of one — k2_ranges is nothing but these round trips (85 % of
// its wave-cycles parked: profiles/r6_pmc_fpfh.json)
__device__ __forceinline__ void lower_bound_hi2(const u64* __restrict__ keys, int n, u32 k0, u32 k1, int* r0, int* r1) {
  if (n <= 0) {
    *r0 = *r1 = 0;
    return;
  }
  int b0 = 0, b1 = 0, len = n;
  while (len > 1) {
    const int half = len >> 1;
    const u32 a0 = (u32)(keys[b0 + half - 1] >> 32), a1 = (u32)(keys[b1 + half - 1] >> 32);
    b0 = (a0 < k0) ? b0 + half : b0;
    b1 = (a1 < k1) ? b1 + half : b1;
    len -= half;
  }
  const u32 a0 = (u32)(keys[b0] >> 32), a1 = (u32)(keys[b1] >> 32);
  *r0 = b0 + ((a0 < k0) ? 1 : 0);
  *r1 = b1 + ((a1 < k1) ? 1 : 0);
}

// ---- the neighbour-search grid as a DENSE cell table (round 6).  Until then the voxel centroids were sorted by packed cell
// keys (a key launch + three radix passes, 31 us of the single registration's chain) and every point found its nine key ranges
// by binary search (k2_ranges: 11 us of dependent round trips).  Nothing downstream needs the ORDER inside a cell — the lists
// are sorted by (d^2, index) afterwards — so a counting sort does: a point takes a place in its cell with one atomic
// (k2_cell_count: the old value is its rank in the cell; a cell holds a handful of points, no hot address), the cells' counts
// are scanned (k2_cell_scan: 2048 cells per workgroup, the workgroups' totals added up through tail_lookback; the counts are
// left ZERO for the next registration), and a point's place is start[cell] + rank, its nine ranges two table reads each
// (k2_cell_place).  The grid is the bounding box of the RAW cloud (the voxel stage left it in C.mm, as before) cut into cells of
// the FPFH radius; the voxel stage mails its cell count (CNT_NCELL) with the voxel counts, and a chain whose grids exceed
// QTR_CELL_CAP cells (or that has no voxel stage in front: qtr_fpfh) sorts keys as before.
struct CellDims {
  int nx, ny, nz;
};
__device__ __forceinline__ CellDims cell_dims(const u32* __restrict__ mm, float cell) {
  const CellGrid g = cell_grid(mm, cell);
  int c[3];
  cell_of(g, make_float4(dec_f32(mm[3]), dec_f32(mm[4]), dec_f32(mm[5]), 0.f), c);
  return CellDims{c[0] + 1, c[1] + 1, c[2] + 1};
}
#define CELL_SCAN_TILE 2048
__device__ __forceinline__ void d_cell_count(const float4* __restrict__ pts, int n, const u32* __restrict__ mm, float cell,
                                             int* __restrict__ cnt, u64* __restrict__ place, int* __restrict__ look) {
  const CellDims d = cell_dims(mm, cell);
  const CellGrid g = cell_grid(mm, cell);
  const int ncell = d.nx * d.ny * d.nz;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  // k2_cell_scan's look-back words ("not published yet")
  for (int w = t; w < (ncell + CELL_SCAN_TILE) / CELL_SCAN_TILE + 1; w += gridDim.x * blockDim.x) look[w] = 0;
  if (t >= n) return;
  int c[3];
  cell_of(g, pts[t], c);
  const int lin = min(c[0], d.nx - 1) + d.nx * (min(c[1], d.ny - 1) + d.ny * min(c[2], d.nz - 1));
  const int r = atomicAdd(&cnt[lin], 1);
  place[t] = ((u64)(u32)lin << 32) | (u32)r;
}
// start[e] = number of points in cells below e, for e = 0 .. ncell (and beyond: the launch's last tile); cnt[] back to zero
__device__ __forceinline__ void d_cell_scan(const u32* __restrict__ mm, float cell, int* __restrict__ cnt, int* __restrict__ start,
                                            int* __restrict__ look, int* __restrict__ counts) {
  const CellDims d = cell_dims(mm, cell);
  const int ncell = d.nx * d.ny * d.nz;
  const int nblk = (ncell + CELL_SCAN_TILE) / CELL_SCAN_TILE;  // covers entry ncell itself
  if ((int)blockIdx.x >= nblk) return;
  __shared__ int s_red[5], s_w[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e0 = blockIdx.x * CELL_SCAN_TILE + tid * 8;
  int4 a = *(const int4*)(cnt + e0), b = *(const int4*)(cnt + e0 + 4);  // (cells past the grid hold zero: nobody counted there)
  const int4 z = make_int4(0, 0, 0, 0);
  *(int4*)(cnt + e0) = z;
  *(int4*)(cnt + e0 + 4) = z;
  const int mine = (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
  int wtot;
  const int ex = wave_excl_scan_i32(mine, &wtot);
  if (lane == 0) s_w[wave] = wtot;
  __syncthreads();
  const int total = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
  int before = tail_lookback(look, (int)blockIdx.x, total, s_red);
  if (before < 0) {  // a predecessor's count never appeared (bounded wait): the chain's result is refused by the host
    if (tid == 0) __hip_atomic_store(counts + CNT_VOX_TAILERR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    before = 0;  // (places stay inside the arrays: every start is at most the cloud's size)
  }
  int run = before + ex;
  for (int w = 0; w < wave; ++w) run += s_w[w];
  int4 sa, sb;
  sa.x = run;
  sa.y = sa.x + a.x;
  sa.z = sa.y + a.y;
  sa.w = sa.z + a.z;
  sb.x = sa.w + a.w;
  sb.y = sb.x + b.x;
  sb.z = sb.y + b.y;
  sb.w = sb.z + b.z;
  *(int4*)(start + e0) = sa;
  *(int4*)(start + e0 + 4) = sb;
}
// the points in cell order (w carries the original index), and the nine candidate ranges of every point
__device__ __forceinline__ void d_cell_place(const float4* __restrict__ pts, int n, const u32* __restrict__ mm, float cell,
                                             const int* __restrict__ start, const u64* __restrict__ place,
                                             float4* __restrict__ spts, int* __restrict__ ranges) {
  const CellDims d = cell_dims(mm, cell);
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) {
    const u64 pl = place[g];
    float4 p = pts[g];
    p.w = __uint_as_float((u32)g);
    spts[start[(int)(pl >> 32)] + (int)(u32)pl] = p;
  }
  if (g >= n * 9) return;
  const int i = g / 9, r = g - i * 9;
  const CellGrid cg = cell_grid(mm, cell);
  int c[3];
  cell_of(cg, pts[i], c);
  const int cx = min(c[0], d.nx - 1), cy = min(c[1], d.ny - 1) + (r % 3) - 1, cz = min(c[2], d.nz - 1) + (r / 3) - 1;
  int s = 0, e = 0;
  if (cy >= 0 && cy < d.ny && cz >= 0 && cz < d.nz) {
    const int row = d.nx * (cy + d.ny * cz);
    s = start[row + max(cx - 1, 0)];
    e = start[row + min(cx + 1, d.nx - 1) + 1];
  }
  ranges[2 * g] = s;
  ranges[2 * g + 1] = e;
}

// points gathered into cell-sorted order (w carries the original index) so that candidate loads are
// contiguous 16-byte reads instead of a dependent key -> point gather
__device__ __forceinline__ void d_sorted_points(const float4* __restrict__ pts, const u64* __restrict__ sorted,
                                                       int n, float4* __restrict__ spts) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const u32 j = (u32)sorted[t];
    float4 p = pts[j];
    p.w = __uint_as_float(j);
    spts[t] = p;
  }
}

// the nine contiguous key ranges (rows cy-1..cy+1 x cz-1..cz+1, cells cx-1..cx+1) of every query point:
// one thread per (point, range) so the binary searches of the whole cloud overlap
__device__ __forceinline__ void d_ranges(const float4* __restrict__ pts, int n, const u64* __restrict__ sorted,
                                                const u32* __restrict__ mm, float cell, int* __restrict__ ranges) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n * 9) return;
  const int i = g / 9, r = g - i * 9;
  CellGrid cg;
  cg.mn[0] = dec_f32(mm[0]);
  cg.mn[1] = dec_f32(mm[1]);
  cg.mn[2] = dec_f32(mm[2]);
  cg.cell = cell;
  int c[3];
  cell_of(cg, pts[i], c);
  const int cy = c[1] + (r % 3) - 1, cz = c[2] + (r / 3) - 1;
  int s = 0, e = 0;
  if (cy >= 0 && cy <= 255 && cz >= 0 && cz <= 255) {
    const u32 klo = cell_key(max(c[0] - 1, 0), cy, cz), khi = cell_key(min(c[0] + 1, 255), cy, cz);
    lower_bound_hi2(sorted, n, klo, khi + 1u, &s, &e);
  }
  ranges[2 * g] = s;
  ranges[2 * g + 1] = e;
}

// one wavefront (= one workgroup) per query point
__device__ __forceinline__ void d_neighbors(const float4* __restrict__ pts, int n, const float4* __restrict__ spts,
                                                  const int* __restrict__ ranges, float r2,
                                                  int* __restrict__ nbr_cnt, int* __restrict__ nbr_idx,
                                                  float* __restrict__ nbr_d2, int* __restrict__ counts) {
  __shared__ u64 buf[QTR_KMAX];
  __shared__ int rs[9], pre[10];
  const int lane = threadIdx.x;
  const int i = blockIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  {
    int len = 0, s = 0;
    if (lane < 9) {
      s = ranges[18 * i + 2 * lane];
      len = ranges[18 * i + 2 * lane + 1] - s;
      rs[lane] = s;
    }
    int tot;
    const int ex = wave_excl_scan_i32(len, &tot);
    if (lane < 9) pre[lane] = ex;
    if (lane == 9) pre[9] = tot;
  }
  __syncthreads();
  const int total = pre[9];
  int k = 0;
  bool overflow = false;
  // (candidates chunk by chunk: batching four chunks per round trip does not pay here — 32 single-wave workgroups per
  // compute unit already cover each other's latency; the same change made k2_spfh slower, k2_normals — half a wave per
  // SIMD — faster)
  for (int c0 = 0; c0 < total; c0 += 64) {
    const int c = c0 + lane;
    bool ok = false;
    u64 key = 0;
    if (c < total) {
      int r = 0;
#pragma unroll
      for (int q = 1; q < 9; ++q) r += (c >= pre[q]);
      const float4 qp = spts[rs[r] + (c - pre[r])];
      float d2 = 0.f, d;
      d = p.x - qp.x;
      d2 += d * d;
      d = p.y - qp.y;
      d2 += d * d;
      d = p.z - qp.z;
      d2 += d * d;
      ok = d2 < r2;
      key = ((u64)__float_as_uint(d2) << 32) | __float_as_uint(qp.w);
    }
    const u64 bal = __ballot(ok);
    if (ok) {
      const int pos = k + __popcll(bal & lanemask_lt());
      if (pos < QTR_KMAX)
        buf[pos] = key;
      else
        overflow = true;
    }
    k += __popcll(bal);
  }
  if (__ballot(overflow) || k > QTR_KMAX) {
    // more neighbours than the slot holds (a dense, un-voxelised cloud): only the count is left behind; k2_neighbors_big
    // searches this point again and keeps its list in the long-list arena (pcl's radius search has no cap, reference
    // src/teaser_utils/fpfh.cc:58-72)
    // (the count is left NEGATIVE: "list pending".  Readers take a negative count as an empty list, so a chain that
    // runs without k2_neighbors_big, or whose arena is too small, stays inside its buffers until the host has seen
    // the flag.)
    if (lane == 0) {
      counts[CNT_NBR_OVERFLOW] = 1;
      atomicMax(&counts[CNT_KMAX], k);
      nbr_cnt[i] = -k;
    }
    return;
  }
  if (k <= 64) {
    // Up to 64 keys (every list of a voxelised scan: ~18 entries): sorted by RANK, no network.  A lane keeps its key in
    // registers and counts the keys below it — the keys are read four at a time from LDS, every lane the same address (a
    // broadcast), one 64-bit compare and one add-with-carry per key; keys are distinct (the index is part of them), so the
    // counts are a permutation and the lane stores its entry at its rank.  The bitonic network this replaces for short
    // lists spent 10 - 15 stages of (barrier, two LDS reads, compare, two conditional LDS writes) with a scalar loop around
    // each: ~120 vector + ~180 scalar + ~60 LDS instructions per point against ~3 per key here (round 6, profiles/
    // r6_pmc_fpfh.json: the kernel issues as many scalar as vector instructions and its SIMDs are 65 % busy).
    __syncthreads();  // (one wave: the list in buf is complete)
    const u64 mine = (lane < k) ? buf[lane] : ~0ULL;
    __syncthreads();
    if (lane < 4) buf[k + lane] = ~0ULL;  // (k + 3 < QTR_KMAX; a pad never counts: nothing is above it)
    __syncthreads();
    int rank = 0;
    for (int j = 0; j < k; j += 4) {
      const u64 a = buf[j], b = buf[j + 1], c = buf[j + 2], d = buf[j + 3];
      rank += (int)(a < mine) + (int)(b < mine) + (int)(c < mine) + (int)(d < mine);
    }
    if (lane < k) {
      nbr_idx[(size_t)i * QTR_KMAX + rank] = (int)(u32)mine;
      nbr_d2[(size_t)i * QTR_KMAX + rank] = __uint_as_float((u32)(mine >> 32));
    }
    if (lane == 0) nbr_cnt[i] = k;
    return;
  }
  int n2 = 128;  // (longer lists: the bitonic network in LDS)
  while (n2 < k) n2 <<= 1;
  for (int t = k + lane; t < n2; t += 64) buf[t] = ~0ULL;
  // bitonic sort of n2 (<= 256) packed keys in LDS: ascending (d2, index)
  for (int kk = 2; kk <= n2; kk <<= 1) {
    for (int j = kk >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = lane; t < n2; t += 64) {
        const int x = t ^ j;
        if (x > t) {
          const u64 a = buf[t], b = buf[x];
          const bool up = ((t & kk) == 0);
          if ((a > b) == up) {
            buf[t] = b;
            buf[x] = a;
          }
        }
      }
    }
  }
  __syncthreads();
  for (int t = lane; t < k; t += 64) {
    const u64 key = buf[t];
    nbr_idx[(size_t)i * QTR_KMAX + t] = (int)(u32)key;
    nbr_d2[(size_t)i * QTR_KMAX + t] = __uint_as_float((u32)(key >> 32));
  }
  if (lane == 0) nbr_cnt[i] = k;  // (no per-point global atomics: one hot address caps at ~90 updates/us)
}

// =================================================================================================
// K2  normals: pcl::NormalEstimation::computeFeature -> computePointNormal ->
// computeMeanAndCovarianceMatrix (float, single pass) -> solvePlaneParameters -> pcl::eigen33.
__device__ __forceinline__ void dev_roots2(float b, float c, float* roots) {
  roots[0] = 0.f;
  float d = (float)((double)(b * b) - 4.0 * (double)c);
  if (d < 0.0f) d = 0.0f;
  const float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}
__device__ __forceinline__ void dev_swapf(float& a, float& b) {
  const float t = a;
  a = b;
  b = t;
}
__device__ void dev_roots(const float* m, float* roots) {
  const float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] -
                   m[8] * m[1] * m[1];
  const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  const float c2 = m[0] + m[4] + m[8];
  if (fabsf(c0) < 1.1920928955078125e-07f) {  // FLT_EPSILON
    dev_roots2(c2, c1, roots);
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
  if (roots[0] >= roots[1]) dev_swapf(roots[0], roots[1]);
  if (roots[1] >= roots[2]) {
    dev_swapf(roots[1], roots[2]);
    if (roots[0] >= roots[1]) dev_swapf(roots[0], roots[1]);
  }
  if (roots[0] <= 0.f) dev_roots2(c2, c1, roots);
}
__device__ __forceinline__ void dev_cross(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// pcl::NormalEstimation tail: the nine single-pass float sums of k neighbours -> covariance -> pcl::eigen33
// smallest eigenpair -> viewpoint flip.  (nx, ny, nz, curvature)
__device__ __forceinline__ float4 normal_from_sums(float (&acc)[9], int k, const float4 p) {
  const float kk = (float)k;
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] /= kk;
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
  // pcl::eigen33 (smallest eigenpair)
  float scale = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) scale = fmaxf(scale, fabsf(cov[t]));
  if (scale <= 1.17549435e-38f) scale = 1.0f;  // FLT_MIN
  float s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = cov[t] / scale;
  float roots[3];
  dev_roots(s, roots);
  const float ev = roots[0] * scale;
  s[0] -= roots[0];
  s[4] -= roots[0];
  s[8] -= roots[0];
  float v1[3], v2[3], v3[3];
  dev_cross(&s[0], &s[3], v1);
  dev_cross(&s[0], &s[6], v2);
  dev_cross(&s[3], &s[6], v3);
  const float l1 = v1[0] * v1[0] + (v1[1] * v1[1] + v1[2] * v1[2]);
  const float l2 = v2[0] * v2[0] + (v2[1] * v2[1] + v2[2] * v2[2]);
  const float l3 = v3[0] * v3[0] + (v3[1] * v3[1] + v3[2] * v3[2]);
  float vx, vy, vz, l;
  if (l1 >= l2 && l1 >= l3) {
    vx = v1[0];
    vy = v1[1];
    vz = v1[2];
    l = l1;
  } else if (l2 >= l1 && l2 >= l3) {
    vx = v2[0];
    vy = v2[1];
    vz = v2[2];
    l = l2;
  } else {
    vx = v3[0];
    vy = v3[1];
    vz = v3[2];
    l = l3;
  }
  const float sl = sqrtf(l);
  vx = vx / sl;
  vy = vy / sl;
  vz = vz / sl;
  const float eig_sum = cov[0] + cov[4] + cov[8];
  const float curv = (eig_sum != 0.f) ? fabsf(ev / eig_sum) : 0.f;
  const float wx = 0.f - p.x, wy = 0.f - p.y, wz = 0.f - p.z;
  const float cos_theta = (wx * vx + wy * vy + wz * vz);
  if (cos_theta < 0) {
    vx *= -1;
    vy *= -1;
    vz *= -1;
  }
  return make_float4(vx, vy, vz, curv);
}
// The list of point i (k entries): its fixed-stride slot, or — longer than QTR_KMAX — the long-list arena at the offset
// the slot's first word holds.
struct NbrLists {
  const int* idx;
  const float* d2;
  const int* big_idx;
  const float* big_d2;
};
__device__ __forceinline__ const int* nbr_list_idx(const NbrLists& N, int i, int k) {
  return k <= QTR_KMAX ? N.idx + (size_t)i * QTR_KMAX : N.big_idx + (size_t)N.idx[(size_t)i * QTR_KMAX];
}
__device__ __forceinline__ const float* nbr_list_d2(const NbrLists& N, int i, int k) {
  return k <= QTR_KMAX ? N.d2 + (size_t)i * QTR_KMAX : N.big_d2 + (size_t)N.idx[(size_t)i * QTR_KMAX];
}

// Lists longer than QTR_KMAX: one 256-thread workgroup per such point (every other workgroup returns at once, and the
// launch is skipped altogether when the cloud comes from this library's voxel grid with a leaf that bounds the count).
// The candidates of the nine cell ranges are tested again, the hits appended to a buffer — LDS up to NBIG_LDS_KEYS,
// beyond that the arena itself —, sorted by (d2, index) with the same bitonic network as the short lists, and stored at
// an offset handed out by one atomic per long list.
template <typename KeyPtr>
__device__ __forceinline__ void nbig_bitonic(KeyPtr buf, int n2, int tid) {
  for (int kk = 2; kk <= n2; kk <<= 1) {
    for (int j = kk >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = tid; t < n2; t += 256) {
        const int x = t ^ j;
        if (x > t) {
          const u64 a = buf[t], b = buf[x];
          const bool up = ((t & kk) == 0);
          if ((a > b) == up) {
            buf[t] = b;
            buf[x] = a;
          }
        }
      }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void d_neighbors_big(const float4* __restrict__ pts, int n, const float4* __restrict__ spts,
                                                const int* __restrict__ ranges, float r2, int* __restrict__ nbr_cnt,
                                                int* __restrict__ nbr_idx, int* __restrict__ big_idx,
                                                float* __restrict__ big_d2, int big_cap, int* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) u64 nbig_buf[];  // [NBIG_LDS_KEYS]
  __shared__ int rs[9], pre[10], s_k, s_off;
  const int tid = threadIdx.x, lane = tid & 63;
  const int i = blockIdx.x;
  if (i >= n || counts[CNT_NBR_OVERFLOW] == 0) return;
  const int k = -nbr_cnt[i];  // negative count: a list k2_neighbors left pending
  if (k <= QTR_KMAX) return;
  int n2 = 512;
  while (n2 < k) n2 <<= 1;
  const int kpad = (k + 63) & ~63;
  const bool in_lds = n2 <= NBIG_LDS_KEYS;
  if (tid == 0) {
    // the arena holds the list (k entries, padded to 64) and — for lists too long for LDS — the sort buffer behind it
    // (n2 eight-byte keys = 2 n2 entries of the index array)
    const int need = kpad + (in_lds ? 0 : 2 * n2);
    const int off = atomicAdd(&counts[CNT_NBR_ARENA], need);
    s_off = (off >= 0 && off <= big_cap - need) ? off : -1;
    s_k = 0;
  }
  if (tid < 64) {
    int len = 0, s = 0;
    if (lane < 9) {
      s = ranges[18 * i + 2 * lane];
      len = ranges[18 * i + 2 * lane + 1] - s;
      rs[lane] = s;
    }
    int tot;
    const int ex = wave_excl_scan_i32(len, &tot);
    if (lane < 9) pre[lane] = ex;
    if (lane == 9) pre[9] = tot;
  }
  __syncthreads();
  const int off = s_off;
  if (off < 0) {  // arena exhausted: the host reports QTR_ERR_CAPACITY (qtr_limits.max_long_neighbors)
    if (tid == 0) counts[CNT_NBR_CAPACITY] = 1;
    return;
  }
  u64* gbuf = (u64*)(big_idx + off + kpad);  // (8-byte aligned: off and kpad are multiples of 64 four-byte entries)
  const float4 p = pts[i];
  const int total = pre[9];
  for (int c0 = 0; c0 < total; c0 += 256) {
    const int c = c0 + tid;
    bool ok = false;
    u64 key = 0;
    if (c < total) {
      int r = 0;
#pragma unroll
      for (int q = 1; q < 9; ++q) r += (c >= pre[q]);
      const float4 qp = spts[rs[r] + (c - pre[r])];
      float d2 = 0.f, d;
      d = p.x - qp.x;
      d2 += d * d;
      d = p.y - qp.y;
      d2 += d * d;
      d = p.z - qp.z;
      d2 += d * d;
      ok = d2 < r2;
      key = ((u64)__float_as_uint(d2) << 32) | __float_as_uint(qp.w);
    }
    const u64 bal = __ballot(ok);
    int base = 0;
    if (lane == 0 && bal) base = atomicAdd(&s_k, __popcll(bal));
    base = __shfl(base, 0, 64);
    if (ok) {
      const int pos = base + __popcll(bal & lanemask_lt());  // (any order: the sort below defines it)
      if (in_lds) nbig_buf[pos] = key;
      else gbuf[pos] = key;
    }
  }
  __syncthreads();
  for (int t = k + tid; t < n2; t += 256) {
    if (in_lds) nbig_buf[t] = ~0ULL;
    else gbuf[t] = ~0ULL;
  }
  if (in_lds) nbig_bitonic(nbig_buf, n2, tid);
  else nbig_bitonic(gbuf, n2, tid);
  for (int t = tid; t < k; t += 256) {
    const u64 key = in_lds ? nbig_buf[t] : gbuf[t];
    big_idx[off + t] = (int)(u32)key;
    big_d2[off + t] = __uint_as_float((u32)(key >> 32));
  }
  if (tid == 0) {
    nbr_idx[(size_t)i * QTR_KMAX] = off;
    nbr_cnt[i] = k;
  }
}

__device__ __forceinline__ void d_normals(const float4* __restrict__ pts, int n, const int* __restrict__ nbr_cnt,
                                                 const NbrLists NL, float rn2, float4* __restrict__ normals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int kf = max(nbr_cnt[i], 0);
  const int* idx = nbr_list_idx(NL, i, kf);
  const float* d2 = nbr_list_d2(NL, i, kf);
  // A thread walks its own list: written as `while (d2[k] < rn2) ++k` and `acc += pts[idx[t]]` the kernel is a chain of
  // ~3 k dependent round trips (k ~ 8: most of its 16 us).  Eight entries per round trip instead — the additions stay in
  // list order.
  int k = 0;
  for (int t0 = 0; t0 < kf; t0 += 8) {  // length of the prefix with d2 < rn2 (the list is sorted by (d2, index))
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = d2[min(t0 + q, kf - 1)];
    int c = 0;
    bool run = true;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      run = run && (t0 + q < kf) && (v[q] < rn2);
      c += run ? 1 : 0;
    }
    k += c;
    if (c < 8) break;
  }
  const float qnan = __uint_as_float(0x7fc00000u);
  if (k < 3) {
    normals[i] = make_float4(qnan, qnan, qnan, qnan);
    return;
  }
  const float4 self = pts[i];
  float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int t0 = 0; t0 < k; t0 += 8) {
    int jj[8];
    float4 qq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) jj[q] = idx[min(t0 + q, k - 1)];
#pragma unroll
    for (int q = 0; q < 8; ++q) qq[q] = pts[jj[q]];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (t0 + q < k) {
        const float4 p = qq[q];
        acc[0] += p.x * p.x;
        acc[1] += p.x * p.y;
        acc[2] += p.x * p.z;
        acc[3] += p.y * p.y;
        acc[4] += p.y * p.z;
        acc[5] += p.z * p.z;
        acc[6] += p.x;
        acc[7] += p.y;
        acc[8] += p.z;
      }
  }
  normals[i] = normal_from_sums(acc, k, self);
}

// =================================================================================================
// K3  SPFH (pcl::computePairFeatures + computePointSPFHSignature).  One wavefront per point: lanes
// evaluate neighbours in parallel and count histogram hits in LDS; the float histogram value is then
// rebuilt as `count` sequential additions of hist_incr, which is exactly what the sequential loop
// produces (every addend is the same constant, so the order of the hits does not matter).
__device__ __forceinline__ float dot4_sse(const float* a, const float* b) {
  return (a[0] * b[0] + a[2] * b[2]) + (a[1] * b[1] + 0.0f);
}
__device__ __forceinline__ int bin11(double x) {
  if (x != x) return 0;
  const double fl = floor(x);
  if (fl < 0.0) return 0;
  if (fl >= 11.0) return 10;
  return (int)fl;
}
// acosf(|angle1|) > acosf(|angle2|), see dev_pair_features
__device__ __forceinline__ bool spfh_swap_roles(float angle1, float angle2) {
  const float x1 = fabsf(angle1), x2 = fabsf(angle2);
  if (x1 == x2) return false;
  if (x1 <= 1.0f && x2 <= 1.0f && fabsf(x1 - x2) > 5e-7f) return x1 < x2;
  return qm_acosf(x1) > qm_acosf(x2);
}
// The bin of f1 = atan2f(y, x) in the first 11-bin block — floor(11 (f1 + pi) / (2 pi)), evaluated by d_spfh in binary64 from
// the ROUNDED binary32 angle, with d_pi = 1.0f / (2.0f * (float)M_PI) — decided WITHOUT the arc tangent (qm_atan2f: five
// binary64 divisions and three square roots for one of eleven answers) wherever that is safe; -1 = "not sure": the caller
// evaluates the function.  The bin boundaries in f1 are T_k = k / (11 d_pi) - pi, k = 1..10: T_5 < 0 < T_6, and T_k and
// -T_(11-k) differ by at most 2.3e-7 (d_pi is a rounded 1 / 2 pi), so with A_j = pi - 2 pi j / 11 (j = 5..1: 0.2856 ... 2.5704)
//     y >= 0:  bin = 5 + #{ j : |phi| >= A_j },      y < 0:  bin = 5 - #{ j : |phi| > A_j }
// and |phi| is on the far side of A_j exactly when c_j = cos A_j |y| - sin A_j x = r sin(|phi| - A_j) is positive (both
// angles in [0, pi]).  Sure means: every |c_j| exceeds 1e-6 (|x| + |y|) >= 1e-6 r — then |phi| is more than 1e-6 rad from
// every A_j, against 1.2e-7 (f1 is the angle rounded to binary32: half an ulp below 4) + 2.3e-7 (T_k against A_j) +
// 1.3e-7 (|x| + |y|) / r (binary32 rounding of c_j: two products, one difference, two rounded constants) — and the
// rounded angle falls into the bin the true one does.  y = +-0 is left to the function (atan2f(-0, x < 0) = -pi rounds
// BELOW -pi: bin 0, not 10), as is anything non-finite or vanishing (NaN compares false; inf - inf is NaN).
__device__ __forceinline__ int spfh_bin_of_angle(float y, float x) {
  const float ya = fabsf(y), s = ya + fabsf(x);
  const float c0 = 0.95949297361449748f * ya - 0.2817325568414295f * x;   // A = 0.2855993321445265
  const float c1 = 0.6548607339452851f * ya - 0.75574957435425827f * x;   // A = 0.8567979964335799
  const float c2 = 0.14231483827328512f * ya - 0.98982144188093268f * x;  // A = 1.4279966607226333
  const float c3 = -0.41541501300188632f * ya - 0.90963199535451844f * x; // A = 1.9991953250116865
  const float c4 = -0.84125353283118109f * ya - 0.54064081745559778f * x; // A = 2.5703939893007397
  const float lim = 1e-6f * s;
  const float low = fminf(fminf(fminf(fabsf(c0), fabsf(c1)), fminf(fabsf(c2), fabsf(c3))), fabsf(c4));
  if (!(low > lim) || !(s > 1e-30f) || !(s < 1e30f) || ya == 0.f) return -1;
  const int m = (int)(c0 > 0.f) + (int)(c1 > 0.f) + (int)(c2 > 0.f) + (int)(c3 > 0.f) + (int)(c4 > 0.f);
  return (y > 0.f) ? 5 + m : 5 - m;
}
// (f[0] is left to the caller: yx = the arc tangent's two arguments, see spfh_bin_of_angle)
__device__ bool dev_pair_features(const float4& p1, const float4& nn1, const float4& p2, const float4& nn2, float* f, float* yx) {
  float dp[3] = {p2.x - p1.x, p2.y - p1.y, p2.z - p1.z};
  const float f4 = sqrtf(dot4_sse(dp, dp));
  if (f4 == 0.0f) return false;
  float n1c[3] = {nn1.x, nn1.y, nn1.z}, n2c[3] = {nn2.x, nn2.y, nn2.z};
  const float angle1 = dot4_sse(n1c, dp) / f4;
  const float angle2 = dot4_sse(n2c, dp) / f4;
  float f3;
  // The role swap asks whether acosf(|angle1|) > acosf(|angle2|) — two binary64 arc cosines (five divisions and four
  // square roots each) for one bit.  acos falls with slope <= -1 on [0, 1]: arguments more than 5e-7 apart give values
  // more than 5e-7 apart, four float ulps of a result below pi/2, so the rounded results are ordered like the arguments
  // are (reversed); equal arguments give equal results.  Only the sliver in between (and arguments above 1 or NaN, where
  // acosf is NaN) evaluates the functions — the answer is the same bit either way.
  if (spfh_swap_roles(angle1, angle2)) {
    n1c[0] = nn2.x;
    n1c[1] = nn2.y;
    n1c[2] = nn2.z;
    n2c[0] = nn1.x;
    n2c[1] = nn1.y;
    n2c[2] = nn1.z;
    dp[0] *= -1.f;
    dp[1] *= -1.f;
    dp[2] *= -1.f;
    f3 = -angle2;
  } else
    f3 = angle1;
  float v[3];
  dev_cross(dp, n1c, v);
  const float v_norm = sqrtf(dot4_sse(v, v));
  if (v_norm == 0.0f) return false;
  v[0] /= v_norm;
  v[1] /= v_norm;
  v[2] /= v_norm;
  float w[3];
  dev_cross(n1c, v, w);
  f[1] = dot4_sse(v, n2c);
  yx[0] = dot4_sse(w, n2c);
  yx[1] = dot4_sse(n1c, n2c);
  f[2] = f3;
  return true;
}

// SPFH of a block of SPFH_PB consecutive points per workgroup.  The work is one Darboux-frame evaluation per (point,
// neighbour) pair — two arc cosines and an arc tangent in binary64-based software — and a point has ~18 neighbours: one
// wavefront per point leaves 70 % of the lanes idle.  The block's pairs are therefore dealt to the 256 threads as one
// flat list (prefix of the neighbour counts in LDS); hits are integer counts per (point, bin) in LDS, so the order in
// which pairs are evaluated cannot matter, and the histogram is rebuilt as `count` additions of hist_incr like the
// reference's accumulation.
#define SPFH_PB 32
__device__ __forceinline__ void d_spfh(const float4* __restrict__ pts, const float4* __restrict__ normals, int n,
                                             const int* __restrict__ nbr_cnt, const NbrLists NL,
                                             float* __restrict__ spfh) {
  __shared__ int cnt[SPFH_PB][33];
  __shared__ int s_off[SPFH_PB + 1], s_k[SPFH_PB];
  __shared__ float4 s_p[SPFH_PB], s_n[SPFH_PB];
  const int tid = threadIdx.x, i0 = blockIdx.x * SPFH_PB;
  if (i0 >= n) return;
  const int np = min(SPFH_PB, n - i0);
  for (int e = tid; e < SPFH_PB * 33; e += 256) (&cnt[0][0])[e] = 0;
  if (tid < 64) {
    const int k = (tid < np) ? max(nbr_cnt[i0 + tid], 0) : 0;
    int tot;
    const int ex = wave_excl_scan_i32(k, &tot);
    if (tid < SPFH_PB) {
      s_off[tid] = ex;
      s_k[tid] = k;
      if (tid < np) {
        s_p[tid] = pts[i0 + tid];
        s_n[tid] = normals[i0 + tid];
      }
    }
    if (tid == 0) s_off[SPFH_PB] = tot;
  }
  __syncthreads();
  const int total = s_off[SPFH_PB];
  const float d_pi = 1.0f / (2.0f * (float)M_PI);
  for (int t = tid; t < total; t += 256) {  // (four pairs per thread and round trip were measured: 20 -> 24 us)
    int pi = 0;  // the point this pair belongs to: largest pi with s_off[pi] <= t (5 halvings of 32)
#pragma unroll
    for (int step = SPFH_PB / 2; step > 0; step >>= 1) pi += (s_off[pi + step] <= t) ? step : 0;
    const int i = i0 + pi;
    const int j = nbr_list_idx(NL, i, s_k[pi])[t - s_off[pi]];
    if (j == i) continue;
    float f[3], yx[2];
    if (!dev_pair_features(s_p[pi], s_n[pi], pts[j], normals[j], f, yx)) continue;
    // (the first block's bin from five cross products instead of the arc tangent: round 2 measured "same kernel time" with
    // the role swap's two arc cosines still in the loop; with those gone — spfh_swap_roles — the arc tangent was a third
    // of the kernel's vector instructions, and the batched path is bound by their count: profiles/r6_pmc_fpfh.json)
    int b1 = spfh_bin_of_angle(yx[0], yx[1]);
    if (b1 < 0) b1 = bin11(11 * (((double)qm_atan2f(yx[0], yx[1]) + M_PI) * (double)d_pi));
    atomicAdd(&cnt[pi][b1], 1);
    atomicAdd(&cnt[pi][11 + bin11(11 * (((double)f[1] + 1.0) * 0.5))], 1);
    atomicAdd(&cnt[pi][22 + bin11(11 * (((double)f[2] + 1.0) * 0.5))], 1);
  }
  __syncthreads();
  for (int e = tid; e < np * 33; e += 256) {
    const int pi = e / 33, b = e - pi * 33;
    const float hist_incr = 100.0f / (float)(s_k[pi] - 1);
    float h = 0.f;
    const int c = cnt[pi][b];
    for (int q = 0; q < c; ++q) h += hist_incr;
    spfh[(size_t)(i0 + pi) * 33 + b] = h;
  }
}

// K4  FPFH weighting (weightPointSPFHSignature).  A workgroup serves FPFH_PB = 7 points: thread (point, bin) adds the
// neighbours' weighted SPFH values in list order (binary32, the reference's accumulation order); the neighbour indices
// and weights of a chunk are staged in LDS by one thread each.
// The three binary64 normalisation sums are, in the reference, one nested (neighbour, bin) loop each — a chain of
// 11 * k dependent additions.  Every term is a non-negative binary32 value; when the largest and the smallest non-zero
// term of a block are at most 17 binary exponents apart, every partial sum of up to 2816 of them is exactly
// representable in binary64 (2816 < 2^12, 24 + 12 + 17 = 53), so ANY summation order gives the reference's bits: the
// threads then keep private binary64 sums and add them up at the end.  Otherwise (not seen on lidar data; covered by a
// test) one thread per block redoes the sum in the reference's order.
// The matcher's preparation of one descriptor (k_desc_prep restated for the end of k2_fpfh): |d|^2 as a binary64 sum rounded
// once, the 64-bit hash, and the row's entry in the duplicate table — slot sequence from the low hash bits, tag = high 32
// bits, value = lowest row with that tag.
// The hash is a SUM of per-component mixes (then one more mix): the 33 threads that hold a descriptor's components in
// k2_fpfh each add theirs to a word in LDS — as a chain over the components (round 2's form) it was 500 dependent
// instructions of one thread per descriptor.
__device__ __forceinline__ u64 desc_mix64(u64 x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
__device__ __forceinline__ u64 desc_hash_term(float v, int k) {
  return desc_mix64(((u64)__float_as_uint(v) | ((u64)(k + 1) << 32)) * 0x9E3779B97F4A7C15ULL);
}
__device__ __forceinline__ u64 desc_hash33(const float* d) {
  u64 h = 0;
  for (int k = 0; k < 33; ++k) h += desc_hash_term(d[k], k);
  return desc_mix64(h);
}
__device__ __forceinline__ void desc_table_insert(u64* __restrict__ table, int mask, u64 h, int i) {
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
      // entries only ever decrease, so a (possibly stale) value that is already <= i makes the atomic redundant
      if ((u32)cur > (u32)i) atomicMin(&table[slot], tag | (u32)i);
      break;
    }
    slot = (slot + 1) & (u32)mask;
  }
}
#define FPFH_PB 7
#define FPFH_CHUNK 32
__device__ __forceinline__ void d_fpfh(const float* __restrict__ spfh, int n, const int* __restrict__ nbr_cnt,
                                             const NbrLists NL, float* __restrict__ fpfh, float* __restrict__ norms,
                                             u64* __restrict__ hashes, u64* __restrict__ table, int mask) {
  // per staged entry: (byte offset of the neighbour's SPFH row, 1 / d^2 — or 0 for an entry the reference skips, d^2 == 0,
  // and for the slots past the end of a list: row 0, weight 0)
  __shared__ __attribute__((aligned(16))) uint2 s_ow[FPFH_PB][FPFH_CHUNK];
  __shared__ int s_k[FPFH_PB];
  __shared__ double s_part[FPFH_PB][33];
  __shared__ float s_vmin[FPFH_PB][33], s_vmax[FPFH_PB][33];
  __shared__ u64 s_hsum[FPFH_PB];
  const int tid = threadIdx.x, i0 = blockIdx.x * FPFH_PB;
  if (i0 >= n) return;
  const int np = min(FPFH_PB, n - i0);
  const int pi = tid / 33, b = tid - pi * 33;      // this thread's (point, bin); tid >= 231: staging only
  const bool owner = pi < np;
  const int lp = tid >> 5, lq = tid & 31;          // staging role: (point, entry of the chunk); 7 x 32 = 224 threads
  if (tid < FPFH_PB) s_k[tid] = (tid < np) ? max(nbr_cnt[i0 + tid], 0) : 0;
  if (tid < FPFH_PB) s_hsum[tid] = 0;  // (the descriptors' hashes, see the end)
  __syncthreads();
  int kmax = 0;
#pragma unroll
  for (int q = 0; q < FPFH_PB; ++q) kmax = max(kmax, s_k[q]);
  const int k = owner ? s_k[pi] : 0;
  // The loop over a list's entries is BRANCH-FREE (round 6: the kernel's SIMDs were 68 % busy with ~20 instructions per
  // value, eight of them the scalar bookkeeping of two guards — profiles/r6_pmc_fpfh.json).  A skipped entry (d^2 == 0)
  // and a slot past the end of the list are staged with weight +0: x * 0 = +0 (an SPFH value is a finite non-negative
  // sum of increments), and adding +0 changes neither the binary32 sum h (never -0) nor the binary64 one — the bits the
  // guarded loop produced.  The extreme terms are tracked on the BIT PATTERNS (non-negative floats order like unsigned
  // integers): the largest as an unsigned maximum, the smallest NON-ZERO one as the minimum of pattern - 1 (+0 wraps to
  // 0xffffffff and never wins); a NaN's pattern lies above infinity's and sends the block to the reference order below.
  float h = 0.f;
  u32 vminb = 0xffffffffu, vmaxb = 0u;
  double part = 0.0;
  const char* __restrict__ spfh_b = (const char*)spfh + 4 * b;  // (uniform base + a 32-bit byte offset per entry)
  for (int t0 = 0; t0 < kmax; t0 += FPFH_CHUNK) {
    __syncthreads();
    if (lp < FPFH_PB) {
      u32 off = 0;
      float w = 0.f;
      if (lp < np && t0 + lq < s_k[lp]) {
        off = (u32)nbr_list_idx(NL, i0 + lp, s_k[lp])[t0 + lq] * 132u;  // (max_voxels < 2^20 rows of 132 bytes)
        const float d2 = nbr_list_d2(NL, i0 + lp, s_k[lp])[t0 + lq];
        w = (d2 == 0) ? 0.f : 1.0f / d2;
      }
      s_ow[lp][lq] = make_uint2(off, __float_as_uint(w));
    }
    __syncthreads();
    if (owner) {
      const int m = min(FPFH_CHUNK, k - t0);
      for (int q0 = 0; q0 < m; q0 += 16) {
        float x[16], w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {  // sixteen independent gathers in flight
          const uint2 ow = s_ow[pi][q0 + u];
          w[u] = __uint_as_float(ow.y);
          x[u] = *(const float*)(spfh_b + ow.x);
        }
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
          const float v0 = x[u] * w[u], v1 = x[u + 1] * w[u + 1];
          h += v0;
          part += (double)v0;
          h += v1;
          part += (double)v1;
          const u32 b0 = __float_as_uint(v0), b1 = __float_as_uint(v1);
          vmaxb = max(max(b0, b1), vmaxb);
          vminb = min(min(b0 - 1u, b1 - 1u), vminb);
        }
      }
    }
  }
  if (owner) {
    s_part[pi][b] = part;
    s_vmin[pi][b] = (vminb == 0xffffffffu) ? INFINITY : __uint_as_float(vminb + 1u);
    s_vmax[pi][b] = __uint_as_float(vmaxb);
  }
  __syncthreads();
  __shared__ float s_out[FPFH_PB][33];
  if (owner) {
  const int blk = b / 11;
  double sum = 0.0;
  float lo = INFINITY, hi = 0.f;
#pragma unroll
  for (int c = 0; c < 11; ++c) {
    sum += s_part[pi][11 * blk + c];
    lo = fminf(lo, s_vmin[pi][11 * blk + c]);
    hi = fmaxf(hi, s_vmax[pi][11 * blk + c]);
  }
  // exponent fields (denormals count as exponent 1: their unit in the last place is that of the smallest normal)
  const int e_hi = (int)((__float_as_uint(hi) >> 23) & 255u), e_lo = max(1, (int)((__float_as_uint(lo) >> 23) & 255u));
  // (11 k terms: their count takes ceil(log2(11 k)) bits off the 53 - 24 a binary64 sum has to spare — 17 for the lists
  // of up to 256 entries the argument above is written for, fewer for the long lists of dense clouds)
  int cnt_bits = 12;
  while ((11 * k) >> cnt_bits) ++cnt_bits;
  const bool exact = !(hi > 0.f) || (hi < INFINITY && e_hi - e_lo <= 29 - cnt_bits);
  if (!exact) {  // the reference's nested order, straight from memory
    sum = 0.0;
    const int i = i0 + pi;
    const int* lidx = nbr_list_idx(NL, i, k);
    const float* ld2 = nbr_list_d2(NL, i, k);
    for (int q = 0; q < k; ++q) {
      const float d2 = ld2[q];
      if (d2 == 0) continue;
      const float w = 1.0f / d2;
      const float* row = spfh + (size_t)lidx[q] * 33 + 11 * blk;
      for (int c = 0; c < 11; ++c) {
        const float val = row[c] * w;
        sum += val;
      }
    }
  }
  if (sum != 0) sum = 100.0 / sum;
  const float out = h * (float)sum;
  fpfh[(size_t)(i0 + pi) * 33 + b] = out;
  s_out[pi][b] = out;
  if (table) atomicAdd(&s_hsum[pi], desc_hash_term(out, b));
  }
  if (!table) return;  // (uniform)
  // the matcher's preparation of the workgroup's descriptors, one thread each: the launch of its own that this used to be
  // (k_desc_prep, 19 us at 16 - 18 k descriptors per cloud) re-read them and paid the table's round trips on the critical
  // path; here they hide behind the other workgroups
  __syncthreads();
  if (tid < np) {
    const float* v = s_out[tid];
    double acc = 0.0;
    for (int k = 0; k < 33; ++k) acc += (double)v[k] * (double)v[k];
    const int i = i0 + tid;
    norms[i] = (float)acc;
    const u64 hh = desc_mix64(s_hsum[tid]);  // (= desc_hash33(v))
    hashes[i] = hh;
    desc_table_insert(table, mask, hh, i);
  }
}

// Matcher::normalizePoints mean (reference src/teaser_utils/feature_matcher.cc:27-36): a plain
// sequential float sum over the cloud.  The additions form one dependent chain per component (float
// addition is not associative, and the oracle defines the sequential order), so three lanes do them —
// but out of LDS: the other waves of the workgroup stream the cloud into a double-buffered SoA tile
// with coalesced 16-byte loads while the chain runs, which removes the HBM/L2 latency from the chain.
#define MEAN_CHUNK 2048
__device__ __forceinline__ void d_seq_mean(const float4* __restrict__ pts, int n, float* __restrict__ mean) {
  __shared__ __attribute__((aligned(16))) float buf[2][3][MEAN_CHUNK];
  const int tid = threadIdx.x;
  const int nchunks = (n + MEAN_CHUNK - 1) / MEAN_CHUNK;
  // loaders: `nld` threads starting at `first` cover one chunk with coalesced float4 loads
  auto load_chunk = [&](int c, int slot, int first, int nld) {
    const int base = c * MEAN_CHUNK;
    for (int t = tid - first; t < MEAN_CHUNK; t += nld) {
      const int i = base + t;
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < n) p = pts[i];
      buf[slot][0][t] = p.x;
      buf[slot][1][t] = p.y;
      buf[slot][2][t] = p.z;
    }
  };
  float m = 0.f;
  if (nchunks > 0) load_chunk(0, 0, 0, 256);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int slot = c & 1;
    if (tid >= 64) {  // waves 1..3 prefetch the next chunk while wave 0 runs the chain
      if (c + 1 < nchunks) load_chunk(c + 1, slot ^ 1, 64, 192);
    } else if (tid < 3) {
      const int cnt = min(MEAN_CHUNK, n - c * MEAN_CHUNK);
      const float* b = buf[slot][tid];
      const float4* b4 = (const float4*)b;
      int t = 0;
      for (; t + 32 <= cnt; t += 32) {  // eight 16-byte LDS reads in flight ahead of 32 dependent additions
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = b4[(t >> 2) + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          m = m + v[q].x;
          m = m + v[q].y;
          m = m + v[q].z;
          m = m + v[q].w;
        }
      }
      for (; t < cnt; ++t) m = m + b[t];
    }
    __syncthreads();
  }
  if (tid < 3) mean[tid] = m / (float)n;
}

// =================================================================================================
// Two-cloud batched launches.  The front end always has two clouds (source, target) of similar size and a
// chain of ~35 small dependent kernels each; one launch serves both (blockIdx.y = cloud), which halves the
// dispatch count on the critical path and doubles the work per dispatch.  A CloudView carries one cloud's
// pointers and sizes; kernels pick theirs with blockIdx.y.
// EXT: view in the kernel arguments (false) or in device memory (true) — a template parameter, see match.hip
#define LAUNCH_CV(kern, a, grid, block, lds, st, ...)                                       \
  do {                                                                                      \
    if ((a).ext)                                                                            \
      hipLaunchKernelGGL((kern<true>), grid, block, lds, st, (ViewExt<CloudView>{(a).ext, {0, 0, 0}}), a, ##__VA_ARGS__);             \
    else                                                                                    \
      hipLaunchKernelGGL((kern<false>), grid, block, lds, st, (ViewExt<CloudView>{nullptr, {0, 0, 0}}), a, ##__VA_ARGS__);            \
  } while (0)
__device__ __forceinline__ const u64* keys_src(const CloudView& C, int src) { return src == 0 ? C.keys_a : C.keys_b; }
__device__ __forceinline__ u64* keys_dst(const CloudView& C, int src) { return src == 0 ? C.keys_b : C.keys_a; }

template <bool EXT>
__global__ __launch_bounds__(256) void k2_minmax(ViewExt<CloudView> x, Clouds2 a, int use_vox, int zero_counts) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  if (zero_counts && blockIdx.x == 0 && threadIdx.x < 16) C.counts[threadIdx.x] = 0;  // first launch of a voxel stage
  d_minmax(use_vox ? C.vox : C.raw, use_vox ? C.n : C.P, C.mm_part);
}
// first launch of a sort: the histogram of the lowest digit of the keys in keys_a — which, with make_keys, are made here
// too: those of the voxel grid (use_vox = 0: raw points, cell side `side` = leaf) or of the neighbour-search grid
// (use_vox = 1: voxel centroids)
// mm_parts: records k2_minmax left for this cloud (0: C.mm already holds the box / origin to use)
template <bool EXT>
__global__ __launch_bounds__(256) void k2_keys_hist(ViewExt<CloudView> x, Clouds2 a, int use_vox, int make_keys, float side,
                                                    int mm_parts, float cell_side /* voxel stage: the FPFH chain's cell, or 0 */) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  const int n = use_vox ? C.n : C.P;
  const int nblk = (n + RADIX_TILE - 1) / RADIX_TILE;
  __shared__ u32 s_mm[6];
  if (make_keys) {
    if (mm_parts > 0) {
      mm_fold(C.mm_part, mm_parts, s_mm);
      if (blockIdx.x == 0 && threadIdx.x < 6) C.mm[threadIdx.x] = s_mm[threadIdx.x];  // for the kernels further down the chain
    } else {
      if (threadIdx.x < 6) s_mm[threadIdx.x] = C.mm[threadIdx.x];
      __syncthreads();
    }
  }
  if (!make_keys) {
    const u64* __restrict__ keys = C.keys_a;
    d_radix_hist<8, RADIX_TILE>([&](int i) { return keys[i]; }, (u64*)nullptr, n, 32, C.hist, nblk);
  } else if (use_vox) {
    const CellGrid g = cell_grid(s_mm, side);
    const float4* __restrict__ pts = C.vox;
    d_radix_hist<8, RADIX_TILE>([&](int i) { return cell_sort_key(g, pts[i], i); }, C.keys_a, n, 32, C.hist, nblk);
  } else {
    const VoxGrid g = vox_grid(s_mm, side);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      vox_grid_counts(g, C.counts);
      if (cell_side > 0.f) {  // cells of the neighbour-search grid over this box: the host picks the FPFH chain's form by it
        const CellDims cd = cell_dims(s_mm, cell_side);
        C.counts[CNT_NCELL] = cd.nx * cd.ny * cd.nz;  // (<= 2^24: 256 cells per axis)
      }
    }
    if (threadIdx.x < RADIX_TILE / VOX_TILE_SINGLE)  // k2_vox_centroids' look-back words ("not published yet"), one per tile
      C.vox_look[blockIdx.x * (RADIX_TILE / VOX_TILE_SINGLE) + threadIdx.x] = 0;
    const float4* __restrict__ pts = C.raw;
    d_radix_hist<8, RADIX_TILE>([&](int i) { return vox_key(g, pts[i], i); }, C.keys_a, n, 32, C.hist, nblk);
  }
}
// pass: 0-based; last: no pass follows; adaptive: the keys carry C.counts[CNT_SORT_BITS] significant bits (known on the
// device only) and a pass whose digit lies wholly above them is skipped — the consumers pick the buffer with
// sorted_src() below
template <bool EXT>
__global__ __launch_bounds__(256) void k2_radix_scatter(ViewExt<CloudView> x, Clouds2 a, int use_vox, int pass, int last,
                                                      int adaptive) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  const int n = use_vox ? C.n : C.P;
  const int nblk = (n + RADIX_TILE - 1) / RADIX_TILE;
  if (adaptive && 8 * pass >= C.counts[CNT_SORT_BITS] && pass > 0) return;
  const bool final_pass = last || (adaptive && 8 * (pass + 1) >= C.counts[CNT_SORT_BITS]);
  const int src = pass & 1;
  u32* H = C.hist;
  const size_t hs = (size_t)nblk * 256;
  d_radix_scatter<8, RADIX_TILE>(keys_src(C, src), keys_dst(C, src), n, 32 + 8 * pass, H + (size_t)(pass % 3) * hs, nblk,
                                 final_pass ? nullptr : H + (size_t)((pass + 1) % 3) * hs,
                                 final_pass ? nullptr : H + (size_t)((pass + 2) % 3) * hs);
}
// which of keys_a (0) / keys_b (1) holds the sorted keys after `passes` passes of which the adaptive ones may have been
// skipped
__device__ __forceinline__ int sorted_src(const CloudView& C, int passes, int adaptive) {
  const int done = adaptive ? max(1, min(passes, (C.counts[CNT_SORT_BITS] + 7) >> 3)) : passes;
  return done & 1;
}
template <bool EXT>
__global__ __launch_bounds__(256) void k2_vox_centroids(ViewExt<CloudView> x, Clouds2 a, int cap, int src) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  if (src < 0) src = sorted_src(C, -src, 1);
  constexpr int TILE = EXT ? VOX_TILE_BATCH : VOX_TILE_SINGLE;
  d_vox_centroids<TILE>(keys_src(C, src), C.raw, C.P, C.vox_look, C.vox, cap, (C.P + TILE - 1) / TILE, C.counts, C.mail,
                        C.mail_seq_slot, C.seq);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k2_ranges(ViewExt<CloudView> x, Clouds2 a, float cell, int src) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  const u64* sorted = keys_src(C, src);
  {  // the first n threads of the launch also gather the points into cell-sorted order (was k2_sorted_points)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < C.n) {
      const u32 j = (u32)sorted[t];
      float4 p = C.vox[j];
      p.w = __uint_as_float(j);
      C.spts[t] = p;
    }
  }
  d_ranges(C.vox, C.n, sorted, C.mm, cell, C.ranges);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k2_cell_count(ViewExt<CloudView> x, Clouds2 a, float cell) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  d_cell_count(C.vox, C.n, C.mm, cell, C.cell_cnt, C.keys_a, (int*)C.hist);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k2_cell_scan(ViewExt<CloudView> x, Clouds2 a, float cell) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  d_cell_scan(C.mm, cell, C.cell_cnt, C.cell_start, (int*)C.hist, C.counts);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k2_cell_place(ViewExt<CloudView> x, Clouds2 a, float cell) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  d_cell_place(C.vox, C.n, C.mm, cell, C.cell_start, C.keys_a, C.spts, C.ranges);
}
template <bool EXT>
__global__ __launch_bounds__(64) void k2_neighbors(ViewExt<CloudView> x, Clouds2 a, float r2) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  d_neighbors(C.vox, C.n, C.spts, C.ranges, r2, C.nbr_cnt, C.nbr_idx, C.nbr_d2, C.counts);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k2_neighbors_big(ViewExt<CloudView> x, Clouds2 a, float r2) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  d_neighbors_big(C.vox, C.n, C.spts, C.ranges, r2, C.nbr_cnt, C.nbr_idx, C.nbr_big_idx, C.nbr_big_d2, C.nbr_big_cap,
                  C.counts);
}
template <bool EXT>
__global__ __launch_bounds__(1024) void k2_nbr_scan(ViewExt<CloudView> x, Clouds2 a) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  d_scan_i32_copy(C.nbr_cnt, C.nbr_off, C.n);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k2_normals(ViewExt<CloudView> x, Clouds2 a, float rn2) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  // (the clean slate of the duplicate table k2_fpfh fills two launches on, when it does the matcher's preparation)
  if (C.dd_table)
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e <= C.dd_mask; e += gridDim.x * blockDim.x) C.dd_table[e] = ~0ULL;
  d_normals(C.vox, C.n, C.nbr_cnt, NbrLists{C.nbr_idx, C.nbr_d2, C.nbr_big_idx, C.nbr_big_d2}, rn2, C.normals);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k2_spfh(ViewExt<CloudView> x, Clouds2 a) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  d_spfh(C.vox, C.normals, C.n, C.nbr_cnt, NbrLists{C.nbr_idx, C.nbr_d2, C.nbr_big_idx, C.nbr_big_d2}, C.spfh);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k2_fpfh(ViewExt<CloudView> x, Clouds2 a) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  d_fpfh(C.spfh, C.n, C.nbr_cnt, NbrLists{C.nbr_idx, C.nbr_d2, C.nbr_big_idx, C.nbr_big_d2}, C.fpfh, C.norms, C.dd_hash,
         C.dd_table, C.dd_mask);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k2_seq_mean(ViewExt<CloudView> x, Clouds2 a, int cap) {
  const CloudView& C = EXT ? x.ext[blockIdx.y] : a.c[blockIdx.y];  // (inline on purpose: see ViewExt)
  int n = C.n;
  if (n < 0) {
    // enqueued behind the voxel stage BEFORE the host has its counters (the whole-path driver: the 70 us chain starts while
    // the host still waits for the mail instead of behind the FPFH chain's launches): the count is the device's own.  A cloud
    // that passes through is copied by the host after the mail; its mean is enqueued again behind that copy.
    if (C.counts[CNT_VOX_OVERFLOW] != 0) return;
    n = min(max(C.counts[CNT_NVOX], 0), cap);
  }
  d_seq_mean(C.vox, n, C.mean);
}
__global__ __launch_bounds__(1024) void k_scan_i32_copy(const int* __restrict__ in, int* __restrict__ out, int n) {
  d_scan_i32_copy(in, out, n);
}

hipError_t exclusive_scan_i32(const int* in, int* out, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_scan_i32_copy, dim3(1), dim3(1024), 0, st, in, out, n);
  return hipGetLastError();
}

static CloudView make_view(CloudBufs& C, const float4* raw, int P, int n, int* mail, int* mail_seq_slot, int seq) {
  CloudView v;
  memset(&v, 0, sizeof(v));
  v.raw = raw;
  v.P = P;
  v.n = n;
  v.counts = C.counts;
  v.mm = C.mm;
  v.mm_part = C.mm_part;
  v.vox = C.vox;
  v.normals = C.normals;
  v.spfh = C.spfh;
  v.fpfh = C.fpfh;
  v.keys_a = C.keys_a;
  v.keys_b = C.keys_b;
  v.hist = C.hist;
  v.mail = mail;
  v.mail_seq_slot = mail_seq_slot;
  v.seq = seq;
  v.blkcnt = (int*)C.hist;
  v.vox_look = C.vox_look;
  v.blkoff = v.blkcnt + (P + 1023) / 1024 + 8;
  v.nbr_cnt = C.nbr_cnt;
  v.nbr_off = C.nbr_off;
  v.nbr_idx = C.nbr_idx;
  v.nbr_d2 = C.nbr_d2;
  v.nbr_big_idx = C.nbr_big_idx;
  v.nbr_big_d2 = C.nbr_big_d2;
  v.nbr_big_cap = C.nbr_big_cap;
  v.spts = C.spts;
  v.ranges = C.ranges;
  v.cell_cnt = C.cell_cnt;
  v.cell_start = C.cell_start;
  v.mean = C.mean;
  return v;
}

// The clouds one launch chain works on: up to two ride in the kernel arguments, more go through the stage.
struct CloudSet {
  Clouds2 a;
  int nc = 0;
  int maxP = 1, maxn = 1;
};
static hipError_t cloudset_finish(CloudSet& S, const CloudView* views, int nc, ViewStage* stage, hipStream_t st) {
  S.nc = nc;
  S.a.ext = nullptr;
  S.maxP = S.maxn = 1;
  for (int c = 0; c < nc; ++c) {
    if (views[c].P > S.maxP) S.maxP = views[c].P;
    if (views[c].n > S.maxn) S.maxn = views[c].n;
  }
  if (nc <= 2) {
    S.a.c[0] = views[0];
    S.a.c[1] = views[nc > 1 ? 1 : 0];
    return hipSuccess;
  }
  S.a.c[0] = S.a.c[1] = views[0];
  S.a.ext = (const CloudView*)stage_push(stage, views, sizeof(CloudView) * (size_t)nc, st);
  return S.a.ext ? hipSuccess : hipErrorOutOfMemory;
}

// stable LSD radix sort of keys_a by bits [32, 32+key_bits); returns which buffer holds the result (0: keys_a)
// adaptive: the keys' significant bits are in counts[CNT_SORT_BITS]; passes above them return at once and the result's
// buffer is only known on the device (the return value is then minus the number of passes launched: consumers call
// sorted_src())
static int radix_sort2(const CloudSet& S, int use_vox, int key_bits, hipStream_t st, bool adaptive = false, bool make_keys = false,
                       float side = 0.f, int mm_parts = 0, float cell_side = 0.f) {
  const int maxblk = ((use_vox ? S.maxn : S.maxP) + RADIX_TILE - 1) / RADIX_TILE;
  const int passes = key_bits / 8;
  // one launch makes the keys and the first pass's histogram; every scatter accumulates the next pass's histogram (see d_radix_hist).
  // A single-launch pass (tiles exchanging offsets through flags) needs device-scope fences, which on this multi-XCD
  // part cost more than the launch boundary.
  LAUNCH_CV(k2_keys_hist, S.a, dim3(max(maxblk, 1), S.nc), dim3(256), 0, st, use_vox, make_keys ? 1 : 0, side, mm_parts, cell_side);
  for (int p = 0; p < passes; ++p)
    LAUNCH_CV(k2_radix_scatter, S.a, dim3(maxblk, S.nc), dim3(256), 0, st, use_vox, p, p + 1 == passes ? 1 : 0, adaptive ? 1 : 0);
  return adaptive ? -passes : (passes & 1);
}

// passes: radix passes to launch (4 covers every grid pcl::VoxelGrid accepts; a grid below 2^24 cells needs 3 — see
// voxelize_enqueue)
static void voxelize_launch(const CloudSet& S, float leaf, int max_voxels, hipStream_t st, int passes = 4, float cell_side = 0.f) {
  const int nc = S.nc;
  const int g = min(1024, (S.maxP + 255) / 256);
  const int mm_parts = max(1, min(g, MM_MAX_PARTS));
  LAUNCH_CV(k2_minmax, S.a, dim3(mm_parts, nc), dim3(256), 0, st, 0, 1);
  const int where = radix_sort2(S, 0, 8 * passes, st, true, true, leaf, mm_parts, cell_side);  // (a pass above the keys' bits returns at once)
  const int tile = S.a.ext ? VOX_TILE_BATCH : VOX_TILE_SINGLE;  // (as the kernel's template argument picks it)
  const int nblk = (S.maxP + tile - 1) / tile;
  LAUNCH_CV(k2_vox_centroids, S.a, dim3(nblk, nc), dim3(256), 0, st, max_voxels, where);
}

// voxel-grid down-sampling of nc (1 or 2) raw clouds
hipError_t voxelize_enqueue(FrontBufs& F, int nc, const float4* const* raw, const int* P, float leaf, hipStream_t st,
                            int passes, float cell_side) {
  (void)hipGetLastError();
  CloudView v[2];
  for (int c = 0; c < nc; ++c) {
    const int ci = (nc > 1 && c == 1) ? 1 : 0;
    v[c] = make_view(F.cloud[ci], raw[c], P[c], 0, F.mail ? F.mail + (ci ? MAIL_VOX1 : MAIL_VOX0) : nullptr,
                     F.mail ? F.mail + (ci ? MAIL_SEQ_VOX1 : MAIL_SEQ_VOX0) : nullptr, F.mail_seq);
  }
  CloudSet S;
  hipError_t e = cloudset_finish(S, v, nc, nullptr, st);
  if (e != hipSuccess) return e;
  voxelize_launch(S, leaf, F.max_voxels, st, min(4, max(1, passes)), cell_side);
  return hipGetLastError();
}

hipError_t voxelize_enqueue_group(FrontBufs* const* F, int G, const float4* const* raw, const int* P, float leaf,
                                  ViewStage* stage, hipStream_t st, float cell_side) {
  (void)hipGetLastError();
  std::vector<CloudView> v((size_t)2 * G);
  for (int g = 0; g < G; ++g)
    for (int c = 0; c < 2; ++c) {
      FrontBufs& Fg = *F[g];
      v[2 * g + c] = make_view(Fg.cloud[c], raw[2 * g + c], P[2 * g + c], 0, Fg.mail ? Fg.mail + (c ? MAIL_VOX1 : MAIL_VOX0) : nullptr,
                               Fg.mail ? Fg.mail + (c ? MAIL_SEQ_VOX1 : MAIL_SEQ_VOX0) : nullptr, Fg.mail_seq);
    }
  CloudSet S;
  hipError_t e = cloudset_finish(S, v.data(), 2 * G, stage, st);
  if (e != hipSuccess) return e;
  voxelize_launch(S, leaf, F[0]->max_voxels, st, 4, cell_side);
  return hipGetLastError();
}

hipError_t set_count_enqueue(CloudBufs& C, int which, int value, hipStream_t st) {
  hipLaunchKernelGGL(k_set_count, dim3(1), dim3(64), 0, st, C.counts, which, value);
  return hipGetLastError();
}

// Matcher::normalizePoints means of nc clouds; independent of the FPFH chain, so the whole-path driver runs it
// on the slot's second stream
// n[c] < 0: the cloud's voxel count is read on the device (at most `cap`; see k2_seq_mean)
hipError_t mean_enqueue(FrontBufs& F, int first, int nc, const int* n, hipStream_t st, int cap) {
  CloudView v[2];
  for (int c = 0; c < nc; ++c) v[c] = make_view(F.cloud[first + c], nullptr, 0, n[c], nullptr, nullptr, 0);
  CloudSet S;
  hipError_t e = cloudset_finish(S, v, nc, nullptr, st);
  if (e != hipSuccess) return e;
  LAUNCH_CV(k2_seq_mean, S.a, dim3(1, nc), dim3(256), 0, st, cap);
  return hipGetLastError();
}
hipError_t mean_enqueue_group(FrontBufs* const* F, int G, const int* n, ViewStage* stage, hipStream_t st) {
  std::vector<CloudView> v((size_t)2 * G);
  for (int g = 0; g < G; ++g)
    for (int c = 0; c < 2; ++c) v[2 * g + c] = make_view(F[g]->cloud[c], nullptr, 0, n[2 * g + c], nullptr, nullptr, 0);
  CloudSet S;
  hipError_t e = cloudset_finish(S, v.data(), 2 * G, stage, st);
  if (e != hipSuccess) return e;
  LAUNCH_CV(k2_seq_mean, S.a, dim3(1, 2 * G), dim3(256), 0, st, 0);
  return hipGetLastError();
}

// normals + SPFH + FPFH (+ the matcher's sequential mean) of the clouds of S
// origin_known: C.mm[0..2] already hold a lower bound of the points (the raw cloud's minimum, left there by the voxel
// stage whose centroids these are) — the neighbour grid only needs an origin at or below every point (cell_of clamps, so
// a centroid that rounds an ulp below it is still in cell 0), which saves two launches of a latency-bound chain
static void fpfh_launch(const CloudSet& S, float r_normal, float r_fpfh, hipStream_t st, bool with_mean, bool origin_known,
                        bool long_lists, int max_ncell = 0 /* > 0: every cloud's grid has at most that many cells: the dense cell table */) {
  const int nc = S.nc, maxn = S.maxn;
  const int g = min(1024, (maxn + 255) / 256);
  const float cell = r_fpfh * 1.001f;
  const float r2 = (float)((double)r_fpfh * (double)r_fpfh);
  const float rn2 = (float)((double)r_normal * (double)r_normal);
  const int mm_parts = origin_known ? 0 : max(1, min(g, MM_MAX_PARTS));
  if (!origin_known) LAUNCH_CV(k2_minmax, S.a, dim3(mm_parts, nc), dim3(256), 0, st, 1, 0);  // (keeps the counters)
  if (origin_known && max_ncell > 0 && max_ncell <= QTR_CELL_CAP) {
    // the dense cell table (see d_cell_count): three short launches instead of a key launch, three radix passes and the
    // binary searches
    LAUNCH_CV(k2_cell_count, S.a, dim3((maxn + 255) / 256, nc), dim3(256), 0, st, cell);
    LAUNCH_CV(k2_cell_scan, S.a, dim3((max_ncell + CELL_SCAN_TILE) / CELL_SCAN_TILE, nc), dim3(256), 0, st, cell);
    LAUNCH_CV(k2_cell_place, S.a, dim3((9 * maxn + 255) / 256, nc), dim3(256), 0, st, cell);
  } else {
    const int where = radix_sort2(S, 1, 24, st, false, true, cell, mm_parts);
    // k2_ranges also gathers the points into cell-sorted order (its first n threads): one launch fewer
    LAUNCH_CV(k2_ranges, S.a, dim3((9 * maxn + 255) / 256, nc), dim3(256), 0, st, cell, where);
  }
  // (fusing the normals into k2_neighbors was tried: the eigen-solve then runs once per WAVE instead of once per
  // thread and the launch went from 21 + 14 us to 51 us)
  LAUNCH_CV(k2_neighbors, S.a, dim3(maxn, nc), dim3(64), 0, st, r2);
  // lists of more than QTR_KMAX entries (one workgroup per such point; returns at once when the cloud has none)
  if (long_lists) LAUNCH_CV(k2_neighbors_big, S.a, dim3(maxn, nc), dim3(256), (size_t)NBIG_LDS_KEYS * 8, st, r2);
  {
    // a thread per point: one pair at a time 256-thread workgroups would fill 140 of the 256 compute units (section 17 of
    // profiles/r6_ab.txt); the batched path has points enough
    const int bs = S.a.ext ? 256 : 64;
    LAUNCH_CV(k2_normals, S.a, dim3((maxn + bs - 1) / bs, nc), dim3(bs), 0, st, rn2);
  }
  LAUNCH_CV(k2_spfh, S.a, dim3((maxn + SPFH_PB - 1) / SPFH_PB, nc), dim3(256), 0, st);
  LAUNCH_CV(k2_fpfh, S.a, dim3((maxn + FPFH_PB - 1) / FPFH_PB, nc), dim3(256), 0, st);
  if (with_mean) LAUNCH_CV(k2_seq_mean, S.a, dim3(1, nc), dim3(256), 0, st, 0);
}

// long_lists: also launch k2_neighbors_big, which serves the points with more than QTR_KMAX neighbours inside r_fpfh.
// Voxel-grid centroids at the demo's leaf never have that many, so the whole-path drivers leave it out (one launch
// fewer on the chain) and check CNT_NBR_OVERFLOW afterwards: if it is set the stage is run again with long_lists.
static void view_desc_prep(CloudView& v, CloudBufs& C, int dd_slots) {
  v.norms = C.norms;
  v.dd_hash = C.dd_hash;
  v.dd_table = C.dd_table;
  v.dd_mask = dd_slots - 1;
}
hipError_t fpfh_enqueue(FrontBufs& F, int first, int nc, const int* n, float r_normal, float r_fpfh, hipStream_t st,
                        bool with_mean, bool origin_known, bool long_lists, bool desc_prep, int max_ncell) {
  (void)hipGetLastError();
  CloudView v[2];
  for (int c = 0; c < nc; ++c) {
    v[c] = make_view(F.cloud[first + c], nullptr, 0, n[c], nullptr, nullptr, 0);
    if (desc_prep) view_desc_prep(v[c], F.cloud[first + c], F.dd_slots);
  }
  CloudSet S;
  hipError_t e = cloudset_finish(S, v, nc, nullptr, st);
  if (e != hipSuccess) return e;
  fpfh_launch(S, r_normal, r_fpfh, st, with_mean, origin_known, long_lists, max_ncell);
  return hipGetLastError();
}
hipError_t fpfh_enqueue_group(FrontBufs* const* F, int G, const int* n, float r_normal, float r_fpfh, ViewStage* stage,
                              hipStream_t st, bool long_lists, bool desc_prep, int max_ncell) {
  (void)hipGetLastError();
  std::vector<CloudView> v((size_t)2 * G);
  for (int g = 0; g < G; ++g)
    for (int c = 0; c < 2; ++c) {
      v[2 * g + c] = make_view(F[g]->cloud[c], nullptr, 0, n[2 * g + c], nullptr, nullptr, 0);
      if (desc_prep) view_desc_prep(v[2 * g + c], F[g]->cloud[c], F[g]->dd_slots);
    }
  CloudSet S;
  hipError_t e = cloudset_finish(S, v.data(), 2 * G, stage, st);
  if (e != hipSuccess) return e;
  fpfh_launch(S, r_normal, r_fpfh, st, false, true, long_lists, max_ncell);  // always behind voxelize_enqueue_group
  return hipGetLastError();
}

// =================================================================================================
static int dedup_slots(int max_voxels) {
  int m = 1024;
  while (m < 2 * max_voxels) m <<= 1;
  return m;
}
size_t frontend_scratch_bytes(int max_points, int max_voxels) {
  size_t per_cloud = 0;
  per_cloud += 4096 + MM_MAX_PARTS * 32 + 256;               // counts, mm, mean, mm_part
  per_cloud += (size_t)max_voxels * (16 + 16 + 132 + 132);   // vox, normals, spfh, fpfh
  per_cloud += 2 * (size_t)max_points * 8;                   // keys
  per_cloud += (size_t)(4096 * ((max_points + RADIX_TILE - 1) / RADIX_TILE) + 8192) * 4 + 65536;  // hist
  per_cloud += (size_t)((max_points + RADIX_TILE - 1) / RADIX_TILE + 64) * 4 * (RADIX_TILE / VOX_TILE_SINGLE) + 256;  // vox_look
  per_cloud += (size_t)max_voxels * 4 * 2 + 64;              // nbr_cnt, nbr_off
  per_cloud += (size_t)max_voxels * QTR_KMAX * 8;            // nbr_idx, nbr_d2
  per_cloud += (size_t)max_voxels * (16 + 72) + 512;         // spts, ranges
  per_cloud += 2 * (size_t)(QTR_CELL_CAP + 4096) * 4 + 512;  // cell_cnt, cell_start
  per_cloud += (size_t)max_points * 16 + 256;                // raw_sorted
  size_t shared = (size_t)max_voxels * 64 + 16384 + 2 * TAIL_MAXWG * 4;
  shared += (size_t)max_voxels * (4 + 4 + 4 + 8) + (size_t)max_voxels * 4 * 6 + 4096;  // doubled pair lists + nc_* (cross-check off)
  const size_t vpad = ((size_t)max_voxels + 511) / 512 * 512;
  per_cloud += 3 * 34 * vpad * 4 + (size_t)max_voxels * 8 + 4096;  // baseT, queryT, baseTb, norms, nb_row, nb_start, max_norm
  per_cloud += 2 * (vpad * 224 + 2 * 7168) + 512;                   // baseH, queryH (+ two tiles the prefetch may touch)
  per_cloud += (size_t)max_voxels * 8 + (size_t)dedup_slots(max_voxels) * 8 + 512;  // dd_hash, dd_table
  shared += vpad * 32 * 16 + 4 * ((size_t)max_voxels * 4 + 1024);     // nn_partial, recheck_rows, recheck_thr, recheck_span
  shared += (size_t)max_voxels * 4 + 34 * vpad * 4 + vpad * 4 + 1024;  // hit_rows, queryT_c, norms_c
  shared += vpad * 224 + 2 * 7168 + 256 + (size_t)max_voxels * 4 + 1024;  // queryH_c, recheck_q
  return 2 * per_cloud + shared + 64 * 256;
}

void frontend_carve(FrontBufs& F, void* base, int max_points, int max_voxels) {
  char* p = (char*)base;
  auto take = [&](size_t bytes) {
    char* r = p;
    p += (bytes + 255) & ~(size_t)255;
    return (void*)r;
  };
  F.max_points = max_points;
  F.max_voxels = max_voxels;
  for (int c = 0; c < 2; ++c) {
    CloudBufs& C = F.cloud[c];
    C.counts = (int*)take(16 * 4);
    C.mm = (u32*)take(8 * 4);
    C.mm_part = (u32*)take(MM_MAX_PARTS * 8 * 4);
    C.mean = (float*)take(4 * 4);
    C.vox = (float4*)take((size_t)max_voxels * 16);
    C.normals = (float4*)take((size_t)max_voxels * 16);
    C.spfh = (float*)take((size_t)max_voxels * 132);
    C.fpfh = (float*)take((size_t)max_voxels * 132);
    C.keys_a = (u64*)take((size_t)max_points * 8);
    C.keys_b = (u64*)take((size_t)max_points * 8);
    C.hist = (u32*)take((size_t)(4096 * ((max_points + RADIX_TILE - 1) / RADIX_TILE) + 8192) * 4);
    C.vox_look = (int*)take((size_t)((max_points + RADIX_TILE - 1) / RADIX_TILE + 64) * 4 * (RADIX_TILE / VOX_TILE_SINGLE));
    C.nbr_cnt = (int*)take((size_t)max_voxels * 4);
    C.nbr_off = (int*)take((size_t)(max_voxels + 1) * 4);
    C.nbr_idx = (int*)take((size_t)max_voxels * QTR_KMAX * 4);
    C.nbr_d2 = (float*)take((size_t)max_voxels * QTR_KMAX * 4);
    C.spts = (float4*)take((size_t)max_voxels * 16);
    C.raw_sorted = (float4*)take((size_t)max_points * 16);
    C.ranges = (int*)take((size_t)max_voxels * 18 * 4);
    C.cell_cnt = (int*)take((size_t)(QTR_CELL_CAP + 4096) * 4);   // (zeroed once by the handle: create_impl)
    C.cell_start = (int*)take((size_t)(QTR_CELL_CAP + 4096) * 4);
    const size_t vpad = ((size_t)max_voxels + 511) / 512 * 512;
    C.baseT = (float*)take(34 * vpad * 4);
    C.queryT = (float*)take(34 * vpad * 4);
    C.baseTb = (float*)take(34 * vpad * 4);
    C.baseH = (uint4*)take(vpad * 224 + 2 * 7168);  // two tiles of slack: the core prefetches past its slice
    C.queryH = (uint4*)take(vpad * 224 + 2 * 7168);
    C.nb_row = (int*)take((size_t)max_voxels * 4);
    C.nb_start = (int*)take((NORM_BINS + 2) * 4);
    C.norms = (float*)take((size_t)max_voxels * 4);
    C.max_norm = (u32*)take(64);
    C.dd_hash = (u64*)take((size_t)max_voxels * 8);
    C.dd_table = (u64*)take((size_t)dedup_slots(max_voxels) * 8);
  }
  F.dd_slots = dedup_slots(max_voxels);
  F.best_small = (u64*)take((size_t)max_voxels * 8);
  F.best_large = (u64*)take((size_t)max_voxels * 8);
  F.nn_of_small = (int*)take((size_t)max_voxels * 4);
  F.nn_of_large = (int*)take((size_t)max_voxels * 4);
  F.cross_i = (int*)take((size_t)max_voxels * 8);  // 2 x max_voxels: without the cross-check the list is corres_ij + corres_ji
  F.cross_j = (int*)take((size_t)max_voxels * 8);
  F.flags = (int*)take((size_t)max_voxels * 4);
  F.scan = (int*)take((size_t)max(max_voxels + 1, 2 * TAIL_MAXWG) * 4);  // (also the tail kernels' look-back words)
  F.passed = (int*)take((size_t)max_voxels * 8);
  F.tgt_of_src = (int*)take((size_t)max_voxels * 4);
  F.corr = (int*)take((size_t)max_voxels * 16);
  F.nc_cnt = (int*)take((size_t)max_voxels * 4);
  F.nc_fill = (int*)take((size_t)max_voxels * 4);
  F.nc_off = (int*)take((size_t)(max_voxels + 1) * 4);
  F.nc_list = (int*)take((size_t)max_voxels * 8);
  F.mcounts = (int*)take(16 * 4);
  F.nn_partial = take((((size_t)max_voxels + 511) / 512 * 512) * 32 * 16);
  F.recheck_rows = (int*)take((size_t)max_voxels * 4);
  F.recheck_thr = (float*)take((size_t)max_voxels * 4);
  F.recheck_span = (int2*)take((size_t)max_voxels * 8);
  F.hit_rows = (int*)take((size_t)max_voxels * 4);
  F.queryT_c = (float*)take(34 * (((size_t)max_voxels + 511) / 512 * 512) * 4);
  F.norms_c = (float*)take((((size_t)max_voxels + 511) / 512 * 512) * 4);
  F.queryH_c = (uint4*)take((((size_t)max_voxels + 511) / 512 * 512) * 224 + 2 * 7168);
  F.recheck_q = (int*)take((size_t)max_voxels * 4);
  {
    const char* e = QTR_ENGINE_ENV("QTR_NN_ENGINE");
    F.nn_engine = (e && strcmp(e, "exact") == 0) ? 0 : (e && strcmp(e, "mfma32") == 0) ? 1 : 2;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      F.n_cu = cus;
  }
}

hipError_t match_init_attributes();
hipError_t frontend_init_attributes() {
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void*)k2_neighbors_big<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               NBIG_LDS_KEYS * 8)) != hipSuccess)
    return e;
  if ((e = hipFuncSetAttribute((const void*)k2_neighbors_big<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               NBIG_LDS_KEYS * 8)) != hipSuccess)
    return e;
  return match_init_attributes();
}

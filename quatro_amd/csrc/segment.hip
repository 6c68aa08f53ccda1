// segment.hip — "next" row (f)1 of SURVEY.md section 8: range-image projection + sub-cluster rejection, the stage
// right before voxelisation in the reference demo (ImageProjection::segmentCloud in "Patchwork" mode,
// include/imageProjection.hpp:273-294; projectPointCloud :308-352, maskGround :354-364, cloudSegmentation :424-483,
// labelComponents :485-581).  The reference labels components with a sequential breadth-first search over the
// 64 x 1800 image; here:
//   K1 project      every point -> pixel, atomicMax of the point index (the sequential loop's "last writer wins")
//   K2 init         pixel range from its owner point; parent[p] = p
//   K3 merge        union-find over the symmetric angle criterion (each undirected edge once, roots = smallest
//                   pixel index = the BFS's seed: the first pixel of the component in row-major order)
//   K4 flatten      root per pixel, component sizes, row masks of the non-seed pixels (atomics aggregated per wave)
//   K5 classify     the BFS's validity rule -> valid / outlier / valid-root flags + per-block counts
//   K6 blockscan    exclusive scan of the (<= 282) block counts
//   K7 rootrank     label of every valid component = 1 + number of valid roots before it (row-major)
//   K8 compact      valid segments (x, y, z, label) and outliers (x, y, z, row + col / 1e4) in row-major order
// All integer outputs are bit-identical to the oracle's BFS restatement by construction (connected components are
// unique); the float predicate uses the shared qtr_math.h functions.
#include <cfloat>

#include "common.h"
#include "../../include/qtr_math.h"

struct IpDev {  // qtr_ip_params + derived constants
  int n_scan, horizon_scan;
  float ang_res_x, ang_res_y, ang_bottom;
  int neighbor_mode, num_min_pts;
  float segment_theta;
  int valid_point_num, valid_line_num;
  float sx, cx, sy, cy;  // sin / cos of segmentAlphaX / segmentAlphaY
};

struct SegBufs {
  int np_cap = 0;
  int* owner = nullptr;      // [NP] point index or -1
  float* range = nullptr;    // [NP]
  int* parent = nullptr;     // [NP] union-find parents, then roots
  int* cnt = nullptr;        // [NP] component size at the root
  u64* rowmask = nullptr;    // [NP] rows holding a non-seed pixel, at the root
  int* cls = nullptr;        // [NP] 0 none, 1 valid, 2 outlier; bit 2: valid root
  int* rootrank = nullptr;   // [NP] label - 1 at valid roots
  int* blk = nullptr;        // [3][nblk + 1] block counts -> block offsets (valid, outlier, valid root)
  float4* out_valid = nullptr;   // [NP]
  float4* out_outl = nullptr;    // [NP]
  int* labelmat = nullptr;   // [NP] the reference's labelMat (-1 / 999999 / label), for inspection
};

__device__ __forceinline__ bool ip_pixel_of(const IpDev& ip, float x, float y, float z, int* pix) {
  const float va = (float)((double)(qm_atan2f(z, sqrtf(x * x + y * y)) * 180) / M_PI);
  const float rf = (va + ip.ang_bottom) / ip.ang_res_y;
  const long long r = (long long)rf;
  if (r < 0 || r >= ip.n_scan) return false;
  const float ha = (float)((double)(qm_atan2f(x, y) * 180) / M_PI);
  const double cd = -round(((double)ha - 90.0) / (double)ip.ang_res_x) + (double)(ip.horizon_scan / 2);
  long long c = (long long)cd;
  if (c < 0) return false;
  if (c >= ip.horizon_scan) c -= ip.horizon_scan;
  if (c < 0 || c >= ip.horizon_scan) return false;
  const float rg = sqrtf(x * x + y * y + z * z);
  if (rg < 0.1) return false;
  *pix = (int)r * ip.horizon_scan + (int)c;
  return true;
}

__global__ __launch_bounds__(256) void k_ip_project(const float4* __restrict__ pts, int P, IpDev ip, int* __restrict__ owner) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float4 p = pts[i];
  int pix;
  if (ip_pixel_of(ip, p.x, p.y, p.z, &pix)) atomicMax(&owner[pix], i);
}

__global__ __launch_bounds__(256) void k_ip_init(const float4* __restrict__ pts, int NP, const int* __restrict__ owner,
                                                 float* __restrict__ range, int* __restrict__ parent,
                                                 int* __restrict__ cnt, u64* __restrict__ rowmask) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= NP) return;
  const int o = owner[p];
  float rg = FLT_MAX;
  if (o >= 0) {
    const float4 q = pts[o];
    rg = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);
  }
  range[p] = rg;
  parent[p] = (o >= 0) ? p : -1;
  cnt[p] = 0;
  rowmask[p] = 0;
}

__device__ __forceinline__ int uf_find(const int* parent, int x) {
  int p = parent[x];
  while (p != x) {
    x = p;
    p = parent[x];
  }
  return x;
}
// find with path halving (used while the forest is still being built; the extra stores are benign races: every
// value written is an ancestor of the node it is written to)
__device__ __forceinline__ int uf_find_halve(int* parent, int x) {
  int p = parent[x];
  while (p != x) {
    const int g = parent[p];
    if (g != p) parent[x] = g;
    x = p;
    p = g;
  }
  return x;
}
// hook the larger root under the smaller one (roots end up being the smallest pixel index of their component)
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  while (true) {
    a = uf_find_halve(parent, a);
    b = uf_find_halve(parent, b);
    if (a == b) return;
    if (a > b) {
      const int t = a;
      a = b;
      b = t;
    }
    const int old = atomicMin(&parent[b], a);
    if (old == b) return;
    b = old;  // somebody re-parented b meanwhile: keep merging with what it points to now
  }
}

__device__ __forceinline__ bool ip_edge(float ra, float rb, float sn, float cs, float theta) {
  const float d1 = ra > rb ? ra : rb, d2 = ra > rb ? rb : ra;
  const float angle = qm_atan2f(d2 * sn, (d1 - d2 * cs));
  return angle > theta;
}

__global__ __launch_bounds__(256) void k_ip_merge(IpDev ip, const float* __restrict__ range, int* __restrict__ parent) {
  const int NP = ip.n_scan * ip.horizon_scan, H = ip.horizon_scan;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= NP || parent[p] < 0) return;
  const int r = p / H, c = p - r * H;
  const float rp = range[p];
  // forward half of the neighbourhood: the relation is symmetric, every undirected edge is visited once
  //   4-neighbour: (0,+1) (+1,0)   8-neighbour: + (+1,+1) (+1,-1)   4-cross: (+1,+1) (+1,-1)
  const bool axis = ip.neighbor_mode != 2, diag = ip.neighbor_mode != 0;
  if (axis) {
    const int q = r * H + (c + 1 >= H ? 0 : c + 1);
    if (parent[q] >= 0 && q != p && ip_edge(rp, range[q], ip.sx, ip.cx, ip.segment_theta)) uf_union(parent, p, q);
  }
  if (r + 1 < ip.n_scan) {
    if (axis) {
      const int q = (r + 1) * H + c;
      if (parent[q] >= 0 && ip_edge(rp, range[q], ip.sy, ip.cy, ip.segment_theta)) uf_union(parent, p, q);
    }
    if (diag) {
      const int q1 = (r + 1) * H + (c + 1 >= H ? 0 : c + 1), q2 = (r + 1) * H + (c - 1 < 0 ? H - 1 : c - 1);
      if (parent[q1] >= 0 && ip_edge(rp, range[q1], ip.sy, ip.cy, ip.segment_theta)) uf_union(parent, p, q1);
      if (parent[q2] >= 0 && ip_edge(rp, range[q2], ip.sy, ip.cy, ip.segment_theta)) uf_union(parent, p, q2);
    }
  }
}

__global__ __launch_bounds__(256) void k_ip_flatten(int NP, int H, const int* __restrict__ parent, int* __restrict__ cnt,
                                                    u64* __restrict__ rowmask, int* __restrict__ rootof) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= NP) return;
  if (parent[p] < 0) {
    rootof[p] = -1;
    return;
  }
  const int root = uf_find(parent, p);
  rootof[p] = root;
  atomicAdd(&cnt[root], 1);
  if (p != root) atomicOr(&rowmask[root], 1ULL << (p / H));  // lineCountFlag is only set for PUSHED pixels (:533)
}
// the same, with the atomics aggregated per wavefront: neighbouring pixels mostly share a root, and one global word
// only takes ~90 updates per microsecond on this part
__global__ __launch_bounds__(256) void k_ip_flatten_agg(int NP, int H, const int* __restrict__ parent, int* __restrict__ cnt,
                                                        u64* __restrict__ rowmask, int* __restrict__ rootof) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool valid = p < NP && parent[p] >= 0;
  const int root = valid ? uf_find(parent, p) : -1;
  if (p < NP) rootof[p] = root;
  const int lane = threadIdx.x & 63;
  const int row = p / H;
  const int row_a = __builtin_amdgcn_readfirstlane(row);  // 64 consecutive pixels span at most two rows (H >= 64)
  u64 todo = __ballot(valid);
  while (todo) {
    const int l = __ffsll((long long)todo) - 1;
    const int r0 = __builtin_amdgcn_readlane(root, l);
    const u64 same = __ballot(valid && root == r0);
    const u64 push_a = __ballot(valid && root == r0 && p != r0 && row == row_a);
    const u64 push_b = __ballot(valid && root == r0 && p != r0 && row != row_a);
    if (lane == l) {
      atomicAdd(&cnt[r0], __popcll(same));
      const u64 m = (push_a ? 1ULL << row_a : 0ULL) | (push_b ? 1ULL << (row_a + 1) : 0ULL);
      if (m) atomicOr(&rowmask[r0], m);
    }
    todo &= ~same;
  }
}
// 1024 pixels per workgroup; cls + the three per-block counts
__global__ __launch_bounds__(1024) void k_ip_classify(IpDev ip, const int* __restrict__ rootof, const int* __restrict__ cnt,
                                                      const u64* __restrict__ rowmask, int* __restrict__ cls,
                                                      int* __restrict__ blk, int nblk) {
  __shared__ int s_cnt[3];
  const int NP = ip.n_scan * ip.horizon_scan;
  const int p = blockIdx.x * 1024 + threadIdx.x;
  if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  int c = 0;
  if (p < NP) {
    const int root = rootof[p];
    if (root >= 0) {
      const int size = cnt[root];
      bool ok = size >= ip.num_min_pts;
      if (!ok && size >= ip.valid_point_num) ok = __popcll(rowmask[root]) >= ip.valid_line_num;
      c = ok ? 1 : 2;
      if (ok && root == p) c |= 4;
    }
    cls[p] = c;
  }
  const u64 bv = __ballot((c & 3) == 1), bo = __ballot((c & 3) == 2), br = __ballot((c & 4) != 0);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&s_cnt[0], __popcll(bv));
    atomicAdd(&s_cnt[1], __popcll(bo));
    atomicAdd(&s_cnt[2], __popcll(br));
  }
  __syncthreads();
  if (threadIdx.x < 3) blk[threadIdx.x * (nblk + 1) + blockIdx.x] = s_cnt[threadIdx.x];
}

__global__ __launch_bounds__(1024) void k_ip_blockscan(int* __restrict__ blk, int nblk, int* __restrict__ totals) {
  // three independent exclusive scans of nblk (<= 1024) counts; totals[0..2] = n_valid, n_outliers, n_segments
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = 0; k < 3; ++k) {
    int* a = blk + k * (nblk + 1);
    const int v = tid < nblk ? a[tid] : 0;
    int tot;
    const int ex = wave_excl_scan_i32(v, &tot);
    __syncthreads();
    if (lane == 63) wsum[wave] = tot;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
      woff += (w < wave) ? wsum[w] : 0;
      total += wsum[w];
    }
    if (tid < nblk) a[tid] = woff + ex;
    if (tid == 0) {
      a[nblk] = total;
      totals[k] = total;
    }
  }
}

// exclusive prefix of a predicate inside a 1024-thread workgroup
__device__ __forceinline__ int block_excl_1024(bool flag, int* wsum /* [16] shared */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u64 b = __ballot(flag);
  __syncthreads();
  if (lane == 0) wsum[wave] = __popcll(b);
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += wsum[w];
  return woff + __popcll(b & lanemask_lt());
}

__global__ __launch_bounds__(1024) void k_ip_rootrank(int NP, const int* __restrict__ cls, const int* __restrict__ blk,
                                                      int nblk, int* __restrict__ rootrank) {
  __shared__ int wsum[16];
  const int p = blockIdx.x * 1024 + threadIdx.x;
  const bool isroot = p < NP && (cls[p] & 4);
  const int ex = block_excl_1024(isroot, wsum);
  if (isroot) rootrank[p] = blk[2 * (nblk + 1) + blockIdx.x] + ex;
}

__global__ __launch_bounds__(1024) void k_ip_compact(const float4* __restrict__ pts, int NP, int H,
                                                     const int* __restrict__ owner, const int* __restrict__ rootof,
                                                     const int* __restrict__ cls, const int* __restrict__ blk, int nblk,
                                                     const int* __restrict__ rootrank, float4* __restrict__ out_valid,
                                                     float4* __restrict__ out_outl, int* __restrict__ labelmat) {
  __shared__ int wsum[16];
  const int p = blockIdx.x * 1024 + threadIdx.x;
  const int c = p < NP ? (cls[p] & 3) : 0;
  const int ev = block_excl_1024(c == 1, wsum);
  const int eo = block_excl_1024(c == 2, wsum);
  if (p >= NP) return;
  int lab = -1;
  if (c == 1) {
    lab = rootrank[rootof[p]] + 1;
    float4 q = pts[owner[p]];
    q.w = (float)lab;
    out_valid[blk[blockIdx.x] + ev] = q;
  } else if (c == 2) {
    lab = 999999;
    float4 q = pts[owner[p]];
    q.w = (float)(p / H) + (float)(p % H) / 10000.0f;
    out_outl[blk[(nblk + 1) + blockIdx.x] + eo] = q;
  }
  labelmat[p] = lab;
}

// ------------------------------------------------------------------------------------------------ host side
size_t segment_scratch_bytes(int np_cap) {
  const size_t nblk = ((size_t)np_cap + 1023) / 1024;
  return (size_t)np_cap * (4 + 4 + 4 + 4 + 8 + 4 + 4 + 4 + 4 + 16 + 16) + 3 * (nblk + 1) * 4 + 4096 + 16 * 256;
}
void segment_carve(SegBufs& S, void* base, int np_cap) {
  char* p = (char*)base;
  auto take = [&](size_t bytes) {
    char* r = p;
    p += (bytes + 255) & ~(size_t)255;
    return (void*)r;
  };
  S.np_cap = np_cap;
  const size_t nblk = ((size_t)np_cap + 1023) / 1024;
  S.rowmask = (u64*)take((size_t)np_cap * 8);
  S.out_valid = (float4*)take((size_t)np_cap * 16);
  S.out_outl = (float4*)take((size_t)np_cap * 16);
  S.owner = (int*)take((size_t)np_cap * 4);
  S.range = (float*)take((size_t)np_cap * 4);
  S.parent = (int*)take((size_t)np_cap * 4);
  S.cnt = (int*)take((size_t)np_cap * 4);
  S.cls = (int*)take((size_t)np_cap * 4);
  S.rootrank = (int*)take((size_t)np_cap * 4);
  S.labelmat = (int*)take((size_t)np_cap * 4);
  S.blk = (int*)take((3 * (nblk + 1) + 8) * 4);
}

// `rootof` (root of every pixel) is kept in S.labelmat until k_ip_compact, whose threads read rootof[p] before they
// overwrite labelmat[p] (same index, same thread).
hipError_t segment_enqueue(const SegBufs& S, const float4* pts, int P, const IpDev& ip, int* totals /* device, 3 ints */,
                           hipStream_t st) {
  const int NP = ip.n_scan * ip.horizon_scan, H = ip.horizon_scan;
  const int nblk = (NP + 1023) / 1024;
  hipError_t e;
  (void)hipGetLastError();
  if ((e = hipMemsetAsync(S.owner, 0xff, (size_t)NP * 4, st)) != hipSuccess) return e;
  if (P > 0) hipLaunchKernelGGL(k_ip_project, dim3((P + 255) / 256), dim3(256), 0, st, pts, P, ip, S.owner);
  hipLaunchKernelGGL(k_ip_init, dim3((NP + 255) / 256), dim3(256), 0, st, pts, NP, S.owner, S.range, S.parent, S.cnt,
                     S.rowmask);
  hipLaunchKernelGGL(k_ip_merge, dim3((NP + 255) / 256), dim3(256), 0, st, ip, S.range, S.parent);
  int* rootof = S.labelmat;
  if (H >= 64)
    hipLaunchKernelGGL(k_ip_flatten_agg, dim3((NP + 255) / 256), dim3(256), 0, st, NP, H, S.parent, S.cnt, S.rowmask, rootof);
  else
    hipLaunchKernelGGL(k_ip_flatten, dim3((NP + 255) / 256), dim3(256), 0, st, NP, H, S.parent, S.cnt, S.rowmask, rootof);
  hipLaunchKernelGGL(k_ip_classify, dim3(nblk), dim3(1024), 0, st, ip, rootof, S.cnt, S.rowmask, S.cls, S.blk, nblk);
  hipLaunchKernelGGL(k_ip_blockscan, dim3(1), dim3(1024), 0, st, S.blk, nblk, totals);
  hipLaunchKernelGGL(k_ip_rootrank, dim3(nblk), dim3(1024), 0, st, NP, S.cls, S.blk, nblk, S.rootrank);
  hipLaunchKernelGGL(k_ip_compact, dim3(nblk), dim3(1024), 0, st, pts, NP, H, S.owner, rootof, S.cls, S.blk, nblk,
                     S.rootrank, S.out_valid, S.out_outl, S.labelmat);
  return hipGetLastError();
}

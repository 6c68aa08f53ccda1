// patchwork.hip — "next" row (f)2 of SURVEY.md section 8: Patchwork ground segmentation, the first stage of the
// reference demo on raw scans (PatchWork::estimate_ground, include/patchwork.hpp:329-476; concentric zone model
// :512-546, seed selection :285-318, plane fit :271-283, region-wise fit :549-586).  Device plan:
//   K1 keys     (order-preserving z bits << 32) | point index, then the stable LSD radix sort of the front end
//   K2 bin      sorted point -> patch id (zone, ring, sector) or "dropped"; second stable sort by patch id keeps
//               the z order inside every patch; per-patch counts
//   K3 starts   exclusive scan of the (<= 1024) patch counts
//   K4 patch    ONE WAVEFRONT PER PATCH: lowest-point seeds, num_iter rounds of {nine moment sums in the fixed
//               sum64 order of qtr_math.h, closed-form smallest eigenpair (the normals' pcl::eigen33 restatement),
//               point-to-plane test}, then the uprightness / elevation / flatness decision of :376-428
//   K5 scan     output offsets of every patch in the reference's zone / ring / sector order
//   K6 emit     stable per-patch compaction into the ground / non-ground clouds
// The CPU restatement used by the tests defines the same arithmetic; outputs are compared bit for bit.
#include "common.h"
#include "../../include/qtr_math.h"

struct PwDev {  // qtr_pw_params + derived constants
  double sensor_height;
  int num_iter, num_lpr, num_min_pts;
  double th_seeds, th_dist, max_range, min_range, uprightness_thr, margin;
  int using_global_thr;
  double global_elevation_thr;
  int num_zones;
  int nsec[4], nring[4];
  double min_ranges[4], ring_size[4], sector_size[4];
  int base[5];
  int ring_base[4];  // concentric index of the first ring of a zone
  int num_thr;
  double elevation_thr[8], flatness_thr[8];
};
#define PW_DROPPED 0xffffu

struct PwBufs {
  int p_cap = 0;
  unsigned char* flag = nullptr;  // [P] per sorted position: bit 0 in current ground set, bit 1 final ground
  int* counts = nullptr;          // [1024] points per patch
  int* starts = nullptr;          // [1025]
  int* info = nullptr;            // [1024][4]: processed, reject_all, n_ground, n
  int* offs = nullptr;            // [1024][2] output offsets (ground, nonground) + totals at [1024]
  float4* out_g = nullptr;        // [P]
  float4* out_n = nullptr;        // [P]
};

__global__ __launch_bounds__(256) void k_pw_keys(const float4* __restrict__ pts, int P, u64* __restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < P) keys[i] = ((u64)enc_f32(pts[i].z) << 32) | (u32)i;
}

__global__ __launch_bounds__(256) void k_pw_bin(const float4* __restrict__ pts, int P, PwDev pw,
                                                const u64* __restrict__ zsorted, u64* __restrict__ keys2,
                                                int* __restrict__ counts) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= P) return;
  const u32 i = (u32)zsorted[t];
  const float4 q = pts[i];
  u32 pid = PW_DROPPED;
  if (!((double)q.z < -1.8 * pw.sensor_height)) {
    const double x = q.x, y = q.y;
    const double r = sqrt(x * x + y * y);
    if ((r <= pw.max_range) && (r > pw.min_range)) {
      const double at = qm_atan2d(y, x);
      const double theta = at > 0 ? at : at + 2 * M_PI;
      int k = pw.num_zones - 1;
      for (int z = 1; z < pw.num_zones; ++z)
        if (r < pw.min_ranges[z]) {
          k = z - 1;
          break;
        }
      const int ring = min((int)((r - pw.min_ranges[k]) / pw.ring_size[k]), pw.nring[k] - 1);
      const int sector = min((int)(theta / pw.sector_size[k]), pw.nsec[k] - 1);
      pid = (u32)(pw.base[k] + ring * pw.nsec[k] + sector);
      atomicAdd(&counts[pid], 1);
    }
  }
  keys2[t] = ((u64)pid << 32) | i;
}

__global__ __launch_bounds__(1024) void k_pw_starts(const int* __restrict__ counts, int npatch, int* __restrict__ starts) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int v = tid < npatch ? counts[tid] : 0;
  int tot;
  const int ex = wave_excl_scan_i32(v, &tot);
  if (lane == 63) wsum[wave] = tot;
  __syncthreads();
  int woff = 0, total = 0;
  for (int w = 0; w < 16; ++w) {
    woff += (w < wave) ? wsum[w] : 0;
    total += wsum[w];
  }
  if (tid < npatch) starts[tid] = woff + ex;
  if (tid == 0) starts[npatch] = total;
}


// smallest eigenpair + the three |eigenvalues| (descending) of a symmetric 3x3 (row-major 9 floats); the same
// operations, in the same order, as the oracle's eigen33_smallest + compute_roots
__device__ __forceinline__ void pw_eigen(const float* cov, float* nrm, float* sv) {
  float scale = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) scale = fmaxf(scale, fabsf(cov[t]));
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = cov[t] / scale;
  float roots[3];
  dev_roots(s, roots);
  float a0 = fabsf(roots[0] * scale), a1 = fabsf(roots[1] * scale), a2 = fabsf(roots[2] * scale), tmp;
  if (a0 < a1) {
    tmp = a0;
    a0 = a1;
    a1 = tmp;
  }
  if (a1 < a2) {
    tmp = a1;
    a1 = a2;
    a2 = tmp;
  }
  if (a0 < a1) {
    tmp = a0;
    a0 = a1;
    a1 = tmp;
  }
  sv[0] = a0;
  sv[1] = a1;
  sv[2] = a2;
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
  const bool flip = vz < 0.f || (vz == 0.f && (vy < 0.f || (vy == 0.f && vx < 0.f)));
  nrm[0] = flip ? -vx : vx;
  nrm[1] = flip ? -vy : vy;
  nrm[2] = flip ? -vz : vz;
}

// one wavefront per patch
__global__ __launch_bounds__(64) void k_pw_patch(const float4* __restrict__ pts, PwDev pw, const u64* __restrict__ sorted,
                                                 const int* __restrict__ starts, unsigned char* __restrict__ flag,
                                                 int* __restrict__ info) {
  const int pid = blockIdx.x, lane = threadIdx.x;
  const int s0 = starts[pid], n = starts[pid + 1] - s0;
  int k = 0;
  while (k + 1 < pw.num_zones && pid >= pw.base[k + 1]) ++k;
  const int ring = (pid - pw.base[k]) / pw.nsec[k];
  if (!(n > pw.num_min_pts)) {
    if (lane == 0) {
      info[4 * pid] = 0;
      info[4 * pid + 1] = 0;
      info[4 * pid + 2] = 0;
      info[4 * pid + 3] = n;
    }
    return;
  }
  // seeds: skip the lowest points of the innermost zone, mean of the next num_lpr heights (binary64, in order)
  int init_idx = 0;
  if (k == 0) {
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int t = c0 + lane;
      const bool low = t < n && (double)pts[(u32)sorted[s0 + t]].z < pw.margin;
      const u64 b = __ballot(low);
      init_idx += __popcll(b);  // heights ascend inside a patch: the low ones are a prefix
      if (b != ~0ULL) break;
    }
  }
  double sum = 0;
  int cnt = 0;
  for (int t = init_idx; t < n && cnt < pw.num_lpr; ++t) {  // (uniform, num_lpr = 20 loads)
    sum += (double)pts[(u32)sorted[s0 + t]].z;
    ++cnt;
  }
  const double lpr_height = cnt != 0 ? sum / cnt : 0;
  for (int t = lane; t < n; t += 64)
    flag[s0 + t] = ((double)pts[(u32)sorted[s0 + t]].z < lpr_height + pw.th_seeds) ? 1 : 0;
  float nrm[3] = {0, 0, 1}, mean[3] = {0, 0, 0}, sv[3] = {0, 0, 0}, th_dist_d = 0;
  int n_ground = 0;
  for (int it = 0; it < pw.num_iter; ++it) {
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int members = 0;
    for (int t = lane; t < n; t += 64) {  // patch position t feeds partial[t & 63] = this lane
      if (flag[s0 + t] & 1) {
        const float4 q = pts[(u32)sorted[s0 + t]];
        acc[0] += q.x * q.x;
        acc[1] += q.x * q.y;
        acc[2] += q.x * q.z;
        acc[3] += q.y * q.y;
        acc[4] += q.y * q.z;
        acc[5] += q.z * q.z;
        acc[6] += q.x;
        acc[7] += q.y;
        acc[8] += q.z;
        ++members;
      }
    }
    members = wave_sum_i32(members);
#pragma unroll
    for (int a = 0; a < 9; ++a) acc[a] = wave_sum64_f32(acc[a]);
    const float kk = (float)members;
#pragma unroll
    for (int a = 0; a < 9; ++a) acc[a] /= kk;
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
    pw_eigen(cov, nrm, sv);
    mean[0] = acc[6];
    mean[1] = acc[7];
    mean[2] = acc[8];
    const float d = -((nrm[0] * mean[0] + nrm[1] * mean[1]) + nrm[2] * mean[2]);
    th_dist_d = (float)(pw.th_dist - (double)d);
    const bool last = it == pw.num_iter - 1;
    int ng = 0;
    for (int t = lane; t < n; t += 64) {
      const float4 q = pts[(u32)sorted[s0 + t]];
      const float res = (q.x * nrm[0] + q.y * nrm[1]) + q.z * nrm[2];
      const bool g = res < th_dist_d;
      flag[s0 + t] = last ? (g ? 2 : 0) : (g ? 1 : 0);
      ng += g ? 1 : 0;
    }
    n_ground = wave_sum_i32(ng);
  }
  // patch status (:376-428)
  const double ground_z_vec = fabs((double)nrm[2]);
  const double ground_z_elevation = mean[2];
  const double surface_variable = (double)sv[2] / (double)((sv[0] + sv[1]) + sv[2]);
  const int concentric_idx = pw.ring_base[k] + ring;
  bool reject_all = false;
  if (ground_z_vec < pw.uprightness_thr)
    reject_all = true;
  else if (concentric_idx < pw.num_thr) {
    const int ti = ring + 2 * k;
    if (ground_z_elevation > pw.elevation_thr[ti] && !(pw.flatness_thr[ti] > surface_variable)) reject_all = true;
  } else if (pw.using_global_thr && ground_z_elevation > pw.global_elevation_thr)
    reject_all = true;
  if (lane == 0) {
    info[4 * pid] = 1;
    info[4 * pid + 1] = reject_all ? 1 : 0;
    info[4 * pid + 2] = n_ground;
    info[4 * pid + 3] = n;
  }
}

__global__ __launch_bounds__(1024) void k_pw_scan(const int* __restrict__ info, int npatch, int* __restrict__ offs) {
  __shared__ int wsum[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g = 0, nn = 0;
  if (tid < npatch && info[4 * tid]) {
    const int rej = info[4 * tid + 1], ng = info[4 * tid + 2], n = info[4 * tid + 3];
    g = rej ? 0 : ng;
    nn = (rej ? ng : 0) + (n - ng);
  }
  int tg, tn;
  const int eg = wave_excl_scan_i32(g, &tg), en = wave_excl_scan_i32(nn, &tn);
  if (lane == 63) {
    wsum[0][wave] = tg;
    wsum[1][wave] = tn;
  }
  __syncthreads();
  int og = 0, on = 0, totg = 0, totn = 0;
  for (int w = 0; w < 16; ++w) {
    og += (w < wave) ? wsum[0][w] : 0;
    on += (w < wave) ? wsum[1][w] : 0;
    totg += wsum[0][w];
    totn += wsum[1][w];
  }
  if (tid < npatch) {
    offs[2 * tid] = og + eg;
    offs[2 * tid + 1] = on + en;
  }
  if (tid == 0) {
    offs[2 * 1024] = totg;
    offs[2 * 1024 + 1] = totn;
  }
}

__global__ __launch_bounds__(64) void k_pw_emit(const float4* __restrict__ pts, const u64* __restrict__ sorted,
                                                const int* __restrict__ starts, const unsigned char* __restrict__ flag,
                                                const int* __restrict__ info, const int* __restrict__ offs,
                                                float4* __restrict__ out_g, float4* __restrict__ out_n) {
  const int pid = blockIdx.x, lane = threadIdx.x;
  if (!info[4 * pid]) return;
  const int s0 = starts[pid], n = info[4 * pid + 3], rej = info[4 * pid + 1], ng = info[4 * pid + 2];
  float4* gdst = rej ? out_n + offs[2 * pid + 1] : out_g + offs[2 * pid];   // ground-classified points first
  float4* ndst = out_n + offs[2 * pid + 1] + (rej ? ng : 0);              // then the rest
  int cg = 0, cn = 0;
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int t = c0 + lane;
    const bool valid = t < n;
    const bool g = valid && (flag[s0 + t] & 2);
    const u64 bg = __ballot(g), bn = __ballot(valid && !g);
    if (valid) {
      const float4 q = pts[(u32)sorted[s0 + t]];
      if (g)
        gdst[cg + __popcll(bg & lanemask_lt())] = q;
      else
        ndst[cn + __popcll(bn & lanemask_lt())] = q;
    }
    cg += __popcll(bg);
    cn += __popcll(bn);
  }
}

// ------------------------------------------------------------------------------------------------ host side
size_t patchwork_scratch_bytes(int p_cap) { return (size_t)p_cap * (1 + 16 + 16) + 1024 * 4 * 12 + 8192; }
void patchwork_carve(PwBufs& B, void* basep, int p_cap) {
  char* p = (char*)basep;
  auto take = [&](size_t bytes) {
    char* r = p;
    p += (bytes + 255) & ~(size_t)255;
    return (void*)r;
  };
  B.p_cap = p_cap;
  B.out_g = (float4*)take((size_t)p_cap * 16);
  B.out_n = (float4*)take((size_t)p_cap * 16);
  B.counts = (int*)take(1024 * 4);
  B.starts = (int*)take(1025 * 4);
  B.info = (int*)take(1024 * 16);
  B.offs = (int*)take((2 * 1024 + 2) * 4);
  B.flag = (unsigned char*)take((size_t)p_cap);
}

// F.cloud[0]'s key buffers and histogram area serve the two radix sorts
hipError_t patchwork_enqueue(FrontBufs& F, const PwBufs& B, const float4* pts, int P, const PwDev& pw, hipStream_t st) {
  hipError_t e;
  (void)hipGetLastError();
  const int npatch = pw.base[pw.num_zones];
  if ((e = hipMemsetAsync(B.counts, 0, 1024 * 4, st)) != hipSuccess) return e;
  CloudBufs* C[2] = {&F.cloud[0], &F.cloud[0]};
  Clouds2 a;
  a.c[0] = make_view(*C[0], pts, P, 0);
  a.c[1] = a.c[0];
  if (P > 0) {
    hipLaunchKernelGGL(k_pw_keys, dim3((P + 255) / 256), dim3(256), 0, st, pts, P, C[0]->keys_a);
    const int w1 = radix_sort2(a, C, 1, 0, 32, st);
    u64* zs = w1 == 0 ? C[0]->keys_a : C[0]->keys_b;
    hipLaunchKernelGGL(k_pw_bin, dim3((P + 255) / 256), dim3(256), 0, st, pts, P, pw, zs, C[0]->keys_a, B.counts);
    const int w2 = radix_sort2(a, C, 1, 0, 16, st);
    const u64* sorted = w2 == 0 ? C[0]->keys_a : C[0]->keys_b;
    hipLaunchKernelGGL(k_pw_starts, dim3(1), dim3(1024), 0, st, B.counts, npatch, B.starts);
    hipLaunchKernelGGL(k_pw_patch, dim3(npatch), dim3(64), 0, st, pts, pw, sorted, B.starts, B.flag, B.info);
    hipLaunchKernelGGL(k_pw_scan, dim3(1), dim3(1024), 0, st, B.info, npatch, B.offs);
    hipLaunchKernelGGL(k_pw_emit, dim3(npatch), dim3(64), 0, st, pts, sorted, B.starts, B.flag, B.info, B.offs, B.out_g,
                       B.out_n);
  } else {
    if ((e = hipMemsetAsync(B.offs + 2 * 1024, 0, 8, st)) != hipSuccess) return e;
  }
  return hipGetLastError();
}

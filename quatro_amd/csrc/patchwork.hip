// patchwork.hip — "next" row (f)2 of SURVEY.md section 8: Patchwork ground segmentation, the first stage of the
// reference demo on raw scans (PatchWork::estimate_ground, include/patchwork.hpp:329-476; concentric zone model
// :512-546, seed selection :285-318, plane fit :271-283, region-wise fit :549-586).  Device plan:
//   K1 keys     (order-preserving z bits << 32) | point index, then the stable LSD radix sort of the front end
//   K2 bin      sorted point -> patch id (zone, ring, sector) or "dropped"; second stable sort by patch id keeps
//               the z order inside every patch
//   K3 bounds   first / one-past-last position of every patch from the sorted keys (no atomics: each boundary has
//               one writer), and the points gathered into sorted order so K4/K6 stream them
//   K4 patch    ONE 256-THREAD WORKGROUP PER PATCH: lowest-point seeds, num_iter rounds of {nine moment sums over the
//               current ground set — membership is re-evaluated from the previous plane in the same pass — in the
//               fixed sum256 order, closed-form smallest eigenpair (the normals' pcl::eigen33 restatement)}, a last
//               pass for the final membership, then the uprightness / elevation / flatness decision of :376-428
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
  unsigned char* flag = nullptr;  // [P] per sorted position: 1 = final ground
  int* first = nullptr;           // [1024] first sorted position of a patch
  int* last = nullptr;            // [1024] one past its last position (both 0 when the patch is empty)
  float4* spts = nullptr;         // [P] points in sorted (patch, height) order
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
                                                const u64* __restrict__ zsorted, u64* __restrict__ keys2) {
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
    }
  }
  keys2[t] = ((u64)pid << 32) | i;
}

__global__ __launch_bounds__(256) void k_pw_bounds(const float4* __restrict__ pts, int P, const u64* __restrict__ sorted,
                                                   int* __restrict__ first, int* __restrict__ last,
                                                   float4* __restrict__ spts) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= P) return;
  const u64 key = sorted[t];
  const u32 pid = (u32)(key >> 32);
  spts[t] = pts[(u32)key];
  if (pid == PW_DROPPED) return;
  if (t == 0 || (u32)(sorted[t - 1] >> 32) != pid) first[pid] = t;
  if (t == P - 1 || (u32)(sorted[t + 1] >> 32) != pid) last[pid] = t + 1;
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

// one 256-thread workgroup per patch; position t of the patch feeds moment slot t & 255 (= this thread)
__global__ __launch_bounds__(256) void k_pw_patch(const float4* __restrict__ spts, PwDev pw, const int* __restrict__ first,
                                                  const int* __restrict__ last, unsigned char* __restrict__ flag,
                                                  int* __restrict__ info) {
  __shared__ double zl[256];
  __shared__ float red[9][4];
  __shared__ int redi[4];
  const int pid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s0 = first[pid], n = last[pid] - s0;
  int k = 0;
  while (k + 1 < pw.num_zones && pid >= pw.base[k + 1]) ++k;
  const int ring = (pid - pw.base[k]) / pw.nsec[k];
  if (!(n > pw.num_min_pts)) {
    if (tid == 0) {
      info[4 * pid] = 0;
      info[4 * pid + 1] = 0;
      info[4 * pid + 2] = 0;
      info[4 * pid + 3] = n;
    }
    return;
  }
  const float4* __restrict__ q = spts + s0;
  // seeds: skip the lowest points of the innermost zone, mean of the next num_lpr heights (binary64, in order)
  int init_idx = 0;
  if (k == 0) {
    for (int c0 = 0; c0 < n; c0 += 256) {
      const int t = c0 + tid;
      const int c = __syncthreads_count(t < n && (double)q[t].z < pw.margin);
      init_idx += c;  // heights ascend inside a patch: the low ones are a prefix
      if (c != 256) break;
    }
  }
  double sum = 0;
  const int cnt = min(pw.num_lpr, n - init_idx);
  for (int b = 0; b < cnt; b += 256) {
    __syncthreads();
    if (b + tid < cnt) zl[tid] = (double)q[init_idx + b + tid].z;
    __syncthreads();
    const int m = min(256, cnt - b);
    for (int j = 0; j < m; ++j) sum += zl[j];
  }
  const double lpr_height = cnt != 0 ? sum / cnt : 0;
  const double seed_thr = lpr_height + pw.th_seeds;
  float nrm[3] = {0, 0, 1}, mean[3] = {0, 0, 0}, sv[3] = {0, 0, 0}, th_dist_d = 0;
  for (int it = 0; it < pw.num_iter; ++it) {
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int members = 0;
    for (int t = tid; t < n; t += 256) {
      const float4 p = q[t];
      bool in;
      if (it == 0)
        in = (double)p.z < seed_thr;
      else
        in = ((p.x * nrm[0] + p.y * nrm[1]) + p.z * nrm[2]) < th_dist_d;
      if (in) {
        acc[0] += p.x * p.x;
        acc[1] += p.x * p.y;
        acc[2] += p.x * p.z;
        acc[3] += p.y * p.y;
        acc[4] += p.y * p.z;
        acc[5] += p.z * p.z;
        acc[6] += p.x;
        acc[7] += p.y;
        acc[8] += p.z;
        ++members;
      }
    }
    members = wave_sum_i32(members);
#pragma unroll
    for (int a = 0; a < 9; ++a) acc[a] = wave_sum64_f32(acc[a]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 9; ++a) red[a][wave] = acc[a];
      redi[wave] = members;
    }
    __syncthreads();
    members = (redi[0] + redi[1]) + (redi[2] + redi[3]);
#pragma unroll
    for (int a = 0; a < 9; ++a) acc[a] = (red[a][0] + red[a][1]) + (red[a][2] + red[a][3]);
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
  }
  int ng = 0;
  for (int t = tid; t < n; t += 256) {
    const float4 p = q[t];
    const bool g = ((p.x * nrm[0] + p.y * nrm[1]) + p.z * nrm[2]) < th_dist_d;
    flag[s0 + t] = g ? 1 : 0;
    ng += g ? 1 : 0;
  }
  ng = wave_sum_i32(ng);
  __syncthreads();
  if (lane == 0) redi[wave] = ng;
  __syncthreads();
  const int n_ground = (redi[0] + redi[1]) + (redi[2] + redi[3]);
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
  if (tid == 0) {
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

__global__ __launch_bounds__(256) void k_pw_emit(const float4* __restrict__ spts, const int* __restrict__ first,
                                                 const unsigned char* __restrict__ flag, const int* __restrict__ info,
                                                 const int* __restrict__ offs, float4* __restrict__ out_g,
                                                 float4* __restrict__ out_n) {
  __shared__ int wc[2][4];
  const int pid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (!info[4 * pid]) return;
  const int s0 = first[pid], n = info[4 * pid + 3], rej = info[4 * pid + 1], ng = info[4 * pid + 2];
  float4* gdst = rej ? out_n + offs[2 * pid + 1] : out_g + offs[2 * pid];   // ground-classified points first
  float4* ndst = out_n + offs[2 * pid + 1] + (rej ? ng : 0);              // then the rest
  int cg = 0, cn = 0;
  for (int c0 = 0; c0 < n; c0 += 256) {
    const int t = c0 + tid;
    const bool valid = t < n;
    const bool g = valid && flag[s0 + t];
    const u64 bg = __ballot(g), bn = __ballot(valid && !g);
    __syncthreads();
    if (lane == 0) {
      wc[0][wave] = __popcll(bg);
      wc[1][wave] = __popcll(bn);
    }
    __syncthreads();
    int og = 0, on = 0, tg = 0, tn = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      og += w < wave ? wc[0][w] : 0;
      on += w < wave ? wc[1][w] : 0;
      tg += wc[0][w];
      tn += wc[1][w];
    }
    if (valid) {
      const float4 p = spts[s0 + t];
      if (g)
        gdst[cg + og + __popcll(bg & lanemask_lt())] = p;
      else
        ndst[cn + on + __popcll(bn & lanemask_lt())] = p;
    }
    cg += tg;
    cn += tn;
  }
}

// ------------------------------------------------------------------------------------------------ host side
size_t patchwork_scratch_bytes(int p_cap) { return (size_t)p_cap * (1 + 16 + 16 + 16) + 1024 * 4 * 12 + 8192; }
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
  B.spts = (float4*)take((size_t)p_cap * 16);
  B.first = (int*)take(2 * 1024 * 4);
  B.last = B.first + 1024;
  B.info = (int*)take(1024 * 16);
  B.offs = (int*)take((2 * 1024 + 2) * 4);
  B.flag = (unsigned char*)take((size_t)p_cap);
}

// F.cloud[0]'s key buffers and histogram area serve the two radix sorts
hipError_t patchwork_enqueue(FrontBufs& F, const PwBufs& B, const float4* pts, int P, const PwDev& pw, hipStream_t st) {
  hipError_t e;
  (void)hipGetLastError();
  const int npatch = pw.base[pw.num_zones];
  if ((e = hipMemsetAsync(B.first, 0, 2 * 1024 * 4, st)) != hipSuccess) return e;
  CloudBufs* C[2] = {&F.cloud[0], &F.cloud[0]};
  CloudSet S;
  {
    const CloudView v = make_view(*C[0], pts, P, 0, nullptr, nullptr, 0);
    if ((e = cloudset_finish(S, &v, 1, nullptr, st)) != hipSuccess) return e;
  }
  if (P > 0) {
    hipLaunchKernelGGL(k_pw_keys, dim3((P + 255) / 256), dim3(256), 0, st, pts, P, C[0]->keys_a);
    const int w1 = radix_sort2(S, 0, 32, st);
    u64* zs = w1 == 0 ? C[0]->keys_a : C[0]->keys_b;
    hipLaunchKernelGGL(k_pw_bin, dim3((P + 255) / 256), dim3(256), 0, st, pts, P, pw, zs, C[0]->keys_a);
    const int w2 = radix_sort2(S, 0, 16, st);
    const u64* sorted = w2 == 0 ? C[0]->keys_a : C[0]->keys_b;
    hipLaunchKernelGGL(k_pw_bounds, dim3((P + 255) / 256), dim3(256), 0, st, pts, P, sorted, B.first, B.last, B.spts);
    hipLaunchKernelGGL(k_pw_patch, dim3(npatch), dim3(256), 0, st, B.spts, pw, B.first, B.last, B.flag, B.info);
    hipLaunchKernelGGL(k_pw_scan, dim3(1), dim3(1024), 0, st, B.info, npatch, B.offs);
    hipLaunchKernelGGL(k_pw_emit, dim3(npatch), dim3(256), 0, st, B.spts, B.first, B.flag, B.info, B.offs, B.out_g, B.out_n);
  } else {
    if ((e = hipMemsetAsync(B.offs + 2 * 1024, 0, 8, st)) != hipSuccess) return e;
  }
  return hipGetLastError();
}

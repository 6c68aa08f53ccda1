// exact.hip — "next" row (f)4 of SURVEY.md section 8: exact maximum clique (teaser::MaxCliqueSolver in PMC_EXACT mode,
// reference src/graph.cc:106-127 -> PMC's pmcx_maxclique branch-and-bound).  Runs after the heuristic of solver.hip
// on the same rank-relabelled bit matrix (ranks ascend with (core, id)):
//   * only ranks >= t0 (core + 1 > heuristic size) can belong to a larger clique;
//   * ONE WAVEFRONT PER ROOT VERTEX, roots handed out in ascending rank by an atomic counter; the candidates of a
//     root are its later neighbours; children are taken in descending rank.  A candidate set is a bit set spread over
//     the wavefront (word j on lane j mod 64), so "intersect with a neighbourhood" is one coalesced row load and an
//     AND, "how many are left" a popcount + wave reduction, "next child" a clz + wave maximum.  The per-level sets live
//     on a per-wave stack in HBM/L2.
//   * bounds: |C| + |P|, then a greedy colouring of P (stopped as soon as it cannot prune).
//   * phase A finds omega with a shared incumbent (atomicMax); phase B, run only when omega beats the heuristic,
//     finds the FIRST clique of size omega in (root ascending, children descending) depth-first order — the defined
//     result (see the CPU restatement's D10) — with roots behind an already successful root abandoned.
#include <climits>

#include "solver.h"

struct ExactCtl {
  int gbest;       // phase A incumbent size; omega afterwards
  int next_root;   // next root rank to hand out
  int first_root;  // phase B: lowest (root << 16 | top-level child index) whose subtree holds a clique of size omega
  int t0;          // first rank with Kp > lb
  int abort;       // time limit hit
  int lb, ub, winner;
  int task;        // small-graph kernel: next (root, slice) task of the current launch
  unsigned long long nodes;
  long long t_start;
};

__global__ void k_exact_init(const int* __restrict__ Kp, int L, const SolverState* __restrict__ st, ExactCtl* ctl) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int lb = st->mc;
  int lo = 0, hi = L;  // first rank with Kp > lb (Kp ascends with rank)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (Kp[mid] > lb) hi = mid; else lo = mid + 1;
  }
  ctl->gbest = lb;
  ctl->lb = lb;
  ctl->ub = st->ub;
  ctl->t0 = lo;
  ctl->next_root = lo;
  ctl->first_root = INT_MAX;
  ctl->abort = 0;
  ctl->winner = -1;
  ctl->task = 0;
  ctl->nodes = 0;
  ctl->t_start = (long long)wall_clock64();
}

__global__ void k_exact_phase_b(const int* __restrict__ Kp, int L, ExactCtl* ctl) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int need = ctl->gbest;  // roots need core + 1 >= omega
  int lo = 0, hi = L;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (Kp[mid] >= need) hi = mid; else lo = mid + 1;
  }
  ctl->next_root = lo;
  ctl->first_root = INT_MAX;
  ctl->task = 0;
}

template <int NW>
__device__ __forceinline__ int ex_count(const u64 (&p)[NW]) {
  int c = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) c += __popcll(p[i]);
  return wave_sum_i32(c);
}
template <int NW>
__device__ __forceinline__ int ex_highest(const u64 (&p)[NW], int lane) {
  int hi = -1;
#pragma unroll
  for (int i = 0; i < NW; ++i)
    if (p[i]) hi = max(hi, (lane + 64 * i) * 64 + 63 - __clzll((long long)p[i]));
  return wave_max_i32(hi);
}
template <int NW>
__device__ __forceinline__ void ex_clear(u64 (&p)[NW], int v, int lane) {
  const int j = v >> 6;
#pragma unroll
  for (int i = 0; i < NW; ++i)
    if (lane + 64 * i == j) p[i] &= ~(1ULL << (v & 63));
}
template <int NW>
__device__ __forceinline__ void ex_row(const u64* __restrict__ adjP, int W, int v, int lane, u64 (&row)[NW]) {
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int j = lane + 64 * i;
    row[i] = j < W ? adjP[(size_t)v * W + j] : 0ULL;
  }
}

// classes of a greedy colouring of p (each class grown from the highest rank down), stopped at limit + 1
template <int NW>
__device__ __forceinline__ int ex_colour_bound(const u64 (&p)[NW], int limit, const u64* __restrict__ adjP, int W, int lane) {
  u64 u[NW], q[NW], row[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) u[i] = p[i];
  int colours = 0;
  while (true) {
    bool any = false;
#pragma unroll
    for (int i = 0; i < NW; ++i) any |= u[i] != 0;
    if (!__any(any)) break;
    if (++colours > limit) return limit + 1;
#pragma unroll
    for (int i = 0; i < NW; ++i) q[i] = u[i];
    while (true) {
      const int v = ex_highest<NW>(q, lane);
      if (v < 0) break;
      ex_clear<NW>(q, v, lane);
      ex_clear<NW>(u, v, lane);
      ex_row<NW>(adjP, W, v, lane, row);
#pragma unroll
      for (int i = 0; i < NW; ++i) q[i] &= ~row[i];
    }
  }
  return colours;
}

// cliq: per wave [depth_cap + 2] ints: root (-1 none), size, members (rank labels)
template <int NW>
__global__ __launch_bounds__(64) void k_exact_search(const u64* __restrict__ adjP, const int* __restrict__ Kp, int L, int W,
                                                     ExactCtl* ctl, u64* __restrict__ stack, int depth_cap,
                                                     int* __restrict__ cliq, int phase_b, long long tick_limit) {
  extern __shared__ int Cl[];  // current clique, rank labels
  const int lane = threadIdx.x, wave = blockIdx.x;
  u64* stk = stack + (size_t)wave * depth_cap * NW * 64;
  int* mine = cliq + (size_t)wave * (depth_cap + 2);
  volatile int* v_gbest = &ctl->gbest;
  volatile int* v_first = &ctl->first_root;
  volatile int* v_abort = &ctl->abort;
  const int omega = ctl->gbest;  // phase B: fixed
  const long long t_start = ctl->t_start;
  unsigned long long nodes = 0;
  bool finished = false;
  while (!finished) {
    int r = 0;
    if (lane == 0) r = atomicAdd(&ctl->next_root, 1);
    r = __shfl(r, 0, 64);
    if (r >= L || *v_abort) break;
    if (phase_b && r > *v_first) break;
    int thr = phase_b ? omega - 1 : *v_gbest;
    if (Kp[r] <= thr) continue;  // core + 1 bounds every clique through r
    u64 p[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int j = lane + 64 * i;
      u64 x = j < W ? adjP[(size_t)r * W + j] : 0ULL;
      if (j < (r >> 6)) x = 0;
      else if (j == (r >> 6)) x &= ~((2ULL << (r & 63)) - 1ULL);  // ranks above r only
      p[i] = x;
    }
    if (lane == 0) Cl[0] = r;
    int d = 0, size = 1;
    while (true) {
      ++nodes;
      if ((nodes & 63) == 0) {
        if (tick_limit > 0 && (long long)wall_clock64() - t_start > tick_limit) {
          if (lane == 0) ctl->abort = 1;
        }
        if (*v_abort) {
          finished = true;
          break;
        }
      }
      if (!phase_b) thr = *v_gbest;
      else if (r > *v_first) break;  // an earlier root already holds a clique of size omega
      const int cnt = ex_count<NW>(p);
      bool prune;
      if (cnt == 0) {
        prune = true;
        if (size > thr) {
          if (!phase_b) {
            if (lane == 0) atomicMax(&ctl->gbest, size);
          } else {
            __syncthreads();  // (one wave per workgroup) lane 0's writes to Cl are visible to every lane
            for (int i = lane; i < size; i += 64) mine[2 + i] = Cl[i];
            if (lane == 0) {
              mine[1] = size;
              mine[0] = r;
              __threadfence();
              atomicMin(&ctl->first_root, r);
            }
            finished = true;  // later roots cannot come first
            break;
          }
        }
      } else if (size + cnt <= thr) {
        prune = true;
      } else {
        prune = size + ex_colour_bound<NW>(p, thr - size, adjP, W, lane) <= thr;
      }
      if (!prune) {
        const int u = ex_highest<NW>(p, lane);
        ex_clear<NW>(p, u, lane);
#pragma unroll
        for (int i = 0; i < NW; ++i) stk[((size_t)d * NW + i) * 64 + lane] = p[i];
        u64 row[NW];
        ex_row<NW>(adjP, W, u, lane, row);
#pragma unroll
        for (int i = 0; i < NW; ++i) p[i] &= row[i];
        if (lane == 0) Cl[size] = u;
        ++size;
        ++d;
        continue;
      }
      if (d == 0) break;
      --d;
      --size;
#pragma unroll
      for (int i = 0; i < NW; ++i) p[i] = stk[((size_t)d * NW + i) * 64 + lane];
    }
  }
  if (lane == 0 && nodes) atomicAdd(&ctl->nodes, nodes);
}

// Small candidate sets (the ranks >= t0 fit 64 words and, with the per-level stack, 60 KB of LDS): adjacency rows and the
// stack live in LDS, and the top-level branches of a root are dealt round-robin to SLICES tasks — a root of a small
// dense graph is a whole search on its own, so roots alone leave the chip idle.  Task (root r, slice s) walks r's
// children in descending rank, expands those whose index i has i mod SLICES = s (the earlier children are removed from
// the candidate set exactly as the sequential search would have done), and in phase B reports the key (r << 16 | i):
// the lowest key is the first maximum clique of the depth-first order.
#define EX_SLICES 32
__global__ __launch_bounds__(64) void k_exact_search_small(const u64* __restrict__ adjP, const int* __restrict__ Kp, int L, int W,
                                                           ExactCtl* ctl, int depth_cap, int* __restrict__ cliq, int phase_b,
                                                           long long tick_limit, int t0_host) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ex_lds[];
  const int lane = threadIdx.x, wave = blockIdx.x;
  const int t0 = t0_host, w0 = t0 >> 6, Wq = W - w0, n = L - (w0 << 6);  // rows / bits are relative to rank w0 * 64
  if (ctl->t0 != t0_host) {  // the host sized the LDS from its copy of the search state: they must agree
    if (lane == 0) ctl->abort = 1;
    return;
  }
  u64* rows = (u64*)ex_lds;                         // [n][Wq]
  u64* stk = rows + (size_t)n * Wq;                 // [depth_cap][Wq]
  int* Cl = (int*)(stk + (size_t)depth_cap * Wq);   // [depth_cap + 2]
  for (int e = lane; e < n * Wq; e += 64) {
    const int v = e / Wq, j = e - v * Wq;
    rows[e] = adjP[(size_t)(v + (w0 << 6)) * W + w0 + j];
  }
  __syncthreads();
  int* mine = cliq + (size_t)wave * (depth_cap + 2);
  volatile int* v_gbest = &ctl->gbest;
  volatile int* v_first = &ctl->first_root;
  volatile int* v_abort = &ctl->abort;
  const int omega = ctl->gbest;
  const long long t_start = ctl->t_start;
  const int r_begin = phase_b ? ctl->next_root : t0;  // (next_root holds phase B's first root; tasks count from it)
  unsigned long long nodes = 0;
  bool finished = false;
  const int wl = lane < Wq ? lane : 0;
  const bool act = lane < Wq;
  auto count = [&](u64 x) { return wave_sum_i32(act ? __popcll(x) : 0); };
  auto highest = [&](u64 x) { return wave_max_i32((act && x) ? lane * 64 + 63 - __clzll((long long)x) : -1); };
  auto colour_bound = [&](u64 pp, int limit) {
    u64 u = pp;
    int colours = 0;
    while (__any(act && u != 0)) {
      if (++colours > limit) return limit + 1;
      u64 q = u;
      while (true) {
        const int v = highest(q);
        if (v < 0) break;
        if (lane == (v >> 6)) {
          q &= ~(1ULL << (v & 63));
          u &= ~(1ULL << (v & 63));
        }
        q &= ~rows[(size_t)v * Wq + wl];
      }
    }
    return colours;
  };
  while (!finished) {
    int task = 0;
    if (lane == 0) task = atomicAdd(&ctl->task, 1);
    task = __shfl(task, 0, 64);
    const int r = r_begin + task / EX_SLICES, slice = task % EX_SLICES;
    if (r >= L || *v_abort) break;
    if (phase_b && r > (*v_first >> 16)) break;
    int thr = phase_b ? omega - 1 : *v_gbest;
    if (Kp[r] <= thr) continue;
    const int rr = r - (w0 << 6);
    u64 p = act ? rows[(size_t)rr * Wq + lane] : 0ULL;
    if (lane < (rr >> 6)) p = 0;
    else if (lane == (rr >> 6)) p &= ~((2ULL << (rr & 63)) - 1ULL);
    if (lane == 0) Cl[0] = r;
    int d = 0, size = 1, top_i = 0, cur_i = 0;
    bool root_done = false;
    while (!root_done) {
      ++nodes;
      if ((nodes & 63) == 0) {
        if (tick_limit > 0 && (long long)wall_clock64() - t_start > tick_limit && lane == 0) ctl->abort = 1;
        if (*v_abort) {
          finished = true;
          break;
        }
      }
      if (!phase_b) thr = *v_gbest;
      else if (((r << 16) | cur_i) > *v_first) break;
      bool prune = false;
      int u = -1;
      if (d == 0) {
        // next child of the root that belongs to this slice
        while (true) {
          const int cnt = count(p);
          if (cnt == 0 || 1 + cnt <= thr) {
            root_done = true;
            break;
          }
          u = highest(p);
          const int i = top_i++;
          if (i % EX_SLICES == slice) {
            cur_i = i;
            break;
          }
          if (lane == (u >> 6)) p &= ~(1ULL << (u & 63));
        }
        if (root_done) break;
        if (phase_b && ((r << 16) | cur_i) > *v_first) break;
        if (1 + colour_bound(p, thr - 1) <= thr) break;  // prunes this child and every later one
      } else {
        const int cnt = count(p);
        if (cnt == 0) {
          prune = true;
          if (size > thr) {
            if (!phase_b) {
              if (lane == 0) atomicMax(&ctl->gbest, size);
            } else {
              __syncthreads();
              for (int i = lane; i < size; i += 64) mine[2 + i] = Cl[i];
              if (lane == 0) {
                mine[1] = size;
                mine[0] = (r << 16) | cur_i;
                __threadfence();
                atomicMin(&ctl->first_root, (r << 16) | cur_i);
              }
              finished = true;
              break;
            }
          }
        } else if (size + cnt <= thr) {
          prune = true;
        } else {
          prune = size + colour_bound(p, thr - size) <= thr;
        }
        if (!prune) u = highest(p);
      }
      if (!prune) {
        const int uu = u;  // relative to rank w0 * 64, like every bit index here
        if (lane == (uu >> 6)) p &= ~(1ULL << (uu & 63));
        if (act) stk[(size_t)d * Wq + lane] = p;
        p &= rows[(size_t)uu * Wq + wl];
        if (!act) p = 0;
        if (lane == 0) Cl[size] = uu + (w0 << 6);
        ++size;
        ++d;
        continue;
      }
      if (d == 0) break;
      --d;
      --size;
      p = act ? stk[(size_t)d * Wq + lane] : 0ULL;
    }
  }
  if (lane == 0 && nodes) atomicAdd(&ctl->nodes, nodes);
}

// the clique of the lowest successful root becomes the search result of solver.hip's state (best_r + picks)
__global__ __launch_bounds__(256) void k_exact_commit(ExactCtl* ctl, const int* __restrict__ cliq, int nwaves, int depth_cap,
                                                      SolverState* st, int* __restrict__ picks) {
  __shared__ int s_w;
  const int first = ctl->first_root;
  if (first == INT_MAX) return;
  if (threadIdx.x == 0) s_w = -1;
  __syncthreads();
  for (int w = threadIdx.x; w < nwaves; w += 256)
    if (cliq[(size_t)w * (depth_cap + 2)] == first) s_w = w;  // exactly one wave ran this root
  __syncthreads();
  const int w = s_w;
  if (w < 0) return;
  const int* mine = cliq + (size_t)w * (depth_cap + 2);
  const int size = mine[1];
  for (int i = threadIdx.x; i < size - 1; i += 256) picks[i] = mine[2 + 1 + i];
  if (threadIdx.x == 0) {
    st->mc = size;
    st->best_r = mine[2];
    ctl->winner = w;
  }
}

// LDS bytes of the small-graph kernel (0: does not apply)
static size_t exact_small_lds(int L, int W, int t0, int depth_cap) {
  const int w0 = t0 >> 6, Wq = W - w0, n = L - (w0 << 6);
  if (Wq < 1 || Wq > 64) return 0;
  const size_t bytes = ((size_t)n * Wq + (size_t)depth_cap * Wq) * 8 + (size_t)(depth_cap + 2) * 4;
  return bytes <= (size_t)60 * 1024 ? bytes : 0;
}

static int exact_nw(int W) { return W <= 64 ? 1 : W <= 128 ? 2 : W <= 256 ? 4 : W <= 512 ? 8 : 0; }

// bytes of scratch for `nwaves` search waves
static size_t exact_scratch_bytes(int W, int depth_cap, int nwaves) {
  const int nw = exact_nw(W);
  return 256 + (size_t)nwaves * (depth_cap + 2) * sizeof(int) + 256 + (size_t)nwaves * depth_cap * nw * 64 * sizeof(u64);
}

struct ExactBufs {
  ExactCtl* ctl = nullptr;
  int* cliq = nullptr;
  u64* stack = nullptr;
};
static void exact_carve(ExactBufs& E, void* base, int depth_cap, int nwaves) {
  char* p = (char*)base;
  E.ctl = (ExactCtl*)p;
  p += 256;
  E.cliq = (int*)p;
  p += (((size_t)nwaves * (depth_cap + 2) * sizeof(int)) + 255) & ~(size_t)255;
  E.stack = (u64*)p;
}

static void exact_launch_search(const SolverBufs& B, const ExactBufs& E, int L, int depth_cap, int nwaves, int phase_b,
                                long long tick_limit, hipStream_t st, int t0_host, size_t small_lds) {
  const int W = (L + 63) / 64;
  if (small_lds) {
    hipLaunchKernelGGL(k_exact_search_small, dim3(nwaves), dim3(64), small_lds, st, B.adjP, B.Kp, L, W, E.ctl, depth_cap,
                       E.cliq, phase_b, tick_limit, t0_host);
    return;
  }
  const size_t lds = (size_t)(depth_cap + 2) * sizeof(int);
#define EX_LAUNCH(NWV)                                                                                              \
  hipLaunchKernelGGL(k_exact_search<NWV>, dim3(nwaves), dim3(64), lds, st, B.adjP, B.Kp, L, W, E.ctl, E.stack, depth_cap, \
                     E.cliq, phase_b, tick_limit)
  switch (exact_nw(W)) {
    case 1: EX_LAUNCH(1); break;
    case 2: EX_LAUNCH(2); break;
    case 4: EX_LAUNCH(4); break;
    default: EX_LAUNCH(8); break;
  }
#undef EX_LAUNCH
}

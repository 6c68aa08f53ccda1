// solver.hip — back-end kernels: pairwise-consistency bit-matrix, k-core, max-clique heuristic,
// GNC-TLS yaw rotation and component-wise translation (COTE).  gfx950 / wave64 only.
//
// Replaces Quatro::computeTransformation (reference include/quatro.hpp:769-936) and everything it
// reaches: computeTIMs (:307-344), solveForScale (:355-386), teaser::Graph::addEdge
// (include/teaser/graph.h:96-104), teaser::MaxCliqueSolver::findMaxClique + PMC (src/graph.cc:12-104),
// solveForRotation2D (:430-572) + svdRot2d (include/teaser/utils.h:151-166), solveForTranslation /
// estimate (:585-747).  No TIM is ever materialised: the L x L predicate is evaluated tile-wise and
// ballot-packed into a bit matrix (L^2/8 bytes instead of the reference's ~32.5 L^2 bytes).
#include <type_traits>
#include <vector>

#include "common.h"
#include "solver.h"

// =================================================================================================
// K9+K10+K11: consistency graph, 64 x 64 tiles of the UPPER triangle, one per workgroup (its four waves take 16 rows
// each).  The 64 points of the row block are staged in LDS; a lane holds ITS column's point pair in
// registers, the 64 row points are broadcast from LDS one after the other, the 64 predicate bits of a row are packed
// with __ballot (-> word `cb` of row i) and every lane collects its own column's bits over the 64 rows (-> word `rb` of
// row j: the transposed tile, so the lower triangle is never evaluated).  Degrees: every (64-column block, vertex) pair is
// one word of the matrix and has exactly one writer, which leaves that word's popcount as a byte in degp; the kernels that
// need degrees add a row's bytes up (solver_degree) — no atomics (device-scope atomics on this 8-XCD part cost the first
// version of this kernel more than the tiles themselves), nothing to zero.
//
// Predicate (reference :372-385): |b/a - 1| <= beta/a  AND  |a/b - 1| <= beta/b, a=|src TIM|, b=|tgt TIM|.
// In exact arithmetic both sides are |a - b| <= beta, i.e. with s = a^2, t = b^2 and s + t > beta^2:
//     (s - t)^2 <= 2 beta^2 (s + t) - beta^4.
// The tile loop SCREENS in binary32 — source and target ride in the two halves of packed registers (v_pk_add / v_pk_mul /
// v_pk_fma: 6 instructions for both squared lengths) — and decides a pair only when the two sides differ by more than
// `margin` relative (a rigorous bound on the binary32 error, see graph_margin(): ~4e-4 for beta = 0.6, which puts
// ~1 pair in 10^5 inside the band); everything else — the band, s + t <= beta^2 (short TIMs, zero-length TIMs with
// their inf/NaN in the reference expression), s + t > 65536, non-finite input — is decided by pair_consistent(), the
// reference expression in binary64 evaluated verbatim.  The result is bit-identical to evaluating the reference
// expression everywhere (the margin is orders of magnitude wider than its own rounding).
__device__ __forceinline__ bool pair_consistent(double s, double t, double beta, double beta2) {
  if (s > 0.0 && t > 0.0) {
    const double u = s + t - beta2;
    if (u > 0.0) {
      const double lhs = u * u, rhs = 4.0 * s * t;
      if (lhs > rhs * (1.0 + 1e-9)) return false;
      if (lhs < rhs * (1.0 - 1e-9)) return true;
    }
  }
  const double a = sqrt(s), b = sqrt(t);
  const bool fwd = fabs(b / a - 1.0) <= beta * (1.0 / a);
  const bool rev = fabs(a / b - 1.0) <= beta * (1.0 / b);
  return fwd && rev;
}
// degree of vertex v: the sum of its row's per-block popcounts (see SolverView::degp)
__device__ __forceinline__ int solver_degree(const SolverView& V, int v) {
  if (!V.degp) return V.deg[v];
  const unsigned char* __restrict__ p = V.degp + v;
  const int nb = (V.L + 63) >> 6, Lp = V.Lp;
  int d = 0;
  int k = 0;
  for (; k + 4 <= nb; k += 4) {
    const int a = p[(size_t)k * Lp], b = p[(size_t)(k + 1) * Lp], c = p[(size_t)(k + 2) * Lp], e = p[(size_t)(k + 3) * Lp];
    d += (a + b) + (c + e);
  }
  for (; k < nb; ++k) d += p[(size_t)k * Lp];
  return d;
}
#define GB_SMAX 65536.0f  // s + t above this (TIMs longer than ~180 m) are left to the binary64 path
// Relative half-width of the band the binary32 screen leaves undecided.  With u = 2^-24: s and t carry a relative error
// <= 5u (difference 1u, square 2u, three fused accumulations), so D = fl(s - t) is off by <= 5u (s + t) + u |D| and, where
// the decision is close (D^2 ~ G = 2 beta^2 (s+t) - beta^4 >= beta^2 (s+t)), D^2 by <= 2 sqrt(2) beta 5u (s+t)^1.5, i.e.
// <= 2 sqrt(2) 5u sqrt(s+t) / beta relative to G; G itself is off by <= 18u G and the final subtraction by u.  Three
// times that bound is used.  A margin above 0.05 (beta below ~0.01) switches the screen off (margin = +inf: until round 4
// it was -1, which made |z| > margin G hold for every pair and left the binary32 screen to decide ALL of them).
static float graph_margin(double beta) {
  const double u = 1.0 / 16777216.0;
  const double m = 3.0 * (2.0 * sqrt(2.0) * 5.0 * u * sqrt((double)GB_SMAX) / beta + 20.0 * u);
  return (m > 0.05 || !(beta > 0.0)) ? INFINITY : (float)m;  // (inf: |z| > inf G never holds — every pair goes to binary64)
}
typedef float gb_f2 __attribute__((ext_vector_type(2)));

// control words of k_hcore_async (below) in V.perm, cleared here when the graph is built for it
#define HCA_CTL_FAILED 2   // ints of V.perm: [2] failed, [3] iterations of the slowest workgroup,
#define HCA_CTL_ITERS 3    // [HCA_CTL_VER + w] version counters, [HCA_CTL_DONE + w] marks  (w < HCA_MAXWG)
#define HCA_CTL_FLOOR 4    // [4] (floor << 1) | decided: values below the floor are not lowered any further,
#define HCA_CTL_FROZE 5    // [5] some row was actually left alone for that reason
#define HCA_CTL_FLOOR2 6   // [6] (floor << 1) | 1 from the scout workgroup (a clique it FOUND, see hca_scout): may arrive at any time
#define HCA_FLOOR_MIN 8
#define HCA_MAXWG 512      // (one workgroup per compute unit: 256 on this part)
#define HCA_CTL_VER 64
#define HCA_CTL_DONE (64 + HCA_MAXWG)
// 64 x 64 bit-matrix transpose inside a wavefront: lane j holds word j (lo, hi); afterwards lane r holds the word made
// of bit r of every lane's word.  Six independent swaps "lane-index bit s <-> bit-index bit s": the 32-step exchanges
// the two halves between lanes 32 apart, the others rotate the partner's word by s and merge under a per-lane mask.
__device__ __forceinline__ void wave_transpose64(u32& lo, u32& hi, int lane) {
  {
    const bool low = lane < 32;
    const u32 send = low ? hi : lo;
    const u32 recv = (u32)__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, (int)send);
    const u32 nlo = low ? lo : recv, nhi = low ? recv : hi;
    lo = nlo;
    hi = nhi;
  }
#pragma unroll
  for (int s = 16; s >= 1; s >>= 1) {
    const u32 M0 = s == 16 ? 0x0000FFFFu : s == 8 ? 0x00FF00FFu : s == 4 ? 0x0F0F0F0Fu : s == 2 ? 0x33333333u : 0x55555555u;
    const bool up = (lane & s) != 0;
    const u32 keep = up ? ~M0 : M0;
    const int sh = up ? s : 32 - s;  // rotate right by s (upper lane of the pair) or left by s
    const int addr = (lane ^ s) << 2;
    u32 ylo = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)lo), yhi = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)hi);
    ylo = __builtin_amdgcn_alignbit(ylo, ylo, sh);
    yhi = __builtin_amdgcn_alignbit(yhi, yhi, sh);
    lo = (lo & keep) | (ylo & ~keep);
    hi = (hi & keep) | (yhi & ~keep);
  }
}

// K9+K10+K11 as 64-row x 256-column STRIPS of the upper triangle (rounds 4-5: the kernel above GB_TILES_MAX_L correspondences;
// since round 6 — k_graph_build_mfma below — the test build's comparison engine), one workgroup of four waves each; wave ct owns
// the 64 x 64 tile (rb, 4 C + ct).  What changed against the tile-per-workgroup kernel (k_graph_build_tiles, kept as a
// comparison engine) and why (tests/probe/issue_probe.hip and profiles/r4_graph_sq.txt have the numbers — the kernel is
// bound by VECTOR INSTRUCTIONS: a SIMD retires one wave64 instruction in four clocks, the tile kernel spent 24 per row of
// 64 predicates and kept the SIMDs 100 % busy with them):
//   * NO value travels through a scalar register in the row loop.  The tile kernel did four compares, a ballot AND and two
//     v_writelane per row; a v_cmp whose mask is consumed by a scalar instruction or as the carry-in of a v_addc also costs
//     the issuing wave ~28 clocks (the result has to come back from the scalar register file).  Here the decision z < 0
//     IS the sign bit of z and is shifted into the lane's column word by one v_alignbit ({cacc, z} >> 31); "the screen is
//     sure" is the sign bit of margin c1 (s + t) - |z|, collected the same way; the range test on s + t is an unsigned
//     min3 / max3 over the bit patterns (NaN and inf sort above every finite value).  Eight rows go by without a branch;
//     per 32 rows ONE vote asks whether any lane has a pair the screen could not decide, and those lanes walk their
//     undecided rows through the binary64 expression;
//   * the packed binary32 arithmetic pairs TWO ROWS (one register half each) instead of (source, target), so the
//     operations behind the two squared lengths (S, D, c2 - c1 S, z) are packed as well: 8 packed + 4 plain vector
//     instructions per row of 64 predicates;
//   * the row words are not collected at all: they come out of a 64 x 64 bit transpose of the column words
//     (wave_transpose64), and the four tiles' words of a row leave as ONE 32-byte store (a whole memory sector — rows
//     start 32-byte aligned: the bit matrix's row stride Wb is a multiple of four words).  The transposed words are still
//     8-byte stores, one row each: WRITE_SIZE 2.5x the matrix instead of 4x.  (256 x 256 super-tiles, sixteen tiles per
//     workgroup, store whole sectors in both directions — built and measured in three shapes, 16 waves x 1 tile, 16 x 2
//     half tiles, 8 x 2 tiles: 118 / 172 / 114 us at L = 20000 against 132 for the tile kernel, and 19 - 22 us against
//     17 at L = 5000: workgroups that long leave the unit idle while they load, meet at barriers and exchange lanes —
//     45 % of their wave-cycles were parked.  The 8 x 2 form is kept in tests/probe/k_graph_build_supertile.txt.)
// The screen and the binary64 decision are the tile kernel's (the "sure" threshold is margin c1 (s + t) instead of
// margin (c1 (s + t) - c2): a little wider, never narrower): the bit matrix is identical.
#define GB2_THREADS 256
#define GB_TILES_MAX_L 8192  // up to here the tile kernel (k_graph_build_tiles, below) builds the graph
template <bool EXT>
__global__ __launch_bounds__(GB2_THREADS, EXT ? 7 : 8) void k_graph_build(ViewExt<SolverView> x, SolverView one, double beta,
                                                                float margin, int prep) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int L = V.L, Wb = V.Wb;
  const int nb = (L + 63) >> 6;
  if (prep) {  // chores of neighbouring launches, see k_graph_build_tiles
    const int nthreads = gridDim.x * gridDim.y * GB2_THREADS;
    const int gi = (blockIdx.y * gridDim.x + blockIdx.x) * GB2_THREADS + threadIdx.x;
    if (prep & 1) {
      for (int e = gi; e < (((L + 63) & ~63) >> 1); e += nthreads) ((unsigned*)V.Kp)[e] = 0xffffffffu;
      for (int e = gi; e < HCA_CTL_DONE + HCA_MAXWG; e += nthreads) V.perm[e] = 0;
    }
    if ((prep & 2) && gi < (int)(sizeof(SolverState) / 4)) ((int*)V.st)[gi] = 0;
  }
  const int rb = blockIdx.y, C = blockIdx.x;
  if (rb >= nb || 4 * C + 3 < rb || 4 * C >= nb) return;  // strips of the upper triangle only
  const float4* __restrict__ src = V.src;
  const float4* __restrict__ tgt = V.tgt;
  u64* __restrict__ bm = V.bm;
  unsigned char* __restrict__ degp = const_cast<unsigned char*>(V.degp);
  const int Lp = V.Lp;
  // the 64 row points, two rows to a record of twelve floats: (sx, sx'), (sy, sy'), (sz, sz'), (tx, tx'), (ty, ty'),
  // (tz, tz') — three 16-byte broadcast reads per pair of rows
  __shared__ __attribute__((aligned(16))) float rowpts[32 * 12];
  __shared__ __attribute__((aligned(16))) u64 rowbuf[64 * 4];  // [row][column tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int ct = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = 4 * C + ct;
  if (tid < 64) {
    const int i = min(rb * 64 + tid, L - 1);
    const float4 a = src[i], b = tgt[i];
    float* rp = rowpts + (tid >> 1) * 12 + (tid & 1);
    rp[0] = a.x;
    rp[2] = a.y;
    rp[4] = a.z;
    rp[6] = b.x;
    rp[8] = b.y;
    rp[10] = b.z;
  }
  const int j = cb * 64 + lane;
  const bool mine = cb < nb && cb >= rb;  // my tile exists and lies in the upper triangle
  const bool jvalid = mine && j < L;
  float csx, csy, csz, ctx, cty, ctz;
  {
    const int jj = max(0, min(j, L - 1));
    const float4 a = src[jj], b = tgt[jj];
    csx = a.x;
    csy = a.y;
    csz = a.z;
    ctx = b.x;
    cty = b.y;
    ctz = b.z;
  }
  __syncthreads();
  const int nrows = min(64, L - rb * 64);  // rows of the strip that exist
  if (mine) {
    const bool diag = rb == cb;
    const float fb2 = (float)(beta * beta);
    const float c1 = 2.0f * fb2, c2 = fb2 * fb2;
    // s + t in (smin, GB_SMAX): above smin the squared form is valid (s + t - beta^2 > 0 in exact arithmetic too); as
    // unsigned integers the bit patterns of non-negative floats order like the floats, NaN / inf / anything negative above
    const u32 smin_bits = __float_as_uint(fb2 * 1.01f), smax_bits = __float_as_uint(GB_SMAX);
    const double beta2 = beta * beta;
    const float mc1 = margin * c1;  // (inf when the screen is switched off)
    u32 acc[2] = {0, 0};
    // `DG` (a tile on the diagonal of the matrix): a lane meets its own vertex in one row (s = t = 0), which the range
    // test would take for a short TIM; there the own row's s + t is replaced by 1 for the test (its bit is cleared below)
    auto sweep = [&](auto dg_tag) __attribute__((always_inline)) {
      constexpr bool DG = decltype(dg_tag)::value;
      const gb_f2 sx2 = {csx, csx}, sy2 = {csy, csy}, sz2 = {csz, csz}, tx2 = {ctx, ctx}, ty2 = {cty, cty}, tz2 = {ctz, ctz};
      const gb_f2 nc1_2 = {-c1, -c1}, c2_2 = {c2, c2};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        u32 cacc = 0;  // decision bits of my column, one row per shift (first row ends up in bit 31)
        u32 sacc = 0;  // "the screen is sure" bits, same order
        u32 oacc = 0;  // one bit per group of eight rows: s + t out of range somewhere in the group
        const u32 selfw = DG ? ((lane >> 5) == half ? (1u << (lane & 31)) : 0u) : 0u;  // my own row within this half
#pragma unroll 1
        for (int g8 = 0; g8 < 4; ++g8) {
          u32 s_lo = 0xffffffffu, s_hi = 0u;
#pragma unroll 2
          for (int k = 0; k < 4; ++k) {
            const float4* rp = (const float4*)(rowpts + (half * 16 + g8 * 4 + k) * 12);
            const float4 p0 = rp[0], p1 = rp[1], p2 = rp[2];
            const gb_f2 dxs = sx2 - gb_f2{p0.x, p0.y}, dys = sy2 - gb_f2{p0.z, p0.w}, dzs = sz2 - gb_f2{p1.x, p1.y};
            const gb_f2 dxt = tx2 - gb_f2{p1.z, p1.w}, dyt = ty2 - gb_f2{p2.x, p2.y}, dzt = tz2 - gb_f2{p2.z, p2.w};
            gb_f2 qs = dxs * dxs;
            qs = __builtin_elementwise_fma(dys, dys, qs);
            qs = __builtin_elementwise_fma(dzs, dzs, qs);  // squared source TIM lengths of the two rows
            gb_f2 qt = dxt * dxt;
            qt = __builtin_elementwise_fma(dyt, dyt, qt);
            qt = __builtin_elementwise_fma(dzt, dzt, qt);
            const gb_f2 S = qs + qt, D = qs - qt;
            const gb_f2 nG = __builtin_elementwise_fma(S, nc1_2, c2_2);  // -(c1 S - c2)
            const gb_f2 z = __builtin_elementwise_fma(D, D, nG);          // < 0: consistent
            // < 0: the screen is sure (|z| above margin c1 S >= margin G)
            const float t0 = __builtin_fmaf(S.x, mc1, -__builtin_fabsf(z.x)), t1 = __builtin_fmaf(S.y, mc1, -__builtin_fabsf(z.y));
            cacc = __builtin_amdgcn_alignbit(cacc, __float_as_uint(z.x), 31);
            cacc = __builtin_amdgcn_alignbit(cacc, __float_as_uint(z.y), 31);
            sacc = __builtin_amdgcn_alignbit(sacc, __float_as_uint(t0), 31);
            sacc = __builtin_amdgcn_alignbit(sacc, __float_as_uint(t1), 31);
            u32 b0 = __float_as_uint(S.x), b1 = __float_as_uint(S.y);
            if (DG) {
              const int rr = g8 * 8 + 2 * k;  // row within the half
              b0 |= (0u - ((selfw >> rr) & 1u)) & 0x3f800000u;
              b1 |= (0u - ((selfw >> (rr + 1)) & 1u)) & 0x3f800000u;
            }
            s_lo = min(min(b0, b1), s_lo);
            s_hi = max(max(b0, b1), s_hi);
          }
          oacc = (oacc << 1) | ((s_lo > smin_bits && s_hi < smax_bits) ? 0u : 1u);
        }
        // one vote per 32 rows: does any lane hold a pair the screen could not decide?
        u32 pend = ~sacc;  // bit 31 - i: row i of this half is undecided
#pragma unroll
        for (int g8 = 0; g8 < 4; ++g8) pend |= ((oacc >> (3 - g8)) & 1u) ? (0xff000000u >> (8 * g8)) : 0u;
        {
          // rows past the end of the last block and a vertex paired with itself are never decided (cleared below)
          const int live = nrows - half * 32;  // rows of this half that exist
          const u32 rowmask = live >= 32 ? 0xffffffffu : live <= 0 ? 0u : ~(0xffffffffu >> live);
          pend &= rowmask;
          if (DG) pend &= ~__brev(selfw);
          if (!jvalid) pend = 0;
        }
        while (__any(pend != 0)) {  // rare: the band is ~1e-5 of the pairs, plus TIMs shorter than beta
          if (pend != 0) {
            const int i = __clz(pend);  // my first undecided row of this half
            const int r = half * 32 + i;
            const float* rp = rowpts + (r >> 1) * 12 + (r & 1);
            const double ex_ = (double)csx - (double)rp[0], ey = (double)csy - (double)rp[2], ez = (double)csz - (double)rp[4];
            const double fx = (double)ctx - (double)rp[6], fy = (double)cty - (double)rp[8], fz = (double)ctz - (double)rp[10];
            const double s = ex_ * ex_ + (ey * ey + ez * ez);
            const double t = fx * fx + (fy * fy + fz * fz);
            const u32 bit = 0x80000000u >> i;
            cacc = pair_consistent(s, t, beta, beta2) ? (cacc | bit) : (cacc & ~bit);
            pend &= ~bit;
          }
        }
        acc[half] = cacc;
      }
    };
    if (diag) sweep(std::true_type{});
    else sweep(std::false_type{});
    // 32 shifts put the first row of a half into bit 31
    u64 colw = ((u64)__brev(acc[1]) << 32) | (u64)__brev(acc[0]);
    if (nrows < 64) colw &= (1ULL << nrows) - 1ULL;  // rows past the end of the last block
    if (diag) colw &= ~(1ULL << lane);                // a vertex is not its own neighbour
    if (!jvalid) colw = 0;                             // columns past the end of the matrix
    if (!diag && jvalid) {  // the transposed word (the lower triangle is never evaluated)
      bm[(size_t)j * Wb + rb] = colw;
      degp[(size_t)rb * Lp + j] = (unsigned char)__popcll(colw);
    }
    u32 lo = (u32)colw, hi = (u32)(colw >> 32);
    wave_transpose64(lo, hi, lane);
    rowbuf[lane * 4 + ct] = ((u64)hi << 32) | lo;  // lane r: the word of row rb * 64 + r (diagonal tile: both triangles)
  }
  __syncthreads();
  // the four tiles' words of a row as one 32-byte store (one memory sector): thread t -> row t / 4, word t % 4; and the
  // word's popcount as the (64-column block, vertex) degree byte — every such pair has exactly one writer
  {
    const int row = tid >> 2, wi = 4 * C + (tid & 3);
    if (row < nrows && wi < nb && wi >= rb) {
      const u64 w = rowbuf[tid];
      bm[(size_t)(rb * 64 + row) * Wb + wi] = w;
      degp[(size_t)wi * Lp + rb * 64 + row] = (unsigned char)__popcll(w);
    }
  }
}

// K9+K10+K11 on the MATRIX PIPE (round 6; VERDICT rounds 3-5: "screen the predicate on the matrix pipe").  Same strips, same
// outputs, same binary64 decision for whatever the screen leaves open — but the two squared TIM lengths of a 32 x 32 block of
// pairs come out of two v_mfma_f32_32x32x16_f16 instead of 12 packed vector instructions per row of 64:
//     R = p . p' - |p|^2 / 2 - |p'|^2 / 2 = -|p - p'|^2 / 2
// with every coordinate (relative to correspondence 0: a translation changes no distance) split in two binary16 halves,
// x = x1 + x2 + eps, |eps| <= 2^-22 |x| (4 u M in R), and a = -|p|^2 / 2 likewise (2 u M): K = 16 exactly —
//     row operand    [x1 x1 x2 x2 | y1 y1 y2 y2 | z1 z1 z2 z2 | a1 a2 1 1]
//     column operand [x1 x2 x1 x2 | y1 y2 y1 y2 | z1 z2 z1 z2 | 1 1 a1 a2]
// (a binary16 x binary16 product is exact in binary32; the accumulation is the matrix unit's).  Error of R against the true
// -|P - P'|^2 / 2 of the given binary32 points, with M = |p|^2 + |p'|^2 and u = 2^-24: the halves' truncation (above),
// |p|^2 rounded to binary32 after three roundings <= 1.5 u M, the origin subtraction
// (|p - p'|^2 of the shifted binary32 points against |P - P'|^2 of the given ones) <= 3.5 u M, the unit's accumulation <= 16 u
// sum |terms| <= 16.1 u M (the budget match.hip measured and uses: <= 5.5 u there) — 27.1 u M = 1.62e-6 M;
// delta = 1.8e-6 (M_s + M_t) + 2e-6 bounds the error of BOTH D = R_s - R_t = (t - s) / 2 and P = -(R_s + R_t) =
// (s + t) / 2, M_s / M_t the largest row norm + the largest column norm of the tile.  The predicate (s - t)^2 <= 2 beta^2
// (s + t) - beta^4 is z = D^2 - beta^2 P + beta^4 / 4 <= 0, and z is off by <= 2 |D| delta + delta^2 + beta^2 delta + its own
// rounding (<= 4u D^2 near the threshold: <= 0.07 x 2 |D| delta, since |D| <= M_s + M_t); a pair is decided by the screen
// only when |z| > 2.4 |D| delta + 1.3 (beta^2 delta + delta^2) + u beta^4 + 1e-7 and P > 0.505 beta^2 + delta (s + t > 1.01 beta^2: the
// squared form is valid, as in the other two kernels); the rest — a band ~1e-4 of the pairs wide at +-50 m, TIMs shorter
// than beta — and every pair of a tile whose norms leave binary16's range (or are not finite) gets pair_consistent(), the
// reference expression in binary64: the bit matrix is identical to the other two kernels' (tests/gpu_graph_bench.py).
// Per entry the vector unit still does: P, D, -beta^2 P + beta^4 / 4, z (four packed instructions per two entries), the
// margin (one fma), |z| - margin (one), two sign-bit shifts, half a min3 for the short-TIM test: 6.5 — against 12 + 5.
#define GBM_KDELTA 1.8e-6f
#define GBM_MIN_L 2048  // below: the tile kernel (one wave of short workgroups)
typedef _Float16 gbm_h8 __attribute__((ext_vector_type(8)));
typedef float gbm_f16x __attribute__((ext_vector_type(16)));
struct GbmRec {
  gbm_h8 lo, hi;  // K slots 0..7, 8..15
};
__device__ __forceinline__ void gbm_records(float x, float y, float z, GbmRec& row, GbmRec& colr, float& n_out) {
  const _Float16 x1 = (_Float16)x, y1 = (_Float16)y, z1 = (_Float16)z;
  const _Float16 x2 = (_Float16)(x - (float)x1), y2 = (_Float16)(y - (float)y1), z2 = (_Float16)(z - (float)z1);
  const float n = x * x + (y * y + z * z);
  const float a = -0.5f * n;
  const _Float16 a1 = (_Float16)a, a2 = (_Float16)(a - (float)a1);
  const _Float16 one = (_Float16)1.0f;
  row.lo = gbm_h8{x1, x1, x2, x2, y1, y1, y2, y2};
  row.hi = gbm_h8{z1, z1, z2, z2, a1, a2, one, one};
  colr.lo = gbm_h8{x1, x2, x1, x2, y1, y2, y1, y2};
  colr.hi = gbm_h8{z1, z2, z1, z2, one, one, a1, a2};
  n_out = n;
}
// (the largest of the wave's squared norms: non-negative floats order like their bit patterns; a NaN's pattern — sign bit
// cleared — lies above infinity's and comes out as the maximum, which then fails every "< limit" test)
__device__ __forceinline__ float gbm_wave_max(float v) {
  return __int_as_float(wave_max_i32(__float_as_int(__builtin_fabsf(v))));
}
template <bool EXT>
__global__ __launch_bounds__(GB2_THREADS, 4) void k_graph_build_mfma(ViewExt<SolverView> x, SolverView one, double beta, int prep) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int L = V.L, Wb = V.Wb;
  const int nb = (L + 63) >> 6;
  if (prep) {  // chores of neighbouring launches, see k_graph_build_tiles
    const int nthreads = gridDim.x * gridDim.y * GB2_THREADS;
    const int gi = (blockIdx.y * gridDim.x + blockIdx.x) * GB2_THREADS + threadIdx.x;
    if (prep & 1) {
      for (int e = gi; e < (((L + 63) & ~63) >> 1); e += nthreads) ((unsigned*)V.Kp)[e] = 0xffffffffu;
      for (int e = gi; e < HCA_CTL_DONE + HCA_MAXWG; e += nthreads) V.perm[e] = 0;
    }
    if ((prep & 2) && gi < (int)(sizeof(SolverState) / 4)) ((int*)V.st)[gi] = 0;
  }
  const int rb = blockIdx.y, C = blockIdx.x;
  if (rb >= nb || 4 * C + 3 < rb || 4 * C >= nb) return;  // strips of the upper triangle only
  const float4* __restrict__ src = V.src;
  const float4* __restrict__ tgt = V.tgt;
  u64* __restrict__ bm = V.bm;
  unsigned char* __restrict__ degp = const_cast<unsigned char*>(V.degp);
  const int Lp = V.Lp;
  __shared__ __attribute__((aligned(16))) GbmRec s_rowrec[2][64];   // role "row" of the strip's 64 rows: [cloud][row]
  __shared__ __attribute__((aligned(16))) GbmRec s_colrec[4][2][64];  // role "column" of every wave's 64 columns
  __shared__ float s_rowpts[64][6];  // the rows' binary32 points, for the binary64 path
  __shared__ float s_colpts[4][64][6];  // ... and every wave's columns'
  __shared__ float s_rowmax[2];      // the largest squared norm (relative to the origin) among the rows, per cloud
  __shared__ __attribute__((aligned(16))) u64 rowbuf[64 * 4];  // [row][column tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int ct = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = 4 * C + ct;
  const float4 o_s = src[0], o_t = tgt[0];  // the origin: correspondence 0
  if (tid < 128) {
    const int cl = tid >> 6, r = tid & 63;
    const int i = min(rb * 64 + r, L - 1);
    const float4 p = cl ? tgt[i] : src[i];
    const float4 o = cl ? o_t : o_s;
    GbmRec rr, cc;
    float n;
    gbm_records(p.x - o.x, p.y - o.y, p.z - o.z, rr, cc, n);
    s_rowrec[cl][r] = rr;
    const float nmax = gbm_wave_max(n);  // (waves 0 and 1: one cloud each)
    if (r == 0) s_rowmax[cl] = nmax;
    s_rowpts[r][3 * cl + 0] = p.x;
    s_rowpts[r][3 * cl + 1] = p.y;
    s_rowpts[r][3 * cl + 2] = p.z;
  }
  const int j = cb * 64 + lane;
  const bool mine = cb < nb && cb >= rb;  // my tile exists and lies in the upper triangle
  const bool jvalid = mine && j < L;
  float ncol_s, ncol_t;
  {
    const int jj = max(0, min(j, L - 1));
    const float4 a = src[jj], b = tgt[jj];
    GbmRec rr, cs, ctg;
    gbm_records(a.x - o_s.x, a.y - o_s.y, a.z - o_s.z, rr, cs, ncol_s);
    gbm_records(b.x - o_t.x, b.y - o_t.y, b.z - o_t.z, rr, ctg, ncol_t);
    s_colrec[ct][0][lane] = cs;
    s_colrec[ct][1][lane] = ctg;
    float* cp = s_colpts[ct][lane];
    cp[0] = a.x;
    cp[1] = a.y;
    cp[2] = a.z;
    cp[3] = b.x;
    cp[4] = b.y;
    cp[5] = b.z;
  }
  __syncthreads();
  const int nrows = min(64, L - rb * 64);  // rows of the strip that exist
  if (mine) {
    const bool diag = rb == cb;
    const int half = lane >> 5, c32 = lane & 31;
    const float fb2 = (float)(beta * beta);
    const double beta2 = beta * beta;
    // the tile's error bound, and whether its norms stay inside binary16's range (else: nothing is decided by the screen)
    const float ms = s_rowmax[0] + gbm_wave_max(ncol_s), mt = s_rowmax[1] + gbm_wave_max(ncol_t);
    const bool safe = (ms < 1.0e5f) && (mt < 1.0e5f) && (beta > 0.01);  // (NaN compares false)
    const float delta = GBM_KDELTA * (ms + mt) + 2e-6f;
    const float nb2 = -fb2, q4 = 0.25f * fb2 * fb2;
    // (the margin: 2 |D| delta + beta^2 delta + delta^2 is z's error from D and P; on top of it the binary32 roundings of P,
    // D, nG and z themselves — u |P|, u |D| -> 2u D^2 ~ 2u beta^2 P near the threshold, u (beta^2 P + beta^4 / 4), u |z| — are
    // covered by 0.4 |D| delta + 0.3 beta^2 delta (>= 5.4e-7 beta^2 M against <= 3.2e-7 beta^2 M, since P <= M_s + M_t) and
    // 4u beta^4 / 4)
    const float m_a = 2.4f * delta, m_b = 1.3f * (fb2 * delta + delta * delta) + 2.4e-7f * q4 + 1e-7f;
    const u32 pmin_bits = __float_as_uint(0.505f * fb2 + delta);
    // operand fragments: lane l holds K slots 8 (l >> 5) .. + 7 of row / column (l & 31) of a 32-block
    gbm_h8 fa[2][2], fbq[2][2];  // [cloud][row half] / [cloud][column half]
#pragma unroll
    for (int cl = 0; cl < 2; ++cl)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        // (one 16-byte read each: the address picks the half)
        fa[cl][h2] = *(const gbm_h8*)((const char*)&s_rowrec[cl][32 * h2 + c32] + 16 * half);
        fbq[cl][h2] = *(const gbm_h8*)((const char*)&s_colrec[ct][cl][32 * h2 + c32] + 16 * half);
      }
    u32 w[2][2];  // [row half][column half]: this lane's 16 decisions, bit r = entry r
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const gbm_f16x zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const gbm_f16x rs = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][rh], fbq[0][ch], zero, 0, 0, 0);
        const gbm_f16x rt = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][rh], fbq[1][ch], zero, 0, 0, 0);
        u32 cacc = 0, sacc = 0;  // decision / "not sure" bits, entry 0 ends up in bit 15
        int plo = 0x7fffffff;    // smallest P of the sixteen, as a bit pattern (a negative P is a negative integer)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const gb_f2 Rs = {rs[r], rs[r + 1]}, Rt = {rt[r], rt[r + 1]};
          const gb_f2 P = -Rs - Rt, D = Rs - Rt;
          const gb_f2 nG = __builtin_elementwise_fma(P, gb_f2{nb2, nb2}, gb_f2{q4, q4});
          const gb_f2 z = __builtin_elementwise_fma(D, D, nG);  // <= 0: consistent
          const float t0 = __builtin_fabsf(z.x) - __builtin_fmaf(__builtin_fabsf(D.x), m_a, m_b);  // < 0: not sure
          const float t1 = __builtin_fabsf(z.y) - __builtin_fmaf(__builtin_fabsf(D.y), m_a, m_b);
          cacc = __builtin_amdgcn_alignbit(cacc, __float_as_uint(z.x), 31);
          cacc = __builtin_amdgcn_alignbit(cacc, __float_as_uint(z.y), 31);
          sacc = __builtin_amdgcn_alignbit(sacc, __float_as_uint(t0), 31);
          sacc = __builtin_amdgcn_alignbit(sacc, __float_as_uint(t1), 31);
          plo = min(min((int)__float_as_uint(P.x), (int)__float_as_uint(P.y)), plo);
        }
        // entry r is row 32 rh + 8 (r >> 2) + 4 half + (r & 3), column 32 ch + c32; bits 15 - r so far
        u32 dec = __brev(cacc) >> 16, pend = __brev(sacc) >> 16;
        const bool some_short = plo <= (int)pmin_bits;  // a short TIM among the lane's sixteen (or its own vertex, on the diagonal)
        if (__any(pend != 0 || some_short || !safe)) {  // rare: which entries exactly, then the reference expression for them
          if (!safe) {
            pend = 0xffffu;
          } else if (some_short) {
            const float pminf = __uint_as_float(pmin_bits);
#pragma unroll
            for (int r = 0; r < 16; ++r) pend |= !(-(rs[r] + rt[r]) > pminf) ? (1u << r) : 0u;
          }
          // never decided (their bits are cleared further down): rows past the end of the last block, a vertex paired with
          // itself on the diagonal tile, columns past the end of the matrix
          u32 live = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 32 * rh + 8 * (r >> 2) + 4 * half + (r & 3);
            const bool ok = row < nrows && !(diag && row == 32 * ch + c32);
            live |= ok ? (1u << r) : 0u;
          }
          if (cb * 64 + 32 * ch + c32 >= L) live = 0;
          pend &= live;
          while (__any(pend != 0)) {
            if (pend != 0) {
              const int r = __ffs((int)pend) - 1;
              const int row = 32 * rh + 8 * (r >> 2) + 4 * half + (r & 3);
              const float* rp = s_rowpts[row];
              const float* cq = s_colpts[ct][32 * ch + c32];
              const double ex_ = (double)cq[0] - (double)rp[0], ey = (double)cq[1] - (double)rp[1], ez = (double)cq[2] - (double)rp[2];
              const double fx = (double)cq[3] - (double)rp[3], fy = (double)cq[4] - (double)rp[4], fz = (double)cq[5] - (double)rp[5];
              const double s2 = ex_ * ex_ + (ey * ey + ez * ez);
              const double t2 = fx * fx + (fy * fy + fz * fz);
              const u32 bit = 1u << r;
              dec = pair_consistent(s2, t2, beta, beta2) ? (dec | bit) : (dec & ~bit);
              pend &= ~bit;
            }
          }
        }
        w[rh][ch] = dec;
      }
    }
    // the lane's column words: nibble g of `dec` holds rows 8 g + 4 half + {0..3} of the 32-row half — spread to bit
    // 8 g + 4 half, merged with the other half-wave's (the lane 32 away owns the same column, the other four rows of
    // every eight); then lane L keeps column L of the tile: the left half's word in lanes 0..31, the right half's above
    u64 colw;
    {
      u32 full[2][2];
#pragma unroll
      for (int rh = 0; rh < 2; ++rh)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const u32 d = w[rh][ch];
          u32 sp = (d & 0xfu) | ((d & 0xf0u) << 4) | ((d & 0xf00u) << 8) | ((d & 0xf000u) << 12);
          sp <<= 4 * half;
          full[rh][ch] = sp | (u32)__shfl_xor((int)sp, 32, 64);
        }
      const u32 lo = half ? full[0][1] : full[0][0], hi = half ? full[1][1] : full[1][0];
      colw = ((u64)hi << 32) | lo;
    }
    if (nrows < 64) colw &= (1ULL << nrows) - 1ULL;  // rows past the end of the last block
    if (diag) colw &= ~(1ULL << lane);                // a vertex is not its own neighbour
    if (!jvalid) colw = 0;                             // columns past the end of the matrix
    if (!diag && jvalid) {  // the transposed word (the lower triangle is never evaluated)
      bm[(size_t)j * Wb + rb] = colw;
      degp[(size_t)rb * Lp + j] = (unsigned char)__popcll(colw);
    }
    u32 lo = (u32)colw, hi = (u32)(colw >> 32);
    wave_transpose64(lo, hi, lane);
    rowbuf[lane * 4 + ct] = ((u64)hi << 32) | lo;  // lane r: the word of row rb * 64 + r (diagonal tile: both triangles)
  }
  __syncthreads();
  {
    const int row = tid >> 2, wi = 4 * C + (tid & 3);
    if (row < nrows && wi < nb && wi >= rb) {
      const u64 wv = rowbuf[tid];
      bm[(size_t)(rb * 64 + row) * Wb + wi] = wv;
      degp[(size_t)wi * Lp + rb * 64 + row] = (unsigned char)__popcll(wv);
    }
  }
}

// The tile-per-workgroup kernel (rounds 2-3): four waves x 16 rows per tile, many short workgroups — the LATENCY form, used up
// to GB_TILES_MAX_L correspondences in rounds 3-5 and below GBM_MIN_L since (a small graph is a single wave of workgroups).
template <bool EXT>
__global__ __launch_bounds__(256) void k_graph_build_tiles(ViewExt<SolverView> x, SolverView one, double beta, float margin,
                                                     int prep) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int L = V.L, Wb = V.Wb;
  const int nb = (L + 63) >> 6;
  // Chores of launches that used to follow or precede this one (~5 us each on the chain), done by the first threads of
  // the grid: bit 0 — k_hcore_async's clean slate (every value at 0xffff, an upper bound of any degree; control words 0;
  // Kp and perm are not in use before the ranking); bit 1 — the solver state starts from zero (k_solver_reset).
  if (prep) {
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (prep & 1) {
      if (gi < (((L + 63) & ~63) >> 1)) ((unsigned*)V.Kp)[gi] = 0xffffffffu;
      if (gi < HCA_CTL_DONE + HCA_MAXWG) V.perm[gi] = 0;
    }
    if ((prep & 2) && gi < (int)(sizeof(SolverState) / 4)) ((int*)V.st)[gi] = 0;
  }
  // workgroup t -> tile (rb, cb), rb <= cb, row by row over the upper triangle: row rb starts at rb nb - rb (rb - 1) / 2.
  // (Grouping the tiles into 8 x 8 super-tiles dealt to one XCD each, so that the eight writers of a 64-byte line of the
  // matrix meet in one L2, was tried: 125 -> 136 us at L = 20000.  The kernel is bound by instruction issue, not by its
  // 8-byte stores.)
  int rb, cb;
  {
    const int t = blockIdx.x;
    if (t >= nb * (nb + 1) / 2) return;
    const float q = (float)(2 * nb + 1);  // (binary32 estimate, corrected by the two loops below)
    rb = (int)((q - __builtin_sqrtf(fmaxf(q * q - 8.0f * (float)t, 0.0f))) * 0.5f);
    rb = max(0, min(rb, nb - 1));
    while (rb > 0 && rb * nb - rb * (rb - 1) / 2 > t) --rb;
    while (rb + 1 < nb && (rb + 1) * nb - (rb + 1) * rb / 2 <= t) ++rb;
    cb = rb + (t - (rb * nb - rb * (rb - 1) / 2));
  }
  const float4* __restrict__ src = V.src;
  const float4* __restrict__ tgt = V.tgt;
  u64* __restrict__ bm = V.bm;
  unsigned char* __restrict__ degp = const_cast<unsigned char*>(V.degp);
  const int Lp = V.Lp;
  // A single wavefront issues one vector instruction every ~8 clocks (tests/probe/clk_probe.hip), so a tile walked by
  // one wave takes ~10 us whatever the machine is doing; the tile's 64 rows are dealt to the four waves of the
  // workgroup, 16 each.  Row points: per row (sx, tx, sy, ty) and (sz, tz) in LDS — one 16-byte and one 8-byte broadcast
  // read per row; every wave stages its own 16 rows (no workgroup barrier in front of the arithmetic).
  __shared__ __attribute__((aligned(16))) float4 rxy[64];
  __shared__ __attribute__((aligned(8))) gb_f2 rz[64];
  __shared__ unsigned short colpart[4][64];
  // (readfirstlane: tells the compiler that the wave index — and every lane mask derived from it — is wave-uniform, so
  // the masks stay in scalar registers and the branches on them are scalar branches)
  const int lane = qk_lane(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i0 = rb * 64, r0 = wave * 16;
  const int j = cb * 64 + lane;
  const bool jvalid = j < L;
  gb_f2 cx, cy, cz;
  {
    const int jj = min(j, L - 1);
    const float4 a = src[jj], b = tgt[jj];
    if (lane < 16) {
      const int i = min(i0 + r0 + lane, L - 1);
      const float4 ra = src[i], rbp = tgt[i];
      rxy[r0 + lane] = make_float4(ra.x, rbp.x, ra.y, rbp.y);
      rz[r0 + lane] = gb_f2{ra.z, rbp.z};
    }
    cx = gb_f2{a.x, b.x};
    cy = gb_f2{a.y, b.y};
    cz = gb_f2{a.z, b.z};
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float fb2 = (float)(beta * beta);
  const float c1 = 2.0f * fb2, c2 = fb2 * fb2;
  const float smin = fb2 * 1.01f;  // above it s + t - beta^2 > 0 holds in exact arithmetic too: the squared form is valid
  const double beta2 = beta * beta;
  const int nrows = min(64, L - i0);
  const bool diag = cb == rb;
  // lane masks live in scalar registers.  `act`: the columns that exist.  The row's own column (diagonal tile) and the
  // rows past the end of the last block go through the loop like everything else — their bits are cleared afterwards.
  const u64 act = __ballot(jvalid);
  u32 row_lo = 0, row_hi = 0, cacc = 0;
  for (int rr = r0; rr < r0 + 16; rr += 8) {
    // eight rows at a time: the arithmetic of the eight is independent (the scheduler interleaves it), and the one
    // scalar branch per group asks whether ANY of the 512 pairs fell into the band the binary32 screen leaves undecided
    u64 m_lt[8], need[8];
    u64 any = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 pxy = rxy[rr + k];
      const gb_f2 pz = rz[rr + k];
      const gb_f2 dx = cx - gb_f2{pxy.x, pxy.y}, dy = cy - gb_f2{pxy.z, pxy.w}, dz = cz - pz;
      gb_f2 q = dx * dx;
      q = __builtin_elementwise_fma(dy, dy, q);
      q = __builtin_elementwise_fma(dz, dz, q);  // (s, t): squared TIM lengths of (source, target)
      const float S = q.x + q.y, D = q.x - q.y;
      const float G = __builtin_fmaf(S, c1, -c2);
      const float z = __builtin_fmaf(D, D, -G);  // < 0: consistent
      m_lt[k] = __ballot(z < 0.0f);
      const u64 sure = __ballot(__builtin_fabsf(z) > margin * G) & __ballot(S > smin) & __ballot(S < GB_SMAX);
      need[k] = act & ~sure;
      any |= need[k];
    }
    if (any != 0) {  // rare: the band is ~1e-5 of the pairs, plus TIMs shorter than beta
      // Not to be decided at all (their bits are cleared below; a zero-length TIM would cost a binary64 evaluation every
      // time): a vertex paired with itself on the diagonal tile, and the rows past the end of the last block.
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (diag) need[k] &= ~(1ULL << (rr + k));
        if (rr + k >= nrows) need[k] = 0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (need[k] != 0) {
          bool ex = false;
          if ((need[k] >> lane) & 1) {
            const float4 pxy = rxy[rr + k];
            const gb_f2 pz = rz[rr + k];
            const double ex_ = (double)cx.x - (double)pxy.x, ey = (double)cy.x - (double)pxy.z, ez = (double)cz.x - (double)pz.x;
            const double fx = (double)cx.y - (double)pxy.y, fy = (double)cy.y - (double)pxy.w, fz = (double)cz.y - (double)pz.y;
            const double s = ex_ * ex_ + (ey * ey + ez * ez);
            const double t = fx * fx + (fy * fy + fz * fz);
            ex = pair_consistent(s, t, beta, beta2);
          }
          m_lt[k] = (m_lt[k] & ~need[k]) | (__ballot(ex) & need[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u64 e = m_lt[k] & act;
      // lane (r mod 16) keeps row r's word (lane select through m0: a second scalar register would break the
      // one-constant-bus operand rule of this instruction set)
      asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                   : "+v"(row_lo), "+v"(row_hi)
                   : "s"((u32)e), "s"((u32)(e >> 32)), "s"(rr + k - r0)
                   : "m0");
      u64 junk;
      // every lane shifts its own bit of `e` into its column word: cacc = 2 cacc + e[lane] (bit order reversed below)
      asm volatile("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(cacc), "=s"(junk) : "s"(e));
    }
  }
  // rows: lane l < 16 of wave w holds the word of row 16 w + l
  {
    u64 roww = ((u64)row_hi << 32) | row_lo;
    const int r = r0 + lane;
    if (diag) roww &= ~(1ULL << (r & 63));  // a vertex is not its own neighbour (the row word covers both triangles)
    if (lane < 16 && r < nrows) {
      bm[(size_t)(i0 + r) * Wb + cb] = roww;
      degp[(size_t)cb * Lp + i0 + r] = (unsigned char)__popcll(roww);
    }
  }
  if (diag) return;  // (uniform over the workgroup) the diagonal tile's row words already hold both triangles
  // columns: the wave's 16 bits (first row in bit 15 after the shifts) — four pieces per column, put together by wave 0
  colpart[wave][lane] = (unsigned short)(__brev(cacc) >> 16);
  __syncthreads();
  if (wave == 0 && jvalid) {
    u64 colw = (u64)colpart[0][lane] | ((u64)colpart[1][lane] << 16) | ((u64)colpart[2][lane] << 32) |
               ((u64)colpart[3][lane] << 48);
    if (nrows < 64) colw &= (1ULL << nrows) - 1ULL;  // rows past the end of the last block
    bm[(size_t)j * Wb + rb] = colw;
    degp[(size_t)rb * Lp + j] = (unsigned char)__popcll(colw);
  }
}

// =================================================================================================
// K12a: exact core numbers by level-synchronous peeling, one workgroup (the graph of one registration
// is small: L <= ~24k vertices).  Degrees and the frontier queue live in LDS; adjacency rows are read
// from the bit matrix (L2 / Infinity-Cache resident).  Core numbers equal pmc_graph::compute_cores'
// (Batagelj-Zaversnik) by uniqueness of the k-core decomposition.
#define KC_REMOVED (-(1 << 30))
// workgroup barrier that orders LDS traffic only: the core_out stores issued inside the peeling loop are
// never read by this kernel, and waiting for their HBM acknowledgement would cost ~1.5 us per round
#define KC_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory")
// (a device function: k_rank_sort runs it in its own launch — always for the graphs that have no other core-number
// kernel, or when k_hcore_async gave up — and the kernel k_kcore below wraps it for the older chains)
__device__ __forceinline__ void d_kcore(const SolverView& V, int use_gqueue /* the queue does not fit LDS */,
                                        int lds_bitmap_max) {
  const u64* __restrict__ bm = V.bm;
  const int L = V.L, W = V.W;
  if (L <= 0) return;
  const int* __restrict__ deg_in = V.deg;
  int* __restrict__ core_out = V.core;
  SolverState* __restrict__ st = V.st;
  int* __restrict__ gqueue = use_gqueue ? V.picks : nullptr;
  // the launch sized its LDS for the largest pair of the group: a smaller matrix fits a fortiori
  const int lds_bitmap = lds_bitmap_max;
  extern __shared__ __attribute__((aligned(16))) int kc_lds[];
  int* deg = kc_lds;
  int* queue = gqueue ? gqueue : kc_lds + L;
  __shared__ int s_qn[2], s_min[2];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  __shared__ int s_edges2;
  if (tid == 0) {
    s_qn[0] = s_qn[1] = 0;
    s_min[0] = s_min[1] = 0x7fffffff;
    s_edges2 = 0;
  }
  __syncthreads();
  {
    int esum = 0;
    for (int v = tid; v < L; v += nthr) {
      const int d = solver_degree(V, v);
      deg[v] = d;
      esum += d;
    }
    esum = wave_sum_i32(esum);
    if (lane == 0 && esum) atomicAdd(&s_edges2, esum);
  }
  // adjacency rows: from LDS when the whole bit matrix fits behind the degree/queue arrays
  const u64* rows = bm;
  if (lds_bitmap) {
    u64* lb = (u64*)(kc_lds + ((2 * L + 1) & ~1));
    for (size_t e = tid; e < (size_t)L * W; e += nthr) lb[e] = bm[(e / W) * (size_t)V.Wb + (e % W)];
    rows = lb;
  }
  __syncthreads();
  if (tid == 0) st->n_edges2 = s_edges2;
  int k = -1;  // first round: nothing has degree <= -1, so it only computes the minimum degree
  int maxcore = 0;
  int kc_rounds = 0;
  for (int round = 0;; ++round) {
    const int p = round & 1;
    kc_rounds = round + 1;
    // detection: vertices with degree <= k leave the graph at core k; the rest vote for the next level
    int m = 0x7fffffff;
    for (int v = tid; v < L; v += nthr) {
      const int d = deg[v];
      if (d > KC_REMOVED / 2) {
        if (d <= k) {
          deg[v] = KC_REMOVED;  // -2^30: at most L <= 2^15 further decrements keep it below KC_REMOVED/2, no wrap-around
          core_out[v] = k;
          queue[atomicAdd(&s_qn[p], 1)] = v;
        } else {
          m = min(m, d);
        }
      }
    }
    m = wave_min_i32(m);
    if (lane == 0 && m != 0x7fffffff) atomicMin(&s_min[p], m);
    if (gqueue) __syncthreads(); else KC_BARRIER_LDS();  // a global queue needs the full fence
    const int n = s_qn[p], mn = s_min[p];
    if (tid == 0) {
      s_qn[p ^ 1] = 0;
      s_min[p ^ 1] = 0x7fffffff;
    }
    if (n == 0) {
      if (mn == 0x7fffffff) break;  // nothing left
      k = mn;                       // level exhausted: jump to the smallest remaining degree
    } else {
      maxcore = k;
      // (frontier vertex, word) items are independent: flattened over all threads so that the row loads
      // of a round overlap instead of forming one latency chain per wave
      // (frontier vertex, word) items are independent: flattened over all threads so that the row loads of
      // a round overlap; each lane walks the set bits of its word (up to 64 conflict-tolerant LDS atomics per
      // step across the wave, whatever the density)
      const int Wr = lds_bitmap ? W : V.Wb;  // (row stride of `rows`: the LDS copy is packed)
      for (int item = tid; item < n * W; item += nthr) {
        const int qi = item / W, w = item - qi * W;
        u64 x = rows[(size_t)queue[qi] * Wr + w];
        while (x) {
          const int b = __ffsll((long long)x) - 1;
          x &= x - 1;
          atomicSub(&deg[w * 64 + b], 1);
        }
      }
    }
    if (gqueue) __syncthreads(); else KC_BARRIER_LDS();  // a global queue needs the full fence
  }
  if (tid == 0) {
    st->max_core = maxcore;
    st->ub = maxcore + 1;
    st->pad[0] = kc_rounds;  // statistics
  }
  for (int v = tid; v < L; v += nthr) V.rankof[v] = 0;  // k_rank_partial accumulates into it
}
template <bool EXT>
__global__ __launch_bounds__(1024) void k_kcore(ViewExt<SolverView> x, SolverView one, int use_gqueue, int lds_bitmap_max,
                                                int after_hcore /* 1: only needed if the h-index sweeps gave up */) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  if (V.L <= 0) return;
  if (after_hcore == 1 && V.st->pad[5] == 0) return;  // (k_hcore_finish left its verdict there)
  d_kcore(V, use_gqueue, lds_bitmap_max);
}

// =================================================================================================
// K12c: core numbers of larger graphs (L > 1280) by h-index iteration.  Peeling is a chain of dependent rounds — one per
// occupied degree level and sub-round, thousands at L = 20000, ~5 us each in one workgroup.  The coreness is also the
// fixed point of  c(v) <- H({c(u) : u ~ v})  started from the degrees (Lu, Zhou, Zhang, Stanley 2016; asynchronous
// updates converge to the same point: Montresor, De Pellegrini, Miorandi 2013), H = the h-index of a multiset, and the
// iteration is a handful of fully parallel sweeps over the bit matrix: one wavefront per row gathers its neighbours'
// current values, counts them into an LDS histogram capped at its own value (values only ever decrease) and reads the
// new value off the suffix counts.  Updates are in place; a stale read is an upper bound, which is all the iteration
// needs.  Core numbers are unique, so the result equals the peeling's (and pmc_graph::compute_cores').
// HC_MAXIT sweeps are enqueued; a sweep returns at once when the one before it changed nothing (14 - 35 do something on
// the synthetic correspondence sets of 2000 - 20000 pairs; the rest cost ~1 us each).  If the last one still changed
// something (long chains could do it) k_kcore runs after all.  Measured against the peeling kernel, whole solve:
// L = 5000 0.80 -> 0.62 ms, 8192 1.63 -> 1.09 ms, 20000 6.95 -> 3.1 ms.  (Round 2's chain: since round 3 the default is
// k_hcore_async below — one launch — and these kernels run under QTR_KCORE=sweeps only.)
#define HC_MAXIT 64
#define HC_BINS 2048
template <bool EXT>
__global__ __launch_bounds__(256) void k_hcore_init(ViewExt<SolverView> x, SolverView one) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v < V.L) {
    const int d = solver_degree(V, v);
    V.deg[v] = d;  // k_hcore_finish adds them up
    V.core[v] = d;
    V.Kp[v] = 0;  // the sweep from which on the vertex has to be looked at again (Kp is not in use yet)
  }
  if (v < HC_MAXIT + 1) V.perm[v] = 0;  // per-sweep "something changed" flags (perm is not in use yet)
}
template <bool EXT>
__global__ __launch_bounds__(256) void k_hcore_sweep(ViewExt<SolverView> x, SolverView one, int it) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  int* __restrict__ flags = V.perm;
  if (it > 0 && flags[it - 1] == 0) return;
  __shared__ int hist[4][HC_BINS];
  const int L = V.L, W = V.W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= L) return;
  int* __restrict__ core = V.core;
  int* __restrict__ due = V.Kp;
  // a value can only drop after a neighbour's did: whoever lowers its value stamps its neighbours "due next sweep", and
  // a vertex nobody stamped since its last visit is skipped (the late sweeps touch a few hundred rows, not all L)
  if (due[row] < it) return;
  const u64* __restrict__ rowp = V.bm + (size_t)row * V.Wb;
  const int cv = core[row];
  if (cv <= 0) return;
  int h;
  if (cv < HC_BINS) {
    int* hw = hist[wave];
    for (int t = lane; t <= cv; t += 64) hw[t] = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int w = lane; w < W; w += 64) {
      u64 bits = rowp[w];
      while (bits) {
        const int b = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        atomicAdd(&hw[min(core[w * 64 + b], cv)], 1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // largest t <= cv with #(values >= t) >= t.  Lane l owns bins [l K, (l + 1) K); suffix counts across lanes by a scan.
    const int K = (cv + 64) / 64;
    const int lo = lane * K, hi = min(cv + 1, lo + K);
    int mine = 0;
    for (int t = lo; t < hi; ++t) mine += hw[t];
    int tot;
    const int ex = wave_excl_scan_i32(mine, &tot);
    int run = tot - ex - mine;  // values in the bins of higher lanes
    int best = -1;
    for (int t = hi - 1; t >= lo; --t) {
      run += hw[t];
      if (run >= t) {
        best = t;
        break;
      }
    }
    h = wave_max_i32(best);
  } else {
    // a vertex of very high degree: count by threshold, bisect (the predicate #(values >= t) >= t is monotone in t)
    auto count_ge = [&](int th) {
      int c = 0;
      for (int w = lane; w < W; w += 64) {
        u64 bits = rowp[w];
        while (bits) {
          const int b = __ffsll((long long)bits) - 1;
          bits &= bits - 1;
          c += core[w * 64 + b] >= th;
        }
      }
      return wave_sum_i32(c);
    };
    if (count_ge(cv) >= cv) {
      h = cv;
    } else {
      int a = 0, z = cv - 1;
      while (a < z) {
        const int mid = (a + z + 1) >> 1;
        if (count_ge(mid) >= mid) a = mid;
        else z = mid - 1;
      }
      h = a;
    }
  }
  if (h < cv) {
    if (lane == 0) {
      core[row] = h;
      flags[it] = 1;
    }
    for (int w = lane; w < W; w += 64) {
      u64 bits = rowp[w];
      while (bits) {
        const int b = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        due[w * 64 + b] = it + 1;
      }
    }
  }
}
// K12d: the same fixed-point iteration in ONE launch.  The sweeps above are 20 - 35 dependent launches of a few
// microseconds of work each (plus the ones that find nothing left to do): half of the whole solve at L = 5000, two
// thirds at L = 20000.  Here the workgroups of one launch stay resident and iterate WITHOUT barriers between them:
//   * workgroup w owns a contiguous block of rows, kept in LDS as neighbour LISTS (built once from the bit rows);
//   * the current values live in a global array of 16-bit words that is read and written with device-scope relaxed
//     atomics only (every access goes to the coherence point of the 8 XCDs; no fences, so no L2 write-back/invalidate —
//     which is what made a grid barrier cost more than a launch on this part).  An iteration of a workgroup = snapshot
//     of all L values into LDS, h-index of its own rows (unless nothing at all moved), new values stored.  A stale value
//     is an upper bound, values only decrease, and the iteration from any upper bounds converges to the core numbers
//     (Montresor et al. 2013), so no ordering between workgroups is needed — not even for the start: the values are
//     0xffff until their owner has published its degrees;
//   * termination without a contended word (256 workgroups doing compare-and-swap on one address cost milliseconds:
//     a device-scope atomic is a round trip to memory): every workgroup has its own version counter ver[w], bumped after
//     it stored lowered values, and its own mark done[w].  The "epoch" is the vector ver[] (its sum E is monotone).  An
//     iteration reads ver[] BEFORE its snapshot; if nothing changed it reads ver[] again, and if the two reads agree no
//     workgroup published a change in between: the workgroup's rows are a fixed point of a snapshot taken at epoch E.
//     It writes done[w] = E + 1 and polls: ver[] moved -> iterate again; every done[u] == E + 1 -> finished.  (No
//     workgroup can register at E after the first bump beyond E became visible, and the first workgroup to bump beyond
//     E cannot have registered at E — it would have had to be woken by an earlier bump.  So "all marks equal E + 1"
//     means everybody verified its rows against the values that stand.)  All workgroups of a pair must be co-resident
//     (the host sizes the grid for that: at most one workgroup per compute unit).
// If HCA_MAXITER iterations do not suffice (never seen) the failure flag sends the pair to the peeling workgroup
// (d_kcore, inside k_rank_sort's launch).
#define HCA_THREADS 1024
#define HCA_MAXITER 4096
#ifdef QTR_HCA_PROF  // diagnostic build only (tests/gpu_hca_prof.py): where an iteration's time goes, per workgroup
__device__ unsigned g_hca_prof[HCA_MAXWG * 8];
#define HCA_MARK(k)                                   \
  do {                                                \
    if (tid == 0) {                                   \
      const unsigned long long now_ = wall_clock64(); \
      prof_acc[k] += (unsigned)(now_ - prof_last);    \
      prof_last = now_;                               \
    }                                                 \
  } while (0)
#define HCA_COUNT(k) \
  do {               \
    if (tid == 0) ++prof_acc[k]; \
  } while (0)
#else
#define HCA_MARK(k)
#define HCA_COUNT(k)
#endif
__device__ __forceinline__ unsigned hca_load_u32(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 hca_load_u64(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The scout: one more workgroup of a single pair's launch on a large graph (8192 < L <= 32768), which owns no rows and
// takes no part in the termination protocol.  The floor H / 2 above is a bet on the degree sequence and stays away where
// the h-index of the degrees does not stand clear of their mean — the generator's graphs with 2 % planted at L = 20000
// (bulk: a 198-core with degrees up to 591; clique 400; fifty iterations, 380 us) among them.  A floor that is no bet on
// the graph's shape: the size of a clique somebody has SEEN.  The scout takes the values as they stand once every degree
// is published, keeps the (at most) HCA_SCOUT_M vertices with the largest ones, copies their sub-matrix into LDS (a wave
// per row: the row staged, one ballot per 64 columns) and peels it — every round the alive vertices count their alive
// neighbours, all of equal count n - 1 is a clique, otherwise those within an eighth of the range above the minimum go.
// On the generator's graphs the planted clique is what is left after eight rounds (tests/probe/scout_sim.py: 1 - 3 %
// planted at L = 10000 / 20000, from the degrees or after one to three iterations; nothing planted: no clique of sixteen,
// no floor).  A clique of s vertices puts s - 1 - s / 8 into HCA_CTL_FLOOR2 (its members' core numbers are >= s - 1; the
// eighth is slack for a search that finds a little less than the scout did).  The iterating workgroups look at that word
// once per iteration and raise their floor: values already below it stop, whatever they were; values at or above it go on
// to their exact core numbers (the argument at "the floor" in k_hcore_async does not ask WHEN a value below the floor
// stopped, only that it is an upper bound below the floor).  k_rank_sort starts the search from the larger of the two
// floors; a search that comes back empty-handed sends the stage round again without any floor, as before.
#define HCA_SCOUT_M 768
__device__ __forceinline__ void hca_scout(const u64* __restrict__ bm, int Wb, int W, int L, unsigned* ctl, const unsigned short* gvals,
                          unsigned char* lds, size_t lds_bytes, unsigned* s_hist /* [HCA_THREADS], the kernel's */,
                          unsigned long long t_start) {
  constexpr int T = HCA_THREADS, MWX = HCA_SCOUT_M / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __shared__ unsigned short s_cand[HCA_SCOUT_M];
  __shared__ u64 s_alive[MWX];
  __shared__ int s_red[4][T / 64];
  __shared__ int s_flag, s_m;
  const int Lp = (L + 63) & ~63;
  if ((size_t)2 * Lp > lds_bytes) return;
  unsigned short* vals = (unsigned short*)lds;
  // every vertex's degree (or what its owner has made of it since): snapshots until none is missing
  while (true) {
    bool open = false;
    for (int i = tid; i < (Lp >> 2); i += T) {
      const u64 nv = hca_load_u64((const u64*)gvals + i);
      ((u64*)vals)[i] = nv;
#pragma unroll
      for (int k = 0; k < 4; ++k) open = open || (4 * i + k < L && ((nv >> (16 * k)) & 0xffffu) == 0xffffu);
    }
    if (tid == 0) s_flag = 0;
    __syncthreads();
    if (__any(open) && lane == 0) atomicOr(&s_flag, 1);
    if (tid == 0 && wall_clock64() - t_start > 20000ull) atomicOr(&s_flag, 2);  // (200 us: somebody is not resident)
    __syncthreads();
    const int fl = s_flag;
    __syncthreads();
    if (!(fl & 1)) break;
    if (fl & 2) return;
  }
  // theta = the smallest value with at most HCA_SCOUT_M vertices at or above it (values clamped to T - 1: a graph whose
  // top values are all beyond that has no use for a floor the scout could find)
  s_hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < L; i += T) atomicAdd(&s_hist[min((int)vals[i], T - 1)], 1u);
  __syncthreads();
  int theta, m;
  {
    const int mine_h = (int)s_hist[tid];
    int wtot = 0;
    const int below_incl = wave_excl_scan_i32(mine_h, &wtot) + mine_h;
    if (lane == 0) s_red[0][wave] = wtot;
    __syncthreads();
    int above = 0;
#pragma unroll
    for (int q = 0; q < T / 64; ++q) above += q > wave ? s_red[0][q] : 0;
    const int cnt_ge = above + (wtot - below_incl) + mine_h;  // #{values >= tid}
    const int t_ok = (tid >= 2 && cnt_ge <= HCA_SCOUT_M) ? tid : T;
    const int wmin = wave_min_i32(t_ok);
    __syncthreads();
    if (lane == 0) s_red[1][wave] = wmin;
    __syncthreads();
    theta = T;
#pragma unroll
    for (int q = 0; q < T / 64; ++q) theta = min(theta, s_red[1][q]);
    if (theta >= T - 1) return;
    if (tid == theta) s_m = cnt_ge;
    __syncthreads();
    m = s_m;
    if (m < 16) return;
    if (tid == 0) s_m = 0;
    __syncthreads();
  }
  for (int i = tid; i < L; i += T)
    if ((int)vals[i] >= theta) s_cand[atomicAdd(&s_m, 1)] = (unsigned short)i;
  __syncthreads();  // (vals is spent: the sub-matrix takes its place)
  const int MW = (m + 63) >> 6;
  u64* sub = (u64*)lds;  // [m][MW]
  const size_t sub_bytes = (((size_t)m * MW * 8) + 15) & ~(size_t)15;
  if (sub_bytes + (size_t)(T / 64) * W * 8 > lds_bytes || W > 512) return;
  {
    u64* stage = (u64*)(lds + sub_bytes) + (size_t)wave * W;  // this wave's copy of one row of the graph
    int cid[MWX];
#pragma unroll
    for (int k = 0; k < MWX; ++k) cid[k] = (64 * k + lane < m) ? (int)s_cand[64 * k + lane] : -1;
    // (the next row's words are on their way — in registers — while this one's ballots run: a wave alone would otherwise
    // sit out a memory round trip per row, 48 rows deep)
    constexpr int WPL = 8;  // words of a row per lane: W <= 512
    u64 nxt[WPL];
    auto fetch = [&](int r) __attribute__((always_inline)) {
      const u64* rowp = bm + (size_t)s_cand[r] * Wb;
#pragma unroll
      for (int q = 0; q < WPL; ++q) nxt[q] = (lane + 64 * q < W) ? rowp[lane + 64 * q] : 0ULL;
    };
    if (wave < m) fetch(wave);
    for (int r = wave; r < m; r += T / 64) {
#pragma unroll
      for (int q = 0; q < WPL; ++q)
        if (lane + 64 * q < W) stage[lane + 64 * q] = nxt[q];
      if (r + T / 64 < m) fetch(r + T / 64);
#pragma unroll
      for (int k = 0; k < MWX; ++k)
        if (k < MW) {
          const bool bit = cid[k] >= 0 && ((stage[cid[k] >> 6] >> (cid[k] & 63)) & 1ULL) != 0;
          const u64 wd = __ballot(bit);
          if (lane == 0) sub[(size_t)r * MW + k] = wd;
        }
    }
  }
  bool alive = tid < m;
  {
    const u64 b = __ballot(alive);
    if (lane == 0 && wave < MWX) s_alive[wave] = b;
  }
  __syncthreads();
  int clique = 0;
  for (int round = 0; round < 512; ++round) {
    int d = 0;
    if (alive)
      for (int k = 0; k < MW; ++k) d += __popcll(sub[(size_t)tid * MW + k] & s_alive[k]);
    const int wmn = wave_min_i32(alive ? d : 0x7fffffff), wmx = wave_max_i32(alive ? d : -1), wn = wave_sum_i32(alive ? 1 : 0);
    if (lane == 0) {
      s_red[0][wave] = wmn;
      s_red[1][wave] = wmx;
      s_red[2][wave] = wn;
    }
    __syncthreads();
    int mn = 0x7fffffff, mx = -1, n = 0;
#pragma unroll
    for (int q = 0; q < T / 64; ++q) {
      mn = min(mn, s_red[0][q]);
      mx = max(mx, s_red[1][q]);
      n += s_red[2][q];
    }
    if (n < 16) break;
    if (mn == n - 1) {
      clique = n;
      break;
    }
    const int thr = mn + ((mx - mn) >> 3);
    alive = alive && d > thr;
    __syncthreads();  // (everybody has read the masks and the partial results of this round)
    const u64 b = __ballot(alive);
    if (lane == 0 && wave < MWX) s_alive[wave] = b;
    __syncthreads();
  }
  const int f2 = clique - 1 - (clique >> 3);
  if (clique >= 16 && f2 >= HCA_FLOOR_MIN && tid == 0)
    __hip_atomic_store(ctl + HCA_CTL_FLOOR2, ((unsigned)f2 << 1) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (the clean slate for graphs that were handed in as bit matrices; k_graph_build prepares its own, see there)
template <bool EXT>
__global__ __launch_bounds__(256) void k_hcore_async_init(ViewExt<SolverView> x, SolverView one) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int gi = blockIdx.x * 256 + threadIdx.x;
  if (gi < (((V.L + 63) & ~63) >> 1)) ((unsigned*)V.Kp)[gi] = 0xffffffffu;
  if (gi < HCA_CTL_DONE + HCA_MAXWG) V.perm[gi] = 0;
}
template <bool EXT>
__global__ __launch_bounds__(HCA_THREADS) void k_hcore_async(ViewExt<SolverView> x, SolverView one, int pool_entries,
                                                              int allow_floor) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int L = V.L, W = V.W;
  if (L <= 0) return;
  // (allow_floor = 2: the last workgroup of the launch is the scout, see hca_scout)
  const bool has_scout = allow_floor == 2;
  const int NWG = (int)gridDim.x - (has_scout ? 1 : 0), w = blockIdx.x;
  const unsigned long long t_start = wall_clock64();
  const int R = (L + NWG - 1) / NWG, Rp = (R + 3) & ~3;
  const int r_lo = min(L, w * R), r_hi = min(L, r_lo + R), nown = r_hi - r_lo;
  const u64* __restrict__ bm = V.bm;
  unsigned short* gvals = (unsigned short*)V.Kp;
  unsigned* ver = (unsigned*)V.perm + HCA_CTL_VER;
  unsigned* done = (unsigned*)V.perm + HCA_CTL_DONE;
  extern __shared__ __attribute__((aligned(16))) unsigned char hca_lds[];
  const int Lp = (L + 63) & ~63;
  unsigned char* lp = hca_lds;
  unsigned short* vals = (unsigned short*)lp;  // [Lp] the snapshot
  lp += (size_t)2 * Lp;
  int* nb_off = (int*)lp;  // [Rp + 4] where my rows' neighbour lists start in the pool (entry nown: the end of the last)
  lp += (size_t)4 * (Rp + 4);
  int* mine = (int*)lp;  // [Rp] my rows' current values (what I stored last)
  lp += (size_t)4 * Rp;
  unsigned short* pool = (unsigned short*)lp;  // [pool_entries] neighbour ids of my rows, row after row
  __shared__ unsigned s_hist[HCA_THREADS];
  if (has_scout && w == NWG) {
    hca_scout(bm, V.Wb, W, L, (unsigned*)V.perm, gvals, hca_lds, (size_t)(lp - hca_lds) + (size_t)2 * pool_entries, s_hist, t_start);
    return;
  }
  __shared__ unsigned s_sum[HCA_THREADS / 64];
  // "does any thread of the workgroup ...": ONE barrier per vote.  (__syncthreads_or / _and compile to three — the flag
  // is initialised, or-ed and read between barriers of its own — and an iteration takes two votes: six barriers of sixteen
  // waves in a ~2 us iteration.)  Four words take turns; a vote clears the word of the next one, which was last read
  // three votes — at least two barriers — ago.
  __shared__ unsigned s_vote[4];
  int vote_ix = 0;
  const int tid = threadIdx.x, lane = tid & 63;
  // (uniform on purpose: what is indexed by the wave — a row's value, where its list starts — then lives in scalar
  // registers and the branches on it are scalar branches instead of exec-mask games)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- set-up: my rows as neighbour LISTS.  A wave issues an instruction every ~8 clocks however many of its lanes are
  // busy, and every iteration of every workgroup waits for its slowest wave: what an iteration costs is the NUMBER of
  // instructions a row takes.  Off the bit row, a neighbour's value costs a find-first-set / clear-lowest / address /
  // read chain of ~17 instructions and most lanes hold no neighbour at all; off a list it is two LDS reads per lane and
  // a row of 80 neighbours is two per lane.  (Rows whose list does not fit the pool keep the bit row, read from memory.)
  for (int i = tid; i < (Lp >> 1); i += HCA_THREADS) ((unsigned*)vals)[i] = 0xffffffffu;  // "everything moved" the first time
  if (tid < 4) s_vote[tid] = 0;
  auto wg_any = [&](bool p) __attribute__((always_inline)) -> bool {
    const int me = vote_ix & 3;
    ++vote_ix;
    if (tid == 0) s_vote[(me + 1) & 3] = 0;
    if (__any(p) && lane == 0) atomicOr(&s_vote[me], 1u);
    __syncthreads();
    return s_vote[me] != 0;
  };
  // my rows' degrees (a wave per row adds up the row's per-block counts) are their first values: published at once — the
  // values everybody starts from are 0xffff, an upper bound like any other, so a workgroup that looks before its
  // neighbours have published loses nothing but a little tightness in its first iteration
  for (int rl = wave; rl < nown; rl += HCA_THREADS / 64) {
    const int v = r_lo + rl;
    int d;
    if (V.degp) {
      int part = 0;
      for (int k = lane; k < ((L + 63) >> 6); k += 64) part += V.degp[(size_t)k * V.Lp + v];
      d = wave_sum_i32(part);
    } else {
      d = V.deg[v];
    }
    if (lane == 0) {
      nb_off[rl] = d;  // (turned into offsets below)
      mine[rl] = min(d, 65535);
      if (V.degp) V.deg[v] = d;  // k_rank_sort adds them up
      __hip_atomic_store(gvals + v, (unsigned short)min(d, 65535), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  // ---- the floor.  Only the vertices whose core number can reach the size of the clique the search will find matter to
  // it (a member of a clique of s vertices has core number >= s - 1; everything else is cut away by the K > mc tests), but
  // MOST of the iteration's length is the slow settling of the graph's bulk (random consistencies: thousands of vertices
  // of degree ~70 creeping down to core ~57 over twenty dependent rounds) far below the planted clique.  Workgroup 0 waits
  // until every vertex has published its degree, takes the h-index H of the degrees (H - 1 bounds every clique from above)
  // and publishes floor = H / 2 (if H stands clear of the mean degree, see there); the others wait for that word.  A value
  // below the floor is left where it is: an upper bound of its core number below the floor — such a row is never counted,
  // a row whose DEGREE is below the floor not even listed.  Values at or above the floor still converge to the exact core
  // numbers (a neighbour below the floor never counts at a threshold at or above it, wherever below it stands).  What the
  // clique search makes of this — it starts from the floor as an injected lower bound and the host repeats the stage
  // without a floor if that search comes back empty — is described at k_rank_sort.
  __shared__ unsigned s_floor;
  __shared__ int s_froze;
  __shared__ int s_hcnt[2][HCA_THREADS / 64];
  if (tid == 0) s_froze = 0;  // (barriers follow before anybody evaluates a row)
  int my_floor = 0;            // (uniform)
  if (allow_floor) {
    unsigned* ctl_floor = (unsigned*)V.perm + HCA_CTL_FLOOR;
    if (w == 0) {
      // every vertex's degree: snapshots until none is missing (they double as the iteration's first snapshot).  A
      // workgroup that has not published 200 us after my start is not resident (see the verdict loop below): no floor.
      bool complete = false;
      while (!complete) {
        bool open = false;
        for (int i = tid; i < (Lp >> 2); i += HCA_THREADS) {
          const u64 nv = hca_load_u64((const u64*)gvals + i);
          ((u64*)vals)[i] = nv;
#pragma unroll
          for (int k = 0; k < 4; ++k) open = open || (4 * i + k < L && ((nv >> (16 * k)) & 0xffffu) == 0xffffu);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        complete = !wg_any(open);
        if (!complete && wg_any(wall_clock64() - t_start > 20000ull)) break;
      }
      int f = 0;
      if (complete) {
        // H = the largest h with at least h values >= h: a histogram of the values (clamped to HCA_THREADS - 1: a larger H
        // only means a lower floor than it could be), thread h then holds #{values >= h} after a suffix sum and the largest
        // h that passes is found by a maximum.  (A bisection of thirteen probes, one barrier each, took 7 us.)
        s_hist[tid] = 0;
        __syncthreads();
        int sum = 0;
        for (int i = tid; i < L; i += HCA_THREADS) {
          const int x = vals[i];
          sum += x;
          atomicAdd(&s_hist[min(x, HCA_THREADS - 1)], 1u);
        }
        sum = wave_sum_i32(sum);
        __syncthreads();
        // inclusive suffix sum: lane order reversed inside the wave, then the waves above mine
        const int mine_h = (int)s_hist[tid];
        int wtot = 0;
        const int below_incl = wave_excl_scan_i32(mine_h, &wtot) + mine_h;  // bins of my wave up to and including mine
        if (lane == 0) {
          s_hcnt[0][wave] = wtot;
          s_hcnt[1][wave] = sum;
        }
        __syncthreads();
        int above = 0;  // bins of the waves above mine
        long long tot = 0;
#pragma unroll
        for (int q = 0; q < HCA_THREADS / 64; ++q) {
          above += q > wave ? s_hcnt[0][q] : 0;
          tot += s_hcnt[1][q];
        }
        const int cnt_ge = above + (wtot - below_incl) + mine_h;  // #{values >= tid} (the last bin: >= HCA_THREADS - 1)
        int lo = wave_max_i32(cnt_ge >= tid ? tid : 0);
        __syncthreads();  // (s_hcnt[0] is read above)
        if (lane == 0) s_hcnt[0][wave] = lo;
        __syncthreads();
        lo = 0;
#pragma unroll
        for (int q = 0; q < HCA_THREADS / 64; ++q) lo = max(lo, s_hcnt[0][q]);
        // ... and only where H stands clear of the bulk: with H below 2.5 x the mean degree it is the bulk's own tail that
        // sets it (no planted clique, or one of a per cent of the vertices: consistency graphs of random correspondences
        // have degrees of mean 66 and h-index ~100 at L = 5000), half of it lies around the bulk's core numbers, and the
        // search under that bound would come back empty or tied — a second run of the whole stage instead of a shorter
        // first one
        f = lo >> 1;
        if (2ll * lo * L < 5ll * tot) f = 0;
        if (f < HCA_FLOOR_MIN) f = 0;
        // Round 5: on LARGE graphs the bet is checked a little before it is placed.  The matcher's own correspondences of a
        // dense scan pair (13 859 mutual nearest neighbours of two 50 000-point clouds) give H = 568 = 2.7 x the mean degree
        // — a dense geometric structure, but its largest clique has 174 members: the search under a floor of 284 came back
        // empty and the whole stage ran twice.  The vertices T of degree >= H of a planted clique are adjacent to each other
        // (edge density among T: 1.00 on the generator's graphs that pass the rule above, 0.79 on that structure — table in
        // tests/probe/next_round.md): 1024 pairs of T are sampled in the bit matrix, and below 0.9 there is no floor.  Only
        // above 8192 vertices, where two microseconds do not show (the headline's L = 5000 keeps the rule as it was).
        if (f > 0 && L > 8192) {
          __shared__ int s_nt, s_edges[HCA_THREADS / 64], s_pairs[HCA_THREADS / 64];
          unsigned short* s_top = (unsigned short*)s_hist;  // (the histogram is spent: 2048 vertex ids fit its 4 KB)
          if (tid == 0) s_nt = 0;
          __syncthreads();
          for (int i = tid; i < L; i += HCA_THREADS)
            if ((int)vals[i] >= lo) {
              const int at = atomicAdd(&s_nt, 1);
              if (at < 2 * HCA_THREADS) s_top[at] = (unsigned short)i;
            }
          __syncthreads();
          const int nt = min(s_nt, 2 * HCA_THREADS);
          int e = 0, pr = 0;
          if (nt >= 2) {
            const int u = s_top[(unsigned)(tid * 7919 + 13) % (unsigned)nt], v = s_top[(unsigned)(tid * 104729 + 71) % (unsigned)nt];
            if (u != v) {
              pr = 1;
              e = (int)((bm[(size_t)u * V.Wb + (v >> 6)] >> (v & 63)) & 1ULL);
            }
          }
          e = wave_sum_i32(e);
          pr = wave_sum_i32(pr);
          if (lane == 0) {
            s_edges[wave] = e;
            s_pairs[wave] = pr;
          }
          __syncthreads();
          e = pr = 0;
#pragma unroll
          for (int q = 0; q < HCA_THREADS / 64; ++q) {
            e += s_edges[q];
            pr += s_pairs[q];
          }
          if (10 * e < 9 * pr) f = 0;
        }
      }
      if (tid == 0) __hip_atomic_store(ctl_floor, ((unsigned)f << 1) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      my_floor = f;
    } else {
      // (one thread polls; a workgroup that gives up — 250 us — goes on without a floor: its rows are then lowered all the
      // way, which costs time and nothing else)
      unsigned f = 0;
      while (true) {
        if (tid == 0) s_floor = hca_load_u32(ctl_floor);
        __syncthreads();
        f = __builtin_amdgcn_readfirstlane(s_floor);
        const bool late = wall_clock64() - t_start > 25000ull;
        __syncthreads();  // (everybody has read the word before thread 0 writes it again)
        if ((f & 1u) || wg_any(late)) break;
        __builtin_amdgcn_s_sleep(8);
      }
      my_floor = (f & 1u) ? (int)(f >> 1) : 0;
    }
  }
  // my rows as neighbour lists — those whose degree reaches the floor
  if (wave == 0) {
    int run = 0;
    bool below = false;
    for (int base = 0; base < nown; base += 64) {
      const int rl = base + lane;
      const bool listed = rl < nown && mine[rl] >= my_floor;
      below = below || (rl < nown && !listed);
      const int d = listed ? nb_off[rl] : 0;
      int tot = 0;
      const int ex = wave_excl_scan_i32(d, &tot);
      if (rl < nown) nb_off[rl] = run + ex;
      run += tot;
    }
    if (lane == 0) nb_off[nown] = run;
    if (__any(below) && lane == 0) s_froze = 1;
  }
  __syncthreads();
  for (int rl = wave; rl < nown; rl += HCA_THREADS / 64) {
    if (mine[rl] < my_floor || nb_off[rl + 1] > pool_entries) continue;
    const u64* rowp = bm + (size_t)(r_lo + rl) * V.Wb;
    int run = nb_off[rl];
    for (int base = 0; base < W; base += 64) {
      const int wd = base + lane;
      u64 bits = wd < W ? rowp[wd] : 0;
      int tot = 0;
      int at = run + wave_excl_scan_i32(__popcll(bits), &tot);
      while (bits) {
        const int b = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        pool[at++] = (unsigned short)(wd * 64 + b);
      }
      run += tot;
    }
  }
  unsigned myver = 0;  // (thread 0) this workgroup's version counter
  bool froze_said = false;
  unsigned v0 = 0, E0 = 0;
  bool bump = true;       // values were stored (set-up) or lowered in the previous iteration: ver[w] has to follow once they have landed
  bool v0_valid = false;  // v0 / E0 were read before the snapshot of THIS iteration (only then may it claim a fixed point)
  int iter = 0;
  // Every row is evaluated at least once.  The other workgroups get that from their snapshot (it starts as "everything
  // moved"); workgroup 0's first snapshot is the one the floor's degree histogram was taken from, and if no other workgroup
  // ever lowers a value — every other vertex's degree IS its core number — nothing moves again and its rows kept their
  // DEGREES: a vertex 0 of degree 2 between two leaves came out with core number 2 (tests/gpu_fuzz.py seed 72: a
  // 64-correspondence pair in a batch group whose largest clique is an edge, the search then started from vertex 0
  // instead of the reference's highest-ranked one; round 6)
  bool first_pass = true;
  bool finished = false;
  __syncthreads();
#ifdef QTR_HCA_PROF
  unsigned prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // 10 ns ticks: [0] set-up, [1] snapshots, [2] rows, [3] protocol; counts: [4] iterations
  unsigned long long prof_last = t_start;            // that lowered something, [5] idle ones; [2] is wave 0 alone, [6] / [7] the wait for the other waves after it (lowering / idle)
#endif
  // the h-index of a row, given how to count: the largest t <= cv with count_ge(t) >= t (monotone in t).  Values drop in
  // small steps: probe cv, cv - 1, cv - 3, cv - 7 ... until one holds, then bisect the last gap.  Returns cv if it stands.
  auto h_index = [&](int cv, auto&& count_ge) __attribute__((always_inline)) -> int {
    if (count_ge(cv) >= cv) return cv;
    int bad = cv, good = 0;  // invariant: predicate holds at `good` (t = 0 always), fails at `bad`
    for (int step = 1; bad - step > 0; step <<= 1) {
      const int t = bad - step;
      if (count_ge(t) >= t) {
        good = t;
        break;
      }
      bad = t;
    }
    while (good + 1 < bad) {
      const int mid = (good + bad) >> 1;
      if (count_ge(mid) >= mid) good = mid;
      else bad = mid;
    }
    return good;
  };
  // a row of at most 64 S neighbours: their values into S registers per lane, a probe is S compares and a wave sum
  auto row_by_list = [&](int cv, int off, int deg, auto s_tag) __attribute__((always_inline)) -> int {
    constexpr int S = decltype(s_tag)::value;
    int xv[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {  // (unconditional reads — the last entry stands in past the end — so that the S index
      const int e = lane + 64 * k;   // reads, then the S value reads, are in flight together)
      const int x = vals[pool[off + min(e, deg - 1)]];
      xv[k] = e < deg ? x : 0;
    }
    return h_index(cv, [&](int th) __attribute__((always_inline)) {  // (th >= 1: the padding zeros never count)
      int cc = 0;
#pragma unroll
      for (int k = 0; k < S; ++k) cc += xv[k] >= th ? 1 : 0;
      return wave_sum_i32(cc);
    });
  };
  // One row (a whole wave).  Returns whether the value was lowered.
  auto evaluate = [&](int rl) __attribute__((always_inline)) -> bool {
    const int cv = __builtin_amdgcn_readfirstlane(mine[rl]);
    if (cv <= 0) return false;
    if (cv < my_floor) {  // below the floor: left alone (an upper bound, which is all anybody needs of it)
      if (!froze_said) {  // (noted in LDS, stored once per workgroup at the end: four thousand waves storing to one word that
        froze_said = true;  // every workgroup's snapshot polls the neighbour of took 45 us of everybody's time)
        if (lane == 0) s_froze = 1;
      }
      return false;
    }
    const int v = r_lo + rl;
    const int off = __builtin_amdgcn_readfirstlane(nb_off[rl]), end = __builtin_amdgcn_readfirstlane(nb_off[rl + 1]), deg = end - off;
    int h;
    if (end <= pool_entries) {
      if (deg <= 128) h = row_by_list(cv, off, deg, std::integral_constant<int, 2>{});
      else if (deg <= 384) h = row_by_list(cv, off, deg, std::integral_constant<int, 6>{});
      else if (deg <= 1024) h = row_by_list(cv, off, deg, std::integral_constant<int, 16>{});
      else
        h = h_index(cv, [&](int th) __attribute__((always_inline)) {  // longer lists are re-read for every probe
          int cc = 0;
          for (int e = lane; e < deg; e += 64) cc += vals[pool[off + e]] >= th ? 1 : 0;
          return wave_sum_i32(cc);
        });
    } else {  // no list: the bit row, from memory, walked for every probe
      const u64* rowp = bm + (size_t)v * V.Wb;
      h = h_index(cv, [&](int th) __attribute__((always_inline)) {
        int cc = 0;
        for (int wd = lane; wd < W; wd += 64) {
          u64 bits = rowp[wd];
          while (bits) {
            const int b = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            cc += vals[wd * 64 + b] >= th;
          }
        }
        return wave_sum_i32(cc);
      });
    }
    if (h == cv) return false;
    // (the snapshot keeps the old value of v: the next one then finds v moved, so this workgroup looks at its rows again)
    if (lane == 0) {
      mine[rl] = h;
      __hip_atomic_store(gvals + v, (unsigned short)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
  };
  for (; iter < HCA_MAXITER && !finished; ++iter) {
    HCA_MARK(iter == 0 ? 0 : 3);
    // 1. snapshot of all values (four per load)
    // (16-byte loads and two 8-byte loads in flight per thread were both tried: no faster)
    bool moved = false;
    for (int i = tid; i < (Lp >> 2); i += HCA_THREADS) {
      const u64 nv = hca_load_u64((const u64*)gvals + i);
      if (nv != ((u64*)vals)[i]) {
        ((u64*)vals)[i] = nv;
        moved = true;
      }
    }
    // (every load above has returned, so the value stores of the previous iteration — issued before them — have reached
    // the coherence point as well: only now may the version say so)
    // (the scout's floor, if it has one by now: thread 0's load rides with the snapshot's, the vote's barrier hands it round)
    if (has_scout && tid == 0) s_floor = hca_load_u32((const unsigned*)V.perm + HCA_CTL_FLOOR2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool any_moved = wg_any(moved) || first_pass;
    first_pass = false;
    if (has_scout) {
      const unsigned f2 = __builtin_amdgcn_readfirstlane(s_floor);
      if (f2 & 1u) my_floor = max(my_floor, (int)(f2 >> 1));
    }
    if (bump && tid == 0) __hip_atomic_store(ver + w, ++myver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bump = false;
    HCA_MARK(1);
    // 2. my rows, each counted afresh (one wave per row) — unless nothing at all moved
    bool changed = false;
    if (any_moved)
      for (int rl = wave; rl < nown; rl += HCA_THREADS / 64) changed |= evaluate(rl);
    // 3. something lowered: straight into the next snapshot (the version is bumped there, once the stores have landed)
    HCA_MARK(2);
    if (wg_any(changed)) {
      HCA_MARK(6);
      HCA_COUNT(4);
      bump = true;
      v0_valid = false;
      continue;
    }
    HCA_MARK(7);
    HCA_COUNT(5);
    // nothing changed.  A fixed point may only be claimed for a snapshot taken AFTER the version vector was read:
    // read it now (thread t reads ver[t]; E0 = its sum) and go round once more — with nothing moved that is one snapshot
    if (!v0_valid) {
      v0 = tid < NWG ? hca_load_u32(ver + tid) : 0u;
      const unsigned ws = (unsigned)wave_sum_i32((int)v0);
      if (lane == 0) s_sum[wave] = ws;
      __syncthreads();
      E0 = 0;
#pragma unroll
      for (int q = 0; q < HCA_THREADS / 64; ++q) E0 += s_sum[q];
      v0_valid = true;
      continue;
    }
    // did anybody publish while I was looking?
    {
      const unsigned v1 = tid < NWG ? hca_load_u32(ver + tid) : 0u;
      if (wg_any(v1 != v0)) {
        v0_valid = false;
        continue;
      }
    }
    // my rows are a fixed point of a snapshot taken at epoch E0: say so, and wait for the verdict
    if (tid == 0) __hip_atomic_store(done + w, E0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int verdict = 0;  // 1: somebody lowered a value, iterate again; 2: everybody idle at E0: fixed point; 3: gave up
    for (unsigned polls = 0; verdict == 0; ++polls) {
      const unsigned v1 = tid < NWG ? hca_load_u32(ver + tid) : 0u;
      const unsigned d1 = tid < NWG ? hca_load_u32(done + tid) : E0 + 1u;
      if (wg_any(v1 != v0)) verdict = 1;
      else if (!wg_any(d1 != E0 + 1u)) verdict = 2;
      // Bounded by the 100 MHz wall clock: a workgroup whose version is still 0 has not finished its set-up — 2 ms after
      // my own start that means it is not RESIDENT (launches sharing the device beyond what the host planned for, a
      // device with fewer usable compute units than it reports), and it may be waiting for the unit I occupy: give up,
      // the peeling workgroup runs instead.  Whatever else keeps the verdict away: 250 ms.
      else if ((polls & 15) == 15) {
        const unsigned long long waited = wall_clock64() - t_start;  // (per thread: the vote makes the decision uniform)
        // ... or somebody else has given up (its mark will never come: without this word the workgroups that did not see
        // the version-0 neighbour in time — it became resident a moment later — waited out the 250 ms)
        const bool abandoned = tid == 0 && hca_load_u32((const unsigned*)V.perm + HCA_CTL_FAILED) != 0u;
        if (wg_any(abandoned || waited > 25000000ull || (waited > 200000ull && tid < NWG && v1 == 0u))) verdict = 3;
      } else __builtin_amdgcn_s_sleep(16);
    }
    v0_valid = false;
    if (verdict == 2) finished = true;
    if (verdict == 3) {
      if (tid == 0) __hip_atomic_store((unsigned*)V.perm + HCA_CTL_FAILED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  if (tid == 0) {
    if (!finished) V.perm[HCA_CTL_FAILED] = 1;
    if (s_froze) V.perm[HCA_CTL_FROZE] = 1;
    atomicMax(&V.perm[HCA_CTL_ITERS], iter);
  }
#ifdef QTR_HCA_PROF
  HCA_MARK(3);
  if (tid == 0 && w < HCA_MAXWG)
    for (int q = 0; q < 8; ++q) g_hca_prof[w * 8 + q] = prof_acc[q];
#endif
  // my rows' final values
  __syncthreads();
  for (int rl = tid; rl < nown; rl += HCA_THREADS) V.core[r_lo + rl] = mine[rl];
}

// what k_kcore leaves behind besides the core numbers: edge total, largest core, the zeroed rank accumulator
template <bool EXT>
__global__ __launch_bounds__(1024) void k_hcore_finish(ViewExt<SolverView> x, SolverView one, int async_mode) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int L = V.L;
  if (L <= 0) return;
  SolverState* __restrict__ st = V.st;
  const int* __restrict__ flags = V.perm;
  if (async_mode ? flags[HCA_CTL_FAILED] != 0 : flags[HC_MAXIT - 1] != 0) {  // not converged: the peeling kernel takes over
    if (threadIdx.x == 0) st->pad[5] = 1;
    return;
  }
  __shared__ int s_max, s_edges, s_iters;
  if (threadIdx.x == 0) s_max = s_edges = s_iters = 0;
  __syncthreads();
  int mx = 0, es = 0;
  for (int v = threadIdx.x; v < L; v += 1024) {
    mx = max(mx, V.core[v]);
    es += V.deg[v];
    V.rankof[v] = 0;  // k_rank_partial accumulates into it
  }
  mx = wave_max_i32(mx);
  es = wave_sum_i32(es);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&s_max, mx);
    atomicAdd(&s_edges, es);
  }
  if (!async_mode && threadIdx.x < HC_MAXIT && flags[threadIdx.x]) atomicAdd(&s_iters, 1);
  if (async_mode && threadIdx.x == 0) s_iters = flags[HCA_CTL_ITERS] - 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    st->n_edges2 = s_edges;
    st->max_core = s_max;
    st->ub = s_max + 1;
    st->pad[0] = s_iters + 1;  // statistics: sweeps that ran
    st->pad[5] = 0;
  }
}

// Level-parallel variant for L <= 1280: peeling is a chain of hundreds of dependent rounds (one per occupied
// level, ~1 us each however few lanes they keep busy), but the k-cores themselves are independent of one
// another.  Workgroup k computes the k-core directly:
//     alive = {deg >= k};  repeat  alive = {v in alive : |N(v) & alive| >= k}  until nothing changes
// (a handful of sweeps) and writes its membership mask M[k].  Thread t owns vertices t, t+256, ... and keeps
// THEIR ROWS IN REGISTERS (VT rows of 4*VT words), so a sweep is one broadcast LDS read of the alive word,
// an AND, a popcount and an add per word — no staging of the matrix, no dynamic LDS, every workgroup of the
// launch co-resident.  The cores are nested, so core(v) = max{k : v in M[k]}; the workgroup that finishes
// next kernel reads every vertex's core number off its column.  Core numbers are unique, so this equals
// compute_cores' (Batagelj-Zaversnik) result.
#define KCL_VPT 5  // most vertices per thread: 256 * 5 = 1280
template <bool EXT, int VT>
__global__ __launch_bounds__(256) void k_kcore_levels(ViewExt<SolverView> x, SolverView one) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const u64* __restrict__ bm = V.bm;
  const int L = V.L, W = V.W;
  if ((int)blockIdx.x + 1 >= L) return;  // levels 1 .. L-1 of THIS pair (the grid is sized for the largest)
  const int* __restrict__ deg_in = V.deg;
  u64* __restrict__ M = V.adjP;
  constexpr int WT = 4 * VT;
  __shared__ u64 s_alive[2][4 * KCL_VPT];
  __shared__ int s_cnt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.x + 1;
  if (tid == 0) s_cnt = 0;
  if (tid < 4 * KCL_VPT) s_alive[0][tid] = s_alive[1][tid] = 0;
  __syncthreads();
  u32 amask = 0;  // bit i: my vertex i*256+tid is alive
  int mydeg[VT];
  {
    int c = 0;
#pragma unroll
    for (int i = 0; i < VT; ++i) {
      const int v = i * 256 + tid;
      mydeg[i] = (v < L) ? solver_degree(V, v) : -1;
      const bool a = mydeg[i] >= k;
      amask |= a ? (1u << i) : 0u;
      const u64 mk = __ballot(a);
      if (lane == 0) s_alive[0][i * 4 + wave] = mk;
      c += (lane == 0) ? __popcll(mk) : 0;
    }
    if (lane == 0 && c) atomicAdd(&s_cnt, c);
  }
  __syncthreads();
  int par = 0, sweeps = 0;
  if (s_cnt > k) {  // a k-core needs k+1 vertices
    u64 row[VT][WT];
#pragma unroll
    for (int i = 0; i < VT; ++i) {
      const u64* src = bm + (size_t)(i * 256 + tid) * V.Wb;
      const bool a = (amask >> i) & 1u;
#pragma unroll
      for (int w = 0; w < WT; ++w) row[i][w] = (a && w < W) ? src[w] : 0ULL;
    }
    while (true) {
      ++sweeps;
      int changed = 0;
#pragma unroll
      for (int i = 0; i < VT; ++i) {
        if ((amask >> i) & 1u) {
          int d = 0;
#pragma unroll
          for (int w = 0; w < WT; ++w) d += __popcll(row[i][w] & s_alive[par][w]);
          if (d < k) {
            amask &= ~(1u << i);
            changed = 1;
          }
        }
        const u64 mk = __ballot((amask >> i) & 1u);
        if (lane == 0) s_alive[par ^ 1][i * 4 + wave] = mk;
      }
      par ^= 1;
      if (!__syncthreads_or(changed)) break;
    }
  } else {
    __syncthreads();
    if (tid < 4 * KCL_VPT) s_alive[0][tid] = 0;
    __syncthreads();
  }
  if (tid < W) M[(size_t)k * W + tid] = s_alive[par][tid];
}

// Second half of the level-parallel path: every vertex's core number = the largest k whose mask holds it (the
// masks are nested), found with an 8-ary search over its column (7 independent probes per step) — a separate
// launch on purpose: on a multi-XCD part a device-scope fence inside the kernel (last-workgroup-done pattern)
// writes back and invalidates the L2s and cost more than the launch boundary that gives the same ordering for
// free.  The same workgroup then ranks the vertices and initialises the search: ONE workgroup of 1024 threads
// (L <= 1280 <= 2 vertices per thread); the ranks come from one stable counting-sort pass keyed by the core
// number, and thread 0 initialises the clique search — one launch instead of five (collect, memset, rank
// partial, rank finish, clique init).
template <bool EXT>
__global__ __launch_bounds__(1024) void k_kcore_collect_rank(ViewExt<SolverView> x, SolverView one, int first_batch) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const u64* __restrict__ M = V.adjP;
  const int L = V.L, W = V.W;
  if (L <= 0) return;
  const int K = L > 1 ? L - 1 : 1;  // levels k_kcore_levels wrote for this pair
  const int* __restrict__ deg_in = V.deg;
  int* __restrict__ core_out = V.core;
  int* __restrict__ perm = V.perm;
  int* __restrict__ Kp = V.Kp;
  SolverState* __restrict__ st = V.st;
  __shared__ __attribute__((aligned(16))) int s_core[2048 + 4];
  __shared__ int s_bin[2048];
  __shared__ int s_red[32], s_tot[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int esum = 0, cmax = 0;
  int myc[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int v = i * 1024 + tid;
    if (v < L) {
      const int dv = solver_degree(V, v);
      int lo = 0, hi = min(dv, K);  // invariant: member at lo (k = 0: everybody), not above hi
      const u64* col = M + (v >> 6);
      const int bv = v & 63;
      while (lo < hi) {
        const int span = hi - lo;
        int probe[7];
        u64 word[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          probe[q] = min(lo + max(1, (int)(((long long)span * (q + 1)) >> 3)), hi);
          word[q] = col[(size_t)probe[q] * W];
        }
        int nlo = lo, nhi = hi;
        bool cut = false;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          const bool in = (word[q] >> bv) & 1ULL;
          if (!cut) {
            if (in)
              nlo = probe[q];
            else {
              nhi = probe[q] - 1;
              cut = true;
            }
          }
        }
        lo = nlo;
        hi = nhi;
      }
      core_out[v] = lo;
      myc[i] = lo;
      cmax = max(cmax, lo);
      esum += dv;
    }
    if (v < 2048) s_core[v] = (v < L) ? myc[i] : 0x7fffffff;
  }
  cmax = wave_max_i32(cmax);
  esum = wave_sum_i32(esum);
  if (lane == 0) {
    s_red[wave] = cmax;
    s_red[16 + wave] = esum;
  }
  __syncthreads();
  if (tid == 0) {
    int mc = 0, es = 0;
    for (int w = 0; w < 16; ++w) {
      mc = max(mc, s_red[w]);
      es += s_red[16 + w];
    }
    s_tot[0] = mc;
    s_tot[1] = es;
  }
  __syncthreads();
  // ranks in the (core, id) order = one stable counting-sort pass keyed by the core number: histogram, exclusive
  // scan over the (at most 2048) core values, then ONE wavefront walks the vertices in id order, 64 at a time,
  // ranking equal keys inside a chunk with ballot match masks (as the radix scatter does)
  for (int c = tid; c < 2048; c += 1024) s_bin[c] = 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int v = i * 1024 + tid;
    if (v < L) atomicAdd(&s_bin[myc[i]], 1);
  }
  __syncthreads();
  {
    const int c0 = s_bin[2 * tid], c1 = s_bin[2 * tid + 1];
    int tot;
    const int ex = wave_excl_scan_i32(c0 + c1, &tot);
    if (lane == 63) s_red[wave] = tot;  // (cmax / esum partials were consumed above)
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_red[w];
    s_bin[2 * tid] = woff + ex;
    s_bin[2 * tid + 1] = woff + ex + c0;
  }
  __syncthreads();
  if (wave == 0) {
    for (int base = 0; base < L; base += 64) {
      const int v = base + lane;
      const bool valid = v < L;
      const int c = valid ? s_core[v] : 0;
      u64 m = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 11; ++bit) {
        const bool one = (c >> bit) & 1;
        const u64 bb = __ballot(valid && one);
        m &= one ? bb : ~bb;
      }
      int r = 0;
      if (valid) r = s_bin[c] + __popcll(m & lanemask_lt());
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads above complete before the updates below
      if (valid && (m & lanemask_lt()) == 0) s_bin[c] += __popcll(m);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (valid) {
        perm[r] = v;
        Kp[r] = c + 1;
      }
    }
  }
  if (tid == 0) {
    const int mc = s_tot[0], es = s_tot[1];
    st->max_core = mc;
    st->ub = mc + 1;
    st->n_edges2 = es;
    // k_clique_init
    st->mc = 0;
    st->best_r = -1;
    st->pos = L - 1;
    st->done = (L <= 0) ? 1 : 0;
    st->batch = first_batch;
    st->t0 = 0;
    st->rounds = 0;
  }
}

// K12b: rank of every vertex in the (core, id) ascending order; perm[rank] = vertex; Kp[rank] = core+1
// (PMC's "kcore" value).  This single order serves both as the outer start order of the heuristic
// (traversed from the back) and as the greedy pick order (highest rank = max (K, id)).
template <bool EXT>
__global__ __launch_bounds__(256) void k_rank_partial(ViewExt<SolverView> x, SolverView one) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int* __restrict__ core = V.core;
  const int L = V.L;
  if ((int)blockIdx.x * 256 >= L) return;
  int* __restrict__ rankof = V.rankof;
  // grid (ceil(L/256), slices): thread = vertex v, blockIdx.y = slice of the comparison range
  __shared__ int tile[1024];
  const int v = blockIdx.x * 256 + threadIdx.x;
  const int c = v < L ? core[v] : 0;
  const int per = ((L + gridDim.y - 1) / gridDim.y + 1023) / 1024 * 1024;
  const int u0 = blockIdx.y * per, u1 = min(L, u0 + per);
  int r = 0;
  for (int base = u0; base < u1; base += 1024) {
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += 256) tile[t] = (base + t < u1) ? core[base + t] : 0x7fffffff;
    __syncthreads();
    const int lim = min(1024, u1 - base);
    for (int t = 0; t < lim; ++t) {
      const int cu = tile[t], u = base + t;
      r += (cu < c) || (cu == c && u < v);
    }
  }
  if (v < L && r) atomicAdd(&rankof[v], r);
}
template <bool EXT>
__global__ __launch_bounds__(256) void k_rank_finish(ViewExt<SolverView> x, SolverView one) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int* __restrict__ core = V.core;
  const int L = V.L;
  const int* __restrict__ rankof = V.rankof;
  int* __restrict__ perm = V.perm;
  int* __restrict__ Kp = V.Kp;
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v < L) {
    const int r = rankof[v];
    perm[r] = v;
    Kp[r] = core[v] + 1;
  }
}

// K12b': the same ranks by a stable counting sort on the core number, one workgroup, one launch (the quadratic count
// above, its finish and k_clique_init took 29 + 5 + 5 us at L = 5000).  Wave w owns a contiguous block of vertex ids:
// per-wave counts per core value in LDS, exclusive prefix over (core, wave) = where wave w's first vertex of core c goes;
// then every wave walks its ids 64 at a time in id order and ranks the lanes that share a core value among themselves
// (ballot per distinct value), so equal cores keep ascending ids — the (core, id) order.  Core numbers above RS_BINS - 2
// (a clique of more than a thousand members) take the quadratic count inside this kernel.
#define RS_THREADS 1024
#define RS_BINS 1024
template <bool EXT>
__global__ __launch_bounds__(RS_THREADS) void k_rank_sort(ViewExt<SolverView> x, SolverView one, int after_async,
                                                          int kcore_mode /* 0: the core numbers are there; 1: peel first;
                                                                            2: peel if k_hcore_async gave up */,
                                                          int kc_gqueue, int kc_lds_bitmap) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int L = V.L;
  SolverState* st = V.st;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int s_nb, s_cf;
  // the peeling kernel's work, without its launch (RS_THREADS = its 1024 threads; the dynamic LDS is sized for both)
  const bool gave_up = after_async && L > 0 && V.perm[HCA_CTL_FAILED] != 0;  // (uniform)
  if (L > 0 && (kcore_mode == 1 || (kcore_mode == 2 && gave_up))) {
    d_kcore(V, kc_gqueue, kc_lds_bitmap);
    __syncthreads();
  }
  if (after_async && L > 0) {
    // what a k_hcore_finish launch used to do behind k_hcore_async: edge total, largest core, statistics (when the
    // iteration gave up, k_kcore has just run and left them)
    __shared__ int s_red[2][RS_THREADS / 64];
    const int failed = V.perm[HCA_CTL_FAILED], iters = V.perm[HCA_CTL_ITERS];  // (perm is overwritten further down)
    // the floor counts only if some value was in fact left standing below it (and the iteration itself went through)
    const int floor_w = V.perm[HCA_CTL_FLOOR], froze = V.perm[HCA_CTL_FROZE], floor2_w = V.perm[HCA_CTL_FLOOR2];
    int mx = 0, es = 0;
    if (!failed)
      for (int v = tid; v < L; v += RS_THREADS) {
        mx = max(mx, V.core[v]);
        es += V.deg[v];
      }
    mx = wave_max_i32(mx);
    es = wave_sum_i32(es);
    if (lane == 0) {
      s_red[0][wave] = mx;
      s_red[1][wave] = es;
    }
    __syncthreads();
    if (tid == 0) {
      if (!failed) {
        mx = 0, es = 0;
        for (int q = 0; q < RS_THREADS / 64; ++q) {
          mx = max(mx, s_red[0][q]);
          es += s_red[1][q];
        }
        st->n_edges2 = es;
        st->max_core = mx;
        st->ub = mx + 1;
        st->pad[0] = iters;  // statistics: iterations of the slowest workgroup
        s_nb = mx + 1;
        s_cf = froze ? max(floor_w >> 1, floor2_w >> 1) : 0;  // (the bet's floor or the scout's: whatever anybody stopped below)
      } else {
        s_nb = st->max_core + 1;
        s_cf = 0;  // (the peeling workgroup's numbers are exact)
      }
    }
    __syncthreads();
  } else if (tid == 0) {
    s_nb = L > 0 ? st->max_core + 1 : 0;
    s_cf = 0;
  }
  // A floor F > 0 (k_hcore_async): core numbers below F are upper bounds, not values, so the search may not look at those
  // vertices at all — it starts as if a clique of F vertices were already known: mc = F with no clique behind it, t0 = the
  // first rank with K > F.  Every test it makes (K[r] > mc, the candidates' K > mc, |P| > mc) then involves exact numbers
  // only.  Against the search that starts from mc = 0 on exact numbers: that one may pick up cliques of at most F vertices
  // on the way (a descent that leaves the K > F region ends in a clique holding a vertex with K <= F, i.e. of at most F
  // vertices), keeps mc_exact <= mc_here, and whatever this search accepts (size > mc_here) it accepts too, with the same
  // members (the members' K exceed the bound in both, the picks go from the top rank down, and nothing below can extend a
  // clique beyond its own K) — after which the two agree for good.  The converse needs |P| > mc to hold here whenever it
  // holds there: d_clique_scan notes a start that passed the size test and failed that one (`tainted`), and with a taint
  // or without any accepted clique the stage is run again without a floor (`redo_cores`, solver_continue): the result is
  // either identical to the exact search's or not used.
  if (tid == 0) {  // what k_clique_init sets
    st->mc = s_cf;
    st->core_floor = s_cf;
    st->best_r = -1;
    st->pos = L - 1;
    st->done = (L <= 0) ? 1 : 0;
    st->batch = 1;
    st->t0 = 0;
    st->rounds = 0;
  }
  if (L <= 0) return;
  const int* __restrict__ core = V.core;
  int* __restrict__ perm = V.perm;
  int* __restrict__ Kp = V.Kp;
  extern __shared__ __attribute__((aligned(16))) int rs_lds[];  // [16][RS_BINS] counts, then offsets
  __shared__ int s_wtot[RS_THREADS / 64];
  __syncthreads();
  const int NB = s_nb;  // core values 0 .. max_core
  const int cf = s_cf;  // t0 = the number of vertices with core number < cf
  if (cf >= NB && tid == 0) st->t0 = L;
  // Core numbers above RS_BINS - 2 (a clique of more than a thousand members: half of the correspondences of a scan pair
  // registered against a near copy of itself are inliers — bench.py connected_leg.l5k) take TWO stable passes of the same
  // counting sort, ten bits of the core number each (an LSD radix sort: the second pass walks the first one's sequence, kept
  // in V.rankof); until round 6 they took a quadratic count in this one workgroup: 1.9 ms at L = 5910.
  const int npass = NB > RS_BINS ? 2 : 1;
  if (npass > 1 && cf > 0 && cf < NB) {  // t0 = the number of vertices below the floor (the one-pass form reads it off its offsets)
    int below = 0;
    for (int v = tid; v < L; v += RS_THREADS) below += core[v] < cf ? 1 : 0;
    below = wave_sum_i32(below);
    if (lane == 0 && below) atomicAdd(&st->t0, below);
  }
  int* __restrict__ seq = V.rankof;  // (L ints; not in use before the permutation)
  const int chunk = (((L + 15) / 16) + 63) & ~63;  // sequence positions per wave, a multiple of 64
  const int id0 = wave * chunk, id1 = min(L, id0 + chunk);
  int* mycnt = rs_lds + wave * RS_BINS;
  // (the one-pass case — every graph with core numbers below 1023 — is compiled on its own: no sequence array, no digit
  // extraction, no barrier in front; as one loop over `npass` it cost that case 1.5 us of its 10)
  auto pass = [&](auto single_tag, const int ps) __attribute__((always_inline)) {
    constexpr bool SINGLE = decltype(single_tag)::value;
    const int shift = SINGLE ? 0 : 10 * ps;
    const bool from_seq = !SINGLE && ps > 0, last = SINGLE || ps == npass - 1;
    const int NBd = SINGLE ? NB : (ps == 0 ? RS_BINS : ((NB - 1) >> 10) + 1);  // digit values of this pass
    if (!SINGLE) __syncthreads();
    for (int i = tid; i < 16 * RS_BINS; i += RS_THREADS) rs_lds[i] = 0;
    __syncthreads();
    for (int p = id0 + lane; p < id1; p += 64) {
      const int v = from_seq ? seq[p] : p;
      atomicAdd(&mycnt[(core[v] >> shift) & (RS_BINS - 1)], 1);
    }
    __syncthreads();
    // thread d: exclusive prefix over the waves for digit value d, total of d
    int total = 0;
    if (tid < NBd) {
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        const int n = rs_lds[w * RS_BINS + tid];
        rs_lds[w * RS_BINS + tid] = total;
        total += n;
      }
    }
    // exclusive prefix of the totals over the digit values (one per thread)
    int wtot;
    const int ex = wave_excl_scan_i32(total, &wtot);
    if (lane == 0) s_wtot[wave] = wtot;
    __syncthreads();
    int wbase = 0;
    for (int q = 0; q < wave; ++q) wbase += s_wtot[q];
    if (tid < NBd) {
      const int start = wbase + ex;
      if (npass == 1 && cf > 0 && tid == cf) st->t0 = start;
#pragma unroll
      for (int w = 0; w < 16; ++w) rs_lds[w * RS_BINS + tid] += start;
    }
    __syncthreads();
    // placement: lanes of equal digit rank among themselves in sequence order; the wave's running offsets are private
    int nbits = 1;
    while ((1 << nbits) < NBd) ++nbits;
    for (int p0 = id0; p0 < id1; p0 += 64) {
      const int p = p0 + lane;
      const bool valid = p < id1;
      const int v = valid ? (from_seq ? seq[p] : p) : 0;
      const int c = valid ? core[v] : -1;
      const int d = valid ? ((c >> shift) & (RS_BINS - 1)) : -1;
      // m = the lanes holding my digit, bit by bit of the value (a ballot per bit: ~3 instructions each, whatever the
      // number of distinct values in the chunk — the loop over distinct values this replaces ran up to 64 times per chunk
      // and was most of the kernel)
      u64 m = __ballot(valid);
      for (int bit = 0; bit < nbits; ++bit) {
        const bool one = (d >> bit) & 1;
        const u64 b = __ballot(valid && one);
        m &= one ? b : ~b;
      }
      if (valid) {
        const int off = mycnt[d];
        const int pos = off + __popcll(m & lanemask_lt());
        // (the wave's LDS operations complete in program order: every lane has read its offset before a leader moves it)
        if ((m & lanemask_lt()) == 0) mycnt[d] = off + __popcll(m);
        if (last) {
          perm[pos] = v;
          Kp[pos] = c + 1;
        } else {
          seq[pos] = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next chunk reads what the leaders stored
    }
    if (!SINGLE) __threadfence_block();  // (the next pass reads seq[] written by other waves of this workgroup)
  };
  if (npass == 1) {
    pass(std::true_type{}, 0);
  } else {
    pass(std::false_type{}, 0);
    pass(std::false_type{}, 1);
  }
}

// K12c: adjacency in rank labels: adjP[r][s] = adj[perm[r]][perm[s]].  One workgroup per PM_ROWS output rows: the source
// rows are staged in LDS, a lane looks perm[s] up ONCE and probes its bit in every staged row (one workgroup per row
// re-read the whole perm array for each row: L^2 * 4 bytes of L2 traffic, 257 us at L = 20000), a ballot per row builds
// the output words.
// The same permutation by SCATTER, for L <= 32768: a row of the consistency graph has a few hundred neighbours out of
// thousands of columns, so instead of asking for every output bit (r, s) "is (perm r, perm s) an edge?" — L bit tests per
// row, 21 us at L = 5000 — the set bits of row perm[r] are walked and each lands at its neighbour's rank, read from an
// inverse of perm that every workgroup keeps in LDS (16-bit entries).  One wave per row, PM2_ROWS rows per workgroup.
#define PM2_ROWS 8
#define PM2_MAXL 32768
template <bool EXT>
__global__ __launch_bounds__(256) void k_permute_scatter(ViewExt<SolverView> x, SolverView one) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const u64* __restrict__ bm = V.bm;
  const int* __restrict__ perm = V.perm;
  const int L = V.L, W = V.W;
  u64* __restrict__ adjP = V.adjP;
  const int r0 = blockIdx.x * PM2_ROWS;
  if (r0 >= L) return;
  // under a floor (k_rank_sort) nobody ever reads a row below the first rank above it — starts, candidates and the exact
  // search's roots all have K > mc >= floor: those rows are left as they are (down to a multiple of 64: the exact search
  // stages rows from the word boundary)
  if (V.st->core_floor > 0 && r0 + PM2_ROWS <= (V.st->t0 & ~63)) return;
  extern __shared__ u64 pm2_lds[];  // [4][W] the row a wave is assembling, then the ranks
  const int lane = qk_lane(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u64* outw = pm2_lds + (size_t)wave * W;
  unsigned short* rk = (unsigned short*)(pm2_lds + (size_t)4 * W);
  for (int s0 = threadIdx.x; s0 < L; s0 += 256) rk[perm[s0]] = (unsigned short)s0;
  __syncthreads();
  for (int i = wave; i < PM2_ROWS && r0 + i < L; i += 4) {
    const int r = r0 + i;
    const u64* __restrict__ rowp = bm + (size_t)perm[r] * V.Wb;
    for (int w = lane; w < W; w += 64) outw[w] = 0;
    for (int w = lane; w < W; w += 64) {
      u64 bits = rowp[w];
      while (bits) {
        const int b = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        const unsigned q = rk[w * 64 + b];
        atomicOr((unsigned*)outw + (q >> 5), 1u << (q & 31));
      }
    }
    // (the wave's LDS operations complete in program order: its reads below see every lane's atomics above)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int w = lane; w < W; w += 64) adjP[(size_t)r * W + w] = outw[w];
  }
}

#define PM_ROWS 8
template <bool EXT>
__global__ __launch_bounds__(256) void k_permute(ViewExt<SolverView> x, SolverView one) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const u64* __restrict__ bm = V.bm;
  const int* __restrict__ perm = V.perm;
  const int L = V.L, W = V.W;
  u64* __restrict__ adjP = V.adjP;
  extern __shared__ u64 prow[];  // [PM_ROWS][W]
  const int r0 = blockIdx.x * PM_ROWS;
  if (r0 >= L) return;
  const int nr = min(PM_ROWS, L - r0);
  for (int e = threadIdx.x; e < nr * W; e += 256) {
    const int i = e / W, w = e - i * W;
    prow[e] = bm[(size_t)perm[r0 + i] * V.Wb + w];
  }
  __syncthreads();
  const int lane = qk_lane(), wave = threadIdx.x >> 6;
  for (int w = wave; w < W; w += 4) {
    const int s = w * 64 + lane;
    const int u = s < L ? perm[s] : -1;
    const int uw = u >> 6;
    const u64 ub = 1ULL << (u & 63);
    u64 mine = 0;  // lane i keeps row i's word
#pragma unroll
    for (int i = 0; i < PM_ROWS; ++i) {
      const bool bit = (i < nr) && (u >= 0) && (prow[i * W + uw] & ub);
      const u64 word = __ballot(bit);
      if (lane == i) mine = word;
    }
    if (lane < nr) adjP[(size_t)(r0 + lane) * W + w] = mine;
  }
}

// K12d: greedy clique growth (pmc_heu::branch) for a batch of start vertices, one wavefront each.
// The candidate set is a bitset spread over the wave (word w lives in lane w%64, slot w/64); a pick is
// the highest set bit (= max (K, id)), an intersection is a coalesced row AND.  Starts are speculated
// with the clique bound mc0 known at batch start; k_clique_scan then replays PMC's sequential
// acceptance rules over the batch, which makes the outcome identical to a single-threaded run.
template <int WPL>
__device__ __forceinline__ int greedy_descent(const u64* __restrict__ adjP, int W, int r, int t0, int lane,
                                              int* __restrict__ picks /* may be null */) {
  u64 cur[WPL];
#pragma unroll
  for (int s = 0; s < WPL; ++s) {
    const int w = s * 64 + lane;
    u64 x = (w < W) ? adjP[(size_t)r * W + w] : 0ULL;
    const int lo = w * 64;
    if (lo + 63 < t0)
      x = 0;
    else if (lo < t0)
      x &= ~((1ULL << (t0 - lo)) - 1ULL);
    cur[s] = x;
  }
  int depth = 1;
  while (true) {
    // highest set bit of the wave-distributed bitset: ballot per slot (highest slot first), then scalar
    // reads of the winning lane's word — no cross-lane reduction network on the dependent chain
    int u = -1;
#pragma unroll
    for (int s = WPL - 1; s >= 0; --s) {
      if (u < 0) {
        const u64 nz = __ballot(cur[s] != 0);
        if (nz) {
          const int l = 63 - __clzll((long long)nz);
          const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)cur[s], l);
          const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(cur[s] >> 32), l);
          const u64 word = ((u64)hi << 32) | lo;
          u = (s * 64 + l) * 64 + 63 - __clzll((long long)word);
        }
      }
    }
    if (u < 0) break;
    if (picks && lane == 0) picks[depth - 1] = u;
    ++depth;
#pragma unroll
    for (int s = 0; s < WPL; ++s) {
      const int w = s * 64 + lane;
      if (w < W) cur[s] &= adjP[(size_t)u * W + w];
    }
  }
  return depth;
}

// W <= 64 (L <= 4096), matrix in LDS: the whole candidate set is one word per lane, so a pick is
//   v_cmp (ballot) -> s_flbit -> 2 x v_readlane -> s_flbit -> address -> ds_read_b64 -> v_and
// with nothing else on the dependent chain: no slot loop, and the picks are parked in LDS (every lane writes the
// same word: no exec-mask switching) and copied out coalesced at the end.
__device__ __forceinline__ int greedy_descent_lds1(const u64* __restrict__ rows, int W, int r, int t0, int lane,
                                                   int* __restrict__ picks_lds, int* __restrict__ picks_out) {
  const int wl = min(lane, W - 1);
  u64 cur = (lane < W) ? rows[(size_t)r * W + lane] : 0ULL;
  {
    const int lo = lane * 64;
    if (lo + 63 < t0)
      cur = 0;
    else if (lo < t0)
      cur &= ~((1ULL << (t0 - lo)) - 1ULL);
  }
  const u64* col = rows + wl;
  int depth = 1;
  while (true) {
    const u64 nz = __ballot(cur != 0);
    if (nz == 0) break;
    const int l = 63 - __clzll((long long)nz);
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)cur, l);
    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(cur >> 32), l);
    const u64 word = ((u64)hi << 32) | lo;
    const int u = l * 64 + 63 - __clzll((long long)word);
    picks_lds[depth - 1] = u;
    ++depth;
    cur &= col[(size_t)u * W];  // lanes >= W hold 0 and stay 0
  }
  // depth - 1 picks, copied out coalesced
  for (int i = lane; i < depth - 1; i += 64) picks_out[i] = picks_lds[i];
  return depth;
}

__device__ __forceinline__ int greedy_dispatch(const u64* adjP, int W, int r, int t0, int lane, int* picks) {
  if (W <= 128) return greedy_descent<2>(adjP, W, r, t0, lane, picks);
  if (W <= 256) return greedy_descent<4>(adjP, W, r, t0, lane, picks);
  return greedy_descent<8>(adjP, W, r, t0, lane, picks);  // W <= 512  (L <= 32768)
}

// Round 0 of the heuristic on a matrix that does not fit LDS: ONE start, whose descent is a chain of |clique| dependent
// steps "highest set bit of the candidate set, AND with its row" (an L2 round trip each, and ~40 instructions of a lone
// wavefront at ~8 clocks apiece: 106 us for a clique of 250 at L = 5000, 305 us at L = 20000).  Here the whole workgroup
// takes the picks a word at a time:
//   1. the top non-empty word of the candidate set holds the next up-to-64 candidates; their rows are fetched into LDS
//      together (one round trip);
//   2. which of them does the descent pick?  Candidate b (a bit position of that word) is picked iff it is adjacent to
//      every PICKED candidate above it — a question about one 64-bit word per candidate (its row's word at the same
//      position).  Lane b iterates  P <- { b : no picked bit above b is missing from row_b }  from P = all candidates;
//      the status of the highest candidate is final after one pass, of the next after two, ..., and candidates from one
//      clique settle in two or three passes (one ballot each) instead of one dependent step per member;
//   3. the picked rows are ANDed into the candidate set by all threads (a word per thread), the picks are written out
//      in descending order.  Every other set bit lies below the whole word, so this is exactly greedy_descent's sequence.
__device__ __forceinline__ void d_clique_scan(const SolverView& V, int next_batch);  // (below)
#define CF_THREADS 512
#define CF_LDS_BYTES (144 * 1024)
template <bool EXT>
__global__ __launch_bounds__(CF_THREADS) void k_clique_first(ViewExt<SolverView> x, SolverView one, int then_scan) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const u64* __restrict__ adjP = V.adjP;
  const int L = V.L, W = V.W;
  if (L <= 0) return;
  const SolverState* st = V.st;
  if (st->done) return;
  extern __shared__ __attribute__((aligned(16))) u64 cf_lds[];
  u64* cur = cf_lds;         // [W] the candidate set
  u64* rowbuf = cf_lds + W;  // [K][W] rows of the candidates of the current word
  __shared__ int s_top, s_np, s_idx[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* __restrict__ picks = V.picks_buf;
  const int r = st->pos, t0 = st->t0;  // batch == 1: the single start of round 0
  int depth = 1;
  // Under a floor (k_rank_sort) the candidates are the few hundred ranks from t0 up: a window of Wn = W - (t0 >> 6) words.
  // When the rows of the whole window fit LDS (64 Wn rows of Wn words: Wn <= 16) they are fetched ONCE — one round trip
  // instead of one per word of candidates — and wave 0 runs the whole descent on them alone, no barrier in it: per word
  // of candidates the picked subset as below, then every candidate of the lower words asks ITS OWN row whether it holds
  // all the picks (the matrix is symmetric: one LDS read and a ballot per word instead of an AND of every picked row).
  const int w0 = st->core_floor > 0 ? (t0 >> 6) : 0, Wn = W - w0;
  if (r >= 0 && V.Kp[r] > st->mc && Wn <= 16) {  // (uniform over the workgroup)
    u64* rows = cf_lds;                              // [64 Wn][Wn]: row of rank 64 w0 + i, words w0 ..
    u64* curw = cf_lds + (size_t)64 * Wn * Wn;       // [Wn] the candidate set
    const int base = w0 * 64;
    for (int e = tid; e < 64 * Wn * Wn; e += CF_THREADS) {
      const int i = e / Wn, w = e - i * Wn;
      rows[e] = (base + i < L) ? adjP[(size_t)(base + i) * W + w0 + w] : 0ULL;
    }
    __syncthreads();
    if (wave == 0) {
      if (lane < Wn) {
        u64 xw = rows[(size_t)(r - base) * Wn + lane];
        const int lo = (w0 + lane) * 64;
        if (lo + 63 < t0)
          xw = 0;
        else if (lo < t0)
          xw &= ~((1ULL << (t0 - lo)) - 1ULL);
        curw[lane] = xw;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const u64 above = lane < 63 ? ~((2ULL << lane) - 1ULL) : 0ULL;
      for (int tw = Wn - 1; tw >= 0; --tw) {
        const u64 cand = curw[tw];  // (uniform: every lane reads the same word)
        if (cand == 0) continue;
        const bool isc = (cand >> lane) & 1ULL;
        const u64 arow = isc ? rows[(size_t)(tw * 64 + lane) * Wn + tw] : 0ULL;
        u64 P = cand;
        for (int pass = 0; pass < 64; ++pass) {
          const u64 Pn = __ballot(isc && (P & above & ~arow) == 0);
          if (Pn == P) break;
          P = Pn;
        }
        if ((P >> lane) & 1ULL) picks[depth - 1 + __popcll(lane < 63 ? (P >> (lane + 1)) : 0ULL)] = base + tw * 64 + lane;
        depth += __popcll(P);
        for (int w = tw - 1; w >= 0; --w) {
          const u64 cw = curw[w];
          if (cw == 0) continue;
          bool alive = (cw >> lane) & 1ULL;
          const u64 rw = alive ? rows[(size_t)(w * 64 + lane) * Wn + tw] : 0ULL;
          alive = alive && (rw & P) == P;
          const u64 nb = __ballot(alive);
          if (lane == 0) curw[w] = nb;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next word reads what lane 0 stored
      }
    }
  } else
  if (r >= 0 && V.Kp[r] > st->mc) {  // (uniform over the workgroup)
    const int K = max(1, min(64, (int)((CF_LDS_BYTES / 8 - W - (CF_THREADS / 64) * W) / W)));
    for (int w = tid; w < W; w += CF_THREADS) {
      u64 xw = adjP[(size_t)r * W + w];
      const int lo = w * 64;
      if (lo + 63 < t0)
        xw = 0;
      else if (lo < t0)
        xw &= ~((1ULL << (t0 - lo)) - 1ULL);
      cur[w] = xw;
    }
    if (tid == 0) s_top = -1;
    __syncthreads();
    while (true) {
      // 1. the top non-empty word
      {
        int mine = -1;
        for (int w = tid; w < W; w += CF_THREADS)
          if (cur[w] != 0) mine = w;
        mine = wave_max_i32(mine);
        if (lane == 0 && mine >= 0) atomicMax(&s_top, mine);
      }
      __syncthreads();
      const int topw = s_top;
      if (topw < 0) break;
      const u64 word = cur[topw];
      u64 cand = word;  // its highest (up to K) set bits
      if (__popcll(word) > K) {
        u64 xw = word;
        cand = 0;
        for (int q = 0; q < K; ++q) {
          const u64 b = 1ULL << (63 - __clzll((long long)xw));
          cand |= b;
          xw &= ~b;
        }
      }
      // row index of a candidate = the number of candidate bits above it; wave g fetches the rows with index = g (mod 8)
      {
        const bool mine = ((cand >> lane) & 1ULL) &&
                          (__popcll(lane < 63 ? (cand >> (lane + 1)) : 0ULL) & (CF_THREADS / 64 - 1)) == wave;
        u64 m = __ballot(mine);
        // all of this wave's rows (at most 64 / 8 = 8) are requested before any is stored: written as a loop of
        // load-then-store per row, every row waited for the one before — eight memory round trips per round
        auto fetch_rows = [&](auto wpl_tag) __attribute__((always_inline)) {
          constexpr int WPL = decltype(wpl_tag)::value;  // row words per lane held in registers
          u64 v[8][WPL];
          int dst[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            dst[q] = -1;
            if (m) {  // (uniform)
              const int b = 63 - __clzll((long long)m);
              m &= ~(1ULL << b);
              dst[q] = __popcll(b < 63 ? (cand >> (b + 1)) : 0ULL);
              const u64* __restrict__ rp = adjP + (size_t)(topw * 64 + b) * W;
#pragma unroll
              for (int e = 0; e < WPL; ++e) v[q][e] = (lane + 64 * e < W) ? rp[lane + 64 * e] : 0;
            }
          }
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (dst[q] >= 0) {
#pragma unroll
              for (int e = 0; e < WPL; ++e)
                if (lane + 64 * e < W) rowbuf[(size_t)dst[q] * W + lane + 64 * e] = v[q][e];
            }
        };
        if (W <= 128) fetch_rows(std::integral_constant<int, 2>{});       // L <= 8192
        else if (W <= 320) fetch_rows(std::integral_constant<int, 5>{});  // L <= 20480
        while (m) {  // (longer rows, or more than eight rows for this wave)
          const int b = 63 - __clzll((long long)m);
          m &= ~(1ULL << b);
          const int i = __popcll(b < 63 ? (cand >> (b + 1)) : 0ULL);
          const u64* __restrict__ rp = adjP + (size_t)(topw * 64 + b) * W;
          for (int w = lane; w < W; w += 64) rowbuf[(size_t)i * W + w] = rp[w];
        }
      }
      __syncthreads();
      // 2. the picked subset (wave 0, lane = bit position)
      if (wave == 0) {
        const bool isc = (cand >> lane) & 1ULL;
        const int ib = __popcll(lane < 63 ? (cand >> (lane + 1)) : 0ULL);
        const u64 arow = isc ? rowbuf[(size_t)ib * W + topw] : 0ULL;
        const u64 above = lane < 63 ? ~((2ULL << lane) - 1ULL) : 0ULL;
        u64 P = cand;
        for (int pass = 0; pass < 64; ++pass) {
          const u64 Pn = __ballot(isc && (P & above & ~arow) == 0);
          if (Pn == P) break;
          P = Pn;
        }
        if ((P >> lane) & 1ULL) {
          const int k = __popcll(lane < 63 ? (P >> (lane + 1)) : 0ULL);  // k-th pick of this word, descending
          picks[depth - 1 + k] = topw * 64 + lane;
          s_idx[k] = ib;
        }
        if (lane == 0) {
          s_np = __popcll(P);
          s_top = -1;  // for the next round's maximum
        }
      }
      __syncthreads();
      const int np = s_np;
      depth += np;
      // 3. AND the picked rows into the candidate set: wave g folds the picks k = g (mod 8) over all words, the eight
      //    partial results meet in LDS
      {
        u64* part = rowbuf + (size_t)K * W;  // [8][W]
        for (int w = lane; w < W; w += 64) {
          u64 acc = ~0ULL;
          for (int k = wave; k < np; k += CF_THREADS / 64) acc &= rowbuf[(size_t)s_idx[k] * W + w];
          part[(size_t)wave * W + w] = acc;
        }
        __syncthreads();
        for (int w = tid; w < W; w += CF_THREADS) {
          u64 acc = cur[w];
#pragma unroll
          for (int g = 0; g < CF_THREADS / 64; ++g) acc &= part[(size_t)g * W + w];
          cur[w] = acc;
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0) V.gsz[0] = (r >= 0 && V.Kp[r] > st->mc) ? depth : 0;
  // the round's replay (was its own launch): one workgroup wrote everything it reads, a barrier is all it takes
  if (then_scan) {
    __syncthreads();
    if (wave == 0) d_clique_scan(V, then_scan);
  }
}

template <bool EXT>
__global__ __launch_bounds__(256) void k_clique_init(ViewExt<SolverView> x, SolverView one) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int L = V.L;
  SolverState* st = V.st;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    st->mc = 0;
    st->best_r = -1;
    st->pos = L - 1;
    st->done = (L <= 0) ? 1 : 0;
    st->batch = 1;
    // t0 = first rank with Kp > 0: every vertex has K >= 1
    st->t0 = 0;
    st->rounds = 0;
  }
}

template <bool EXT>
__global__ __launch_bounds__(256) void k_clique_batch(ViewExt<SolverView> x, SolverView one) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const u64* __restrict__ adjP = V.adjP;
  const int* __restrict__ Kp = V.Kp;
  const int L = V.L, W = V.W;
  if (L <= 0) return;
  const SolverState* __restrict__ st = V.st;
  int* __restrict__ gsz = V.gsz;
  int* __restrict__ picks_buf = V.picks_buf;
  const int lane = qk_lane();
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (st->done || wid >= st->batch) return;
  const int r = st->pos - wid;
  int g = 0;
  if (r >= 0 && Kp[r] > st->mc) g = greedy_dispatch(adjP, W, r, st->t0, lane, picks_buf + (size_t)wid * L);
  if (lane == 0) gsz[wid] = g;
}

// Round 0 of the heuristic is a single start (the top-ranked vertex) whose greedy descent is one long
// dependent chain (one row AND per clique member).  When the rank-labelled bit matrix fits in LDS the
// chain runs out of LDS (~100 cycles per step instead of an L2 round trip).
__device__ __forceinline__ void d_clique_batch_lds(const SolverView& V) {
  const u64* __restrict__ adjP = V.adjP;
  const int* __restrict__ Kp = V.Kp;
  const int L = V.L, W = V.W;
  if (L <= 0) return;
  const SolverState* st = V.st;
  int* __restrict__ gsz = V.gsz;
  int* __restrict__ picks_buf = V.picks_buf;
  extern __shared__ __attribute__((aligned(16))) u64 cl_rows[];
  if (st->done) return;
  const int lane = qk_lane();
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int batch = st->batch;
  if (blockIdx.x * 4 >= batch) return;
  // any start of this workgroup worth descending from?  (uniform per block after the vote below)
  const int r = st->pos - wid;
  const bool want = (wid < batch) && (r >= 0) && (Kp[r] > st->mc);
  if (!__syncthreads_or(want ? 1 : 0)) {
    if (lane == 0 && wid < batch) gsz[wid] = 0;
    return;
  }
#pragma unroll 4
  for (int e = threadIdx.x; e < L * W; e += 256) cl_rows[e] = adjP[e];
  __syncthreads();
  int g = 0;
  if (want) {
    if (W <= 64) {
      int* pk = (int*)(cl_rows + (size_t)L * W) + (size_t)(threadIdx.x >> 6) * L;  // per-wave pick list after the matrix
      g = greedy_descent_lds1(cl_rows, W, r, st->t0, lane, pk, picks_buf + (size_t)wid * L);
    } else {
      g = greedy_dispatch(cl_rows, W, r, st->t0, lane, picks_buf + (size_t)wid * L);
    }
  }
  if (lane == 0 && wid < batch) gsz[wid] = g;
}
// then_scan (a launch of ONE workgroup only): the round's replay follows in the same launch
template <bool EXT>
__global__ __launch_bounds__(256) void k_clique_batch_lds(ViewExt<SolverView> x, SolverView one, int then_scan) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  d_clique_batch_lds(V);
  if (then_scan) {
    __syncthreads();
    if ((threadIdx.x >> 6) == 0) d_clique_scan(V, then_scan);
  }
}

// Sequential replay of pmc_heu::search_bounds over one batch (single wavefront).
// (one wavefront; a device function so that the launches on either side of a round can take it in: a dependent launch
// costs ~4.5 us on this chain whatever it does)
__device__ __forceinline__ void d_clique_scan(const SolverView& V, int next_batch) {
  const u64* __restrict__ adjP = V.adjP;
  const int* __restrict__ Kp = V.Kp;
  const int L = V.L, W = V.W;
  if (L <= 0) return;
  SolverState* __restrict__ st = V.st;
  const int* __restrict__ gsz = V.gsz;
  const int* __restrict__ picks_buf = V.picks_buf;
  int* __restrict__ best_picks = V.picks;
  if (st->done) return;
  const int lane = qk_lane();
  const int B = st->batch, pos = st->pos, ub = st->ub, cf = st->core_floor;
  int mc = st->mc, t = st->t0, best = st->best_r, done = 0, tainted = st->tainted;
  int cursor = 0;
  while (cursor < B) {
    const int wid = cursor + lane;
    const int r = pos - wid;
    const bool cand = (wid < B) && (r >= 0) && (Kp[r] > mc) && (gsz[wid] > mc);
    const u64 bal = __ballot(cand);
    if (!bal) {
      cursor += 64;
      continue;
    }
    const int first = __ffsll((long long)bal) - 1;
    const int wsel = cursor + first;
    const int rsel = pos - wsel;
    // |P| = #{u in N(v) : K[u] > mc} = popcount of row bits at ranks >= t
    int cnt = 0;
    for (int w = lane; w < W; w += 64) {
      u64 x = adjP[(size_t)rsel * W + w];
      const int lo = w * 64;
      if (lo + 63 < t)
        x = 0;
      else if (lo < t)
        x &= ~((1ULL << (t - lo)) - 1ULL);
      cnt += __popcll(x);
    }
    cnt = wave_sum_i32(cnt);
    if (cnt > mc) {
      mc = gsz[wsel];
      best = rsel;
      for (int i = lane; i < mc - 1; i += 64) best_picks[i] = picks_buf[(size_t)wsel * L + i];
      // t = first rank with Kp > mc (Kp is non-decreasing in rank)
      int lo = 0, hi = L;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (Kp[mid] > mc)
          hi = mid;
        else
          lo = mid + 1;
      }
      t = lo;
      if (mc >= ub) {
        done = 1;
        break;
      }
    } else if (cf > 0) {
      tainted = 1;  // (see k_rank_sort: under an injected bound this rule may not be what turns a start down)
    }
    cursor = wsel + 1;
  }
  const int newpos = pos - B;
  if (newpos < 0 || Kp[newpos] <= mc) done = 1;
  if (done && cf > 0 && (best < 0 || tainted)) {  // nothing this search can vouch for: again, with exact core numbers
    done = 0;
    mc = 0;
    best = -1;
    if (lane == 0) st->redo_cores = 1;
  }
  if (lane == 0) {
    st->tainted = tainted;
    st->mc = mc;
    st->best_r = best;
    st->t0 = t;
    st->pos = newpos;
    st->done = done;
    st->batch = next_batch;
    st->rounds += 1;
  }
}
template <bool EXT>
__global__ __launch_bounds__(64) void k_clique_scan(ViewExt<SolverView> x, SolverView one, int next_batch) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  d_clique_scan(V, next_batch);
}

// The rest of the heuristic in ONE round (round 5).  On a graph whose bulk has core numbers ABOVE the largest clique
// (use_crosscheck = 0 at L ~ 20 k: mean degree 775, bulk cores ~400, clique 345) the test K > mc never cuts anything
// and pmc_heu tries every vertex as a start: 20 rounds of CLIQUE_BATCH starts, each as long as its longest descent
// (a chain of ~345 dependent row reads, ~230 us) plus a host check — 5 ms.  The replay is exact whatever bound the
// descents were started with (k_clique_batch_lds's comment: a descent that strays below the current bound cannot return
// more than that bound), so ALL remaining starts can be speculated at once with the bound of this moment: k_clique_sweep
// runs them grid-strided (sizes only — a pick list per start would be L x L ints), d_clique_scan_all replays the
// sequential acceptance over all of them and re-runs the descent of the few starts it accepts to get their members
// (same start, same bound: the same picks).
template <bool EXT>
__global__ __launch_bounds__(256) void k_clique_sweep(ViewExt<SolverView> x, SolverView one) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const u64* __restrict__ adjP = V.adjP;
  const int* __restrict__ Kp = V.Kp;
  const int L = V.L, W = V.W;
  if (L <= 0) return;
  const SolverState* __restrict__ st = V.st;
  if (st->done) return;
  int* __restrict__ gsz = V.gsz;
  const int lane = qk_lane();
  const int pos = st->pos, mc0 = st->mc, t0 = st->t0;
  const int nwaves = gridDim.x * 4;
  for (int wid = blockIdx.x * 4 + (threadIdx.x >> 6); wid <= pos; wid += nwaves) {
    const int r = pos - wid;
    int g = 0;
    if (Kp[r] > mc0) g = greedy_dispatch(adjP, W, r, t0, lane, nullptr);
    if (lane == 0) gsz[wid] = g;
  }
}
#define CSA_THREADS 1024
#define CSA_LIST 256
template <bool EXT>
__global__ __launch_bounds__(CSA_THREADS) void k_clique_scan_all(ViewExt<SolverView> x, SolverView one) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const u64* __restrict__ adjP = V.adjP;
  const int* __restrict__ Kp = V.Kp;
  const int L = V.L, W = V.W;
  if (L <= 0) return;
  SolverState* __restrict__ st = V.st;
  const int* __restrict__ gsz = V.gsz;
  int* __restrict__ best_picks = V.picks;
  if (st->done) return;
  const int lane = qk_lane(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pos = st->pos, B = pos + 1, ub = st->ub, cf = st->core_floor, t_sweep = st->t0, mc0 = st->mc;
  // A start can only be accepted if its descent beat the bound the sweep started from: sixteen waves list those (in
  // order, a contiguous range of starts each) and one wave replays the few that are left — walking all the starts 64 at
  // a time was 320 dependent round trips, 170 us at 20 k starts.  A range with more than CSA_LIST of them is walked whole.
  __shared__ int s_list[CSA_THREADS / 64][CSA_LIST];
  __shared__ int s_cnt[CSA_THREADS / 64];
  const int chunk = (((B + CSA_THREADS / 64 - 1) / (CSA_THREADS / 64)) + 63) & ~63;
  {
    const int lo = wave * chunk, hi = min(B, lo + chunk);
    int cnt = 0;
    for (int base = lo; base < hi; base += 64) {
      const int wid = base + lane;
      const bool c = wid < hi && Kp[pos - wid] > mc0 && gsz[wid] > mc0;
      const u64 bal = __ballot(c);
      if (c) {
        const int at = cnt + __popcll(bal & lanemask_lt());
        if (at < CSA_LIST) s_list[wave][at] = wid;
      }
      cnt += __popcll(bal);
    }
    if (lane == 0) s_cnt[wave] = cnt;
  }
  __syncthreads();
  if (wave != 0) return;
  int mc = mc0, t = st->t0, best = st->best_r, tainted = st->tainted;
  bool full = false, accepted = false;
  for (int seg = 0; seg < CSA_THREADS / 64 && !full; ++seg) {
    const int n_listed = s_cnt[seg];
    const bool listed = n_listed <= CSA_LIST;
    const int seg_lo = seg * chunk, seg_n = listed ? n_listed : max(0, min(B, seg_lo + chunk) - seg_lo);
    int cursor = 0;
    while (cursor < seg_n) {
      const int idx = cursor + lane;
      const int wid = idx < seg_n ? (listed ? s_list[seg][idx] : seg_lo + idx) : 0;
      const int r = pos - wid;
      const bool cand = (idx < seg_n) && (Kp[r] > mc) && (gsz[wid] > mc);
      const u64 bal = __ballot(cand);
      if (!bal) {
        cursor += 64;
        continue;
      }
      const int first = __ffsll((long long)bal) - 1;
      const int wsel = __builtin_amdgcn_readlane(wid, first);
      const int rsel = pos - wsel;
      // |P| = #{u in N(v) : K[u] > mc} = popcount of row bits at ranks >= t
      int cnt = 0;
      for (int w = lane; w < W; w += 64) {
        u64 xw = adjP[(size_t)rsel * W + w];
        const int lo = w * 64;
        if (lo + 63 < t)
          xw = 0;
        else if (lo < t)
          xw &= ~((1ULL << (t - lo)) - 1ULL);
        cnt += __popcll(xw);
      }
      cnt = wave_sum_i32(cnt);
      if (cnt > mc) {
        mc = gsz[wsel];
        best = rsel;
        accepted = true;  // (its members: after the replay, see below)
        int lo = 0, hi = L;  // t = first rank with Kp > mc (Kp is non-decreasing in rank)
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (Kp[mid] > mc)
            hi = mid;
          else
            lo = mid + 1;
        }
        t = lo;
        if (mc >= ub) {
          full = true;
          break;
        }
      } else if (cf > 0) {
        tainted = 1;  // (see k_rank_sort: under an injected bound this rule may not be what turns a start down)
      }
      cursor += first + 1;
    }
  }
  // The members of the LAST accepted start only: the sweep's descent once more (same start, same bound t_sweep: the same
  // picks).  Until round 6 every accepted start re-ran its descent on the spot — a chain of |clique| dependent row ANDs by
  // this one wave: four or five improvements on the way to a clique of 2639 members were 3.4 ms of this kernel
  // (bench.py connected_leg.l5k); the replay itself needs sizes only.
  if (accepted) (void)greedy_dispatch(adjP, W, best, t_sweep, lane, best_picks);
  int done = 1;  // every start has been replayed
  if (cf > 0 && (best < 0 || tainted)) {  // nothing this search can vouch for: again, with exact core numbers
    done = 0;
    mc = 0;
    best = -1;
    if (lane == 0) st->redo_cores = 1;
  }
  if (lane == 0) {
    st->tainted = tainted;
    st->mc = mc;
    st->best_r = best;
    st->t0 = t;
    st->pos = -1;
    st->done = done;
    st->batch = CLIQUE_BATCH;
    st->rounds += 1;
  }
}

// KCORE_HEU shortcut (reference src/graph.cc:67-82, including its shifted indexing): decided on device.
// Writes the member bitset directly; st->mc = clique size, st->best_r = -2 marks "bitset already built".
template <bool EXT>
__global__ __launch_bounds__(256) void k_kcore_heu(ViewExt<SolverView> x, SolverView one, double thr) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  const int* __restrict__ core = V.core;
  const int L = V.L, W = V.W;
  if (L <= 0) return;
  SolverState* st = V.st;
  u64* __restrict__ member_bits = V.member_bits;
  __shared__ int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int max_core = st->max_core;
  const bool take = (thr != 1.0) && (max_core > (int)(thr * (double)L));
  if (!take) return;  // falls through to the PMC heuristic
  for (int w = threadIdx.x; w < W; w += 256) member_bits[w] = 0;
  __syncthreads();
  for (int i = 1 + threadIdx.x; i <= L; i += 256) {
    const int kc = (i < L) ? core[i] + 1 : core[L - 1];  // PMC's k_cores[] has V+1 entries; last one is unshifted
    if (kc >= max_core) {
      atomicOr(&member_bits[(i - 1) >> 6], 1ULL << ((i - 1) & 63));
      atomicAdd(&s_cnt, 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    st->mc = s_cnt;
    st->best_r = -2;
    st->done = 1;
  }
}

// =================================================================================================
// K13-K16: chain TIMs, GNC-TLS yaw, rotation-inlier chain rule, COTE, final inliers.  One workgroup
// of 256 threads; wavefront 0 runs the GNC loop with the fixed-shape sum64 reductions.
#define FIN_LDS_BYTES (152 * 1024)
#define FIN_THREADS 768  // three groups of four wavefronts: one COTE axis each

// GNC-TLS yaw estimation on ONE wavefront (reference solveForRotation2D, include/quatro.hpp:430-572; closed-form
// 2x2 rotation instead of JacobiSVD, fixed 64-lane summation order — oracle divergence D6).  X = source, Y =
// destination chain TIMs (xy rows), Wt = weights (in: 1, out: final weights).  __forceinline__: at the LDS call
// site the pointers derive from the dynamic shared array and every access becomes a ds_* instruction.
__device__ __forceinline__ void gnc_wave(int lane, const double* X0, const double* X1, const double* Y0, const double* Y1,
                                         double* Wt, int M, double rot_nb, double gnc_factor, int max_it, double cost_thr,
                                         double (&Rout)[4], double* cost_out, int* iters_out) {
  double nb_sq = rot_nb * rot_nb;
  if (nb_sq < 1e-16) nb_sq = 1e-2;
  double mu = 1.0, prev_cost = INFINITY, cost = INFINITY;
  double R0 = 1, R1 = 0, R2 = 0, R3 = 1;
  int iters = 0;
  for (int it = 0; it < max_it; ++it) {
    iters = it + 1;
    double h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    for (int j = lane; j < M; j += 64) {
      const double wx0 = Wt[j] * X0[j], wx1 = Wt[j] * X1[j];
      h0 = h0 + wx0 * Y0[j];
      h1 = h1 + wx0 * Y1[j];
      h2 = h2 + wx1 * Y0[j];
      h3 = h3 + wx1 * Y1[j];
    }
    h0 = wave_sum64_f64_brev(h0);
    h1 = wave_sum64_f64_brev(h1);
    h2 = wave_sum64_f64_brev(h2);
    h3 = wave_sum64_f64_brev(h3);
    {
      const double a = h0 + h3, b = h1 - h2;
      const double nrm = sqrt(a * a + b * b);
      double c = 1.0, s = 0.0;
      if (nrm > 0.0) {
        c = a / nrm;
        s = b / nrm;
      }
      R0 = c;
      R1 = -s;
      R2 = s;
      R3 = c;
    }
    if (it == 0) {  // the largest residual only seeds mu (reference :494-500); later iterations do not need it
      double max_r = -INFINITY;
      for (int j = lane; j < M; j += 64) {
        const double e0 = Y0[j] - (R0 * X0[j] + R1 * X1[j]), e1 = Y1[j] - (R2 * X0[j] + R3 * X1[j]);
        const double r2 = e0 * e0 + e1 * e1;
        max_r = fmax(max_r, r2);
      }
      max_r = wave_max_f64(max_r);
      mu = 1 / (2 * max_r / nb_sq - 1);
      if (mu <= 0) break;
    }
    const double th1 = (mu + 1) / mu * nb_sq, th2 = mu / (mu + 1) * nb_sq;
    double cpart = 0;
    for (int j = lane; j < M; j += 64) {
      const double e0 = Y0[j] - (R0 * X0[j] + R1 * X1[j]), e1 = Y1[j] - (R2 * X0[j] + R3 * X1[j]);
      const double r2 = e0 * e0 + e1 * e1;
      cpart = cpart + Wt[j] * r2;
      double w;
      if (r2 >= th1)
        w = 0;
      else if (r2 <= th2)
        w = 1;
      else
        w = sqrt(nb_sq * mu * (mu + 1) / r2) - mu;
      Wt[j] = w;
    }
    cost = wave_sum64_f64_brev(cpart);
    const double cost_diff = fabs(cost - prev_cost);
    mu = mu * gnc_factor;
    prev_cost = cost;
    if (cost_diff < cost_thr) break;
  }
  Rout[0] = R0;
  Rout[1] = R1;
  Rout[2] = R2;
  Rout[3] = R3;
  *cost_out = cost;
  *iters_out = iters;
}

// The 3-DoF counterpart ("next" row (f)4, reg_name "TEASER"): the same GNC-TLS loop over 3-D TIMs with the weighted
// 3x3 rotation of qtr_math.h (teaser::utils::svdRot restated in Horn's quaternion form, reference
// include/teaser/utils.h:123-149).  X = source, Y = destination rows; Wt = weights (in: 1).
__device__ __forceinline__ void gnc3_wave(int lane, const double* X0, const double* X1, const double* X2, const double* Y0,
                                          const double* Y1, const double* Y2, double* Wt, int M, double rot_nb,
                                          double gnc_factor, int max_it, double cost_thr, double (&Rout)[9],
                                          double* cost_out, int* iters_out) {
  double nb_sq = rot_nb * rot_nb;
  if (nb_sq < 1e-16) nb_sq = 1e-2;
  double mu = 1.0, prev_cost = INFINITY, cost = INFINITY;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  int iters = 0;
  for (int it = 0; it < max_it; ++it) {
    iters = it + 1;
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = lane; j < M; j += 64) {
      const double w = Wt[j];
      const double wx0 = w * X0[j], wx1 = w * X1[j], wx2 = w * X2[j];
      const double y0 = Y0[j], y1 = Y1[j], y2 = Y2[j];
      H[0] = H[0] + wx0 * y0;
      H[1] = H[1] + wx0 * y1;
      H[2] = H[2] + wx0 * y2;
      H[3] = H[3] + wx1 * y0;
      H[4] = H[4] + wx1 * y1;
      H[5] = H[5] + wx1 * y2;
      H[6] = H[6] + wx2 * y0;
      H[7] = H[7] + wx2 * y1;
      H[8] = H[8] + wx2 * y2;
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) H[a] = wave_sum64_f64_brev(H[a]);
    qm_rot3_from_h(H, R);
    if (it == 0) {
      double max_r = -INFINITY;
      for (int j = lane; j < M; j += 64) {
        const double x0 = X0[j], x1 = X1[j], x2 = X2[j];
        const double e0 = Y0[j] - ((R[0] * x0 + R[1] * x1) + R[2] * x2);
        const double e1 = Y1[j] - ((R[3] * x0 + R[4] * x1) + R[5] * x2);
        const double e2 = Y2[j] - ((R[6] * x0 + R[7] * x1) + R[8] * x2);
        max_r = fmax(max_r, (e0 * e0 + e1 * e1) + e2 * e2);
      }
      max_r = wave_max_f64(max_r);
      mu = 1 / (2 * max_r / nb_sq - 1);
      if (mu <= 0) break;
    }
    const double th1 = (mu + 1) / mu * nb_sq, th2 = mu / (mu + 1) * nb_sq;
    double cpart = 0;
    for (int j = lane; j < M; j += 64) {
      const double x0 = X0[j], x1 = X1[j], x2 = X2[j];
      const double e0 = Y0[j] - ((R[0] * x0 + R[1] * x1) + R[2] * x2);
      const double e1 = Y1[j] - ((R[3] * x0 + R[4] * x1) + R[5] * x2);
      const double e2 = Y2[j] - ((R[6] * x0 + R[7] * x1) + R[8] * x2);
      const double r2 = (e0 * e0 + e1 * e1) + e2 * e2;
      cpart = cpart + Wt[j] * r2;
      double w;
      if (r2 >= th1)
        w = 0;
      else if (r2 <= th2)
        w = 1;
      else
        w = sqrt(nb_sq * mu * (mu + 1) / r2) - mu;
      Wt[j] = w;
    }
    cost = wave_sum64_f64_brev(cpart);
    const double cost_diff = fabs(cost - prev_cost);
    mu = mu * gnc_factor;
    prev_cost = cost;
    if (cost_diff < cost_thr) break;
  }
#pragma unroll
  for (int a = 0; a < 9; ++a) Rout[a] = R[a];
  *cost_out = cost;
  *iters_out = iters;
}

// One COTE axis on one GROUP of four wavefronts (256 threads; reference estimate(), include/quatro.hpp:618-747).
// Every thread of the workgroup calls it with identical N so that the barriers match; `act` is false for
// threads that only keep the barriers company.  Steps:
//   1. the 2N interval endpoints, key = X -/+ range, insertion position = 2p / 2p+1
//   2. bitonic sort by (key, position) across the 256 threads (O(n log^2 n) compare-exchanges: a rank-counting
//      sort was VALU-bound, all three axes share ONE compute unit)
//   3. the signed per-event terms of the six running sums, in parallel
//   4. six lanes of one wave, one running sum each, add the terms in the reference's sequential order
//      (binary64 addition is not associative, so the order is part of the result); loads run one batch of
//      eight ahead of the dependent additions
//   5. x_hat / cost per event and the arg-min with Eigen's minCoeff tie rule (first strict minimum)
//   6. median of the consensus window: its values are two sorted sequences, merged by binary searches (see there)
// __forceinline__ on purpose: at the LDS call site the pointers derive directly from the dynamic shared
// array, so address-space inference turns every access into a ds_* instruction (generic pointers ran ~10x
// slower through the flat path).
struct CoteOut {
  double est;
  int ncard;
};
__device__ __forceinline__ CoteOut cote_axis4(bool act, int tl, const double* __restrict__ X, int N, int nc,
                                              double range, const double* __restrict__ R /* per-element ranges, or null */,
                                              int median_sel, double* sxv, int* spos,
                                              double* T /* 6 arrays of nc; first holds the sort keys/positions */,
                                              double* s_bcast /* [4] per axis */, double* s_redc /* [4] */,
                                              int* s_redi /* [4] */, int* dbg,
                                              int sw /* which of the group's four waves runs the serial parts */,
                                              const double* range_sum = nullptr /* the N uniform ranges, added up */) {
  const int lane = tl & 63, gw = tl >> 6;
  long long tc0 = clock64(), tc1;
#define COTE_TICK(slot)                                  \
  if (dbg && tl == 0) {                                  \
    tc1 = clock64();                                     \
    dbg[slot] = (int)((tc1 - tc0) >> 4);                 \
    tc0 = tc1;                                           \
  }
  // ---- 1./2. sort of (key, position), ascending; keys/positions live in the (still unused) T area (two buffers).
  // A merge sort by RANKS: sorted blocks of B endpoints are merged pairwise by letting every endpoint find, by binary
  // search, how many endpoints of the partner block precede it — its place in the merged block is that plus its place in
  // its own — log2(n) rounds of one barrier each, ~45 dependent LDS reads per endpoint in all.  (Rounds 2 - 3: a bitonic
  // network, 45 barrier-separated stages at 512 endpoints, 15.9 us; up to 256 endpoints every endpoint counted the
  // smaller ones, 11 us at 248.)  (key, position) is a strict total order, so there are no ties to break between blocks.
  int n2 = 1;
  while (n2 < nc) n2 <<= 1;
  {
    double* k0 = T;
    int* p0 = (int*)(T + n2);
    double* k1 = T + n2 + (n2 >> 1) + 1;
    int* p1 = (int*)(k1 + n2);
    if (act && nc > 0)  // (no endpoints: not even the padding is written — the scratch may be empty)
      for (int i = tl; i < n2; i += 256) {
        const double ri = (R && i < nc) ? R[i >> 1] : range;
        const double k = (i < nc) ? ((i & 1) ? X[i >> 1] + ri : X[i >> 1] - ri) : INFINITY;
        k0[i] = (k != k) ? INFINITY : k;  // NaN sorts as +inf (ties by position); the padding follows every endpoint
        p0[i] = i;
      }
    __syncthreads();
    for (int B = 1; B < n2; B <<= 1) {
      // (branch-free lower bound over the partner block — a power of two — and two endpoints of a thread side by side: the
      // reads of a step are independent, and a step is what the round's time is made of)
      auto place = [&](int i, int lo, double k, int pp) __attribute__((always_inline)) {
        const int pair_base = i & ~(2 * B - 1), own = i & (B - 1);
        k1[pair_base + own + lo] = k;
        p1[pair_base + own + lo] = pp;
      };
      if (act)
        for (int i = tl; i < n2; i += 512) {
          const int ia = i, ib = i + 256;
          const bool two = ib < n2;
          const double ka = k0[ia], kb = two ? k0[ib] : 0.0;
          const int pa = p0[ia], pb = two ? p0[ib] : 0;
          const int parta = (ia & ~(2 * B - 1)) + ((ia & B) ? 0 : B), partb = two ? (ib & ~(2 * B - 1)) + ((ib & B) ? 0 : B) : parta;
          int la = 0, lb = 0;
          for (int h = B >> 1; h >= 1; h >>= 1) {
            const double kma = k0[parta + la + h - 1], kmb = k0[partb + lb + h - 1];
            const int pma = p0[parta + la + h - 1], pmb = p0[partb + lb + h - 1];
            la += ((kma < ka) | ((kma == ka) & (pma < pa))) ? h : 0;
            lb += ((kmb < kb) | ((kmb == kb) & (pmb < pb))) ? h : 0;
          }
          {
            const double kma = k0[parta + la], kmb = k0[partb + lb];
            const int pma = p0[parta + la], pmb = p0[partb + lb];
            la += ((kma < ka) | ((kma == ka) & (pma < pa))) ? 1 : 0;
            lb += ((kmb < kb) | ((kmb == kb) & (pmb < pb))) ? 1 : 0;
          }
          place(ia, la, ka, pa);
          if (two) place(ib, lb, kb, pb);
        }
      __syncthreads();
      double* tk = k0;
      k0 = k1;
      k1 = tk;
      int* tp = p0;
      p0 = p1;
      p1 = tp;
    }
    if (act)
      for (int i = tl; i < nc; i += 256) {
        const int p = p0[i];
        spos[i] = p;
        sxv[i] = X[p >> 1];
      }
  }
  COTE_TICK(0)
  if (act && tl == 64 * ((sw + 1) & 3)) {  // sum of N ranges in the reference's order (:660), off the serial wave
    double r = 0;
    if (range_sum) r = *range_sum;  // (uniform range: the caller had it added up earlier)
    else
      for (int i = 0; i < N; ++i) r += R ? R[i] : range;
    s_bcast[2] = r;
  }
  __syncthreads();
  // ---- 3. per-event terms (sort keys are dead: T takes their place)
  if (act) {
    const double weight_u = 1.0 / (range * range);
    for (int i = tl; i < nc; i += 256) {
      const int eps = (spos[i] & 1) ? -1 : 1;
      const double xv = sxv[i];
      const double rv = R ? R[spos[i] >> 1] : range;
      const double weight = R ? 1.0 / (rv * rv) : weight_u;  // weights = ranges.square().inverse()
      T[i] = eps * weight;
      T[(size_t)nc + i] = eps * weight * xv;
      T[2 * (size_t)nc + i] = -(eps * rv);
      T[3 * (size_t)nc + i] = eps * xv;
      T[4 * (size_t)nc + i] = eps * xv * xv;
      T[5 * (size_t)nc + i] = (double)eps;
    }
  }
  __syncthreads();
  COTE_TICK(1)
  // ---- 4. running sums, in place: lane c of wave 0 owns array c
  // (wave sw of the group: the three axes' groups are waves 0-3, 4-7, 8-11 of one workgroup and a wave's SIMD is its
  // index mod 4 — with wave 0 of every group the three serial chains shared one SIMD's issue slots and ran at half speed:
  // clocks / 16 of this step at 500 endpoints, 1660 with three axes against 835 with one)
  if (act && gw == sw && lane < 6) {
    double* t = T + (size_t)lane * nc;
    double acc = (lane == 2) ? s_bcast[2] : 0.0;
    int i = 0;
    // sixteen terms per round: the loads go out back to back, the additions stay a dependent chain in the reference's
    // order, the stores follow (one wavefront issues an instruction every ~5 clocks, so instructions per event matter)
    // (the arrays start 16-byte aligned — nc is even and the carving rounds to 16 — so two terms travel per LDS access:
    // the wave issues one instruction every ~8 clocks, and it is the instruction count that bounds this chain)
    typedef double cote_d2 __attribute__((ext_vector_type(2)));
    if ((((size_t)t) & 15) == 0) {
      for (; i + 16 <= nc; i += 16) {
        cote_d2 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *(const cote_d2*)(t + i + 2 * q);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          acc += v[q].x;
          v[q].x = acc;
          acc += v[q].y;
          v[q].y = acc;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) *(cote_d2*)(t + i + 2 * q) = v[q];
      }
    }
    for (; i + 16 <= nc; i += 16) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = t[i + q];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc += v[q];
        v[q] = acc;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) t[i + q] = v[q];
    }
    for (; i < nc; ++i) {
      acc += t[i];
      t[i] = acc;
    }
  }
  __syncthreads();
  COTE_TICK(2)
  // ---- 5. per-event estimate and cost, arg-min (first strict minimum; NaN is never selected unless first)
  int mi = 0, ncard = 0;
  double est = 0;
  {
    double bc = INFINITY;
    int bi = 0x7fffffff;
    if (act)
      for (int i = tl; i < nc; i += 256) {
        const double xh = T[(size_t)nc + i] / T[i];
        const double residual = T[5 * (size_t)nc + i] * xh * xh + T[4 * (size_t)nc + i] - 2 * T[3 * (size_t)nc + i] * xh;
        const double c = residual + T[2 * (size_t)nc + i];
        if (i == 0) s_bcast[3] = c;
        if (c < bc || (c == bc && i < bi)) {
          bc = c;
          bi = i;
        }
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double oc = __shfl_xor(bc, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (oc < bc || (oc == bc && oi < bi)) {
        bc = oc;
        bi = oi;
      }
    }
    if (act && lane == 0) {
      s_redc[gw] = bc;
      s_redi[gw] = bi;
    }
    if (act && tl == 0) {
      s_bcast[0] = 0;
      s_bcast[1] = 0;
    }
    __syncthreads();
    if (act) {
      bc = s_redc[0];
      bi = s_redi[0];
#pragma unroll
      for (int g = 1; g < 4; ++g) {
        const double oc = s_redc[g];
        const int oi = s_redi[g];
        if (oc < bc || (oc == bc && oi < bi)) {
          bc = oc;
          bi = oi;
        }
      }
      mi = bi;
      const double c0 = s_bcast[3];
      if (mi == 0x7fffffff || c0 != c0) mi = 0;
      ncard = (int)T[5 * (size_t)nc + mi];
      est = T[(size_t)nc + mi] / T[mi];
    }
  }
  __syncthreads();
  COTE_TICK(3)
  // ---- 6. the two middle order statistics of {X of events mi, mi-1, ..., mi-ncard+1}
  // The window is a run of the SORTED events, and inside it the opening endpoints (key X - range) stand in ascending X, the
  // closing ones (key X + range) too: the window's values are two sorted sequences.  Where an event of either kind stands
  // among its own needs no scan — the running count of step 4 gives #openings up to event e as (e + 1 + count(e)) / 2 — so
  // the two sequences are written out side by side, every value finds its place in the merged order by four binary
  // searches, and the values at ranks ncard / 2 - 1 and ncard / 2 report themselves (equal values: the same number, whoever
  // writes it).  250 members: 1.6 us instead of the 10.8 of counting every value against every other.  Rounding can put two
  // DIFFERENT X on one key (then their order is by position, not by X) and NaN compares with nothing: a sequence that is
  // not ascending sends the axis back to the count.
  const int ra = ncard / 2 - 1, rb = ncard / 2;
  const bool med = act && median_sel && ncard >= 2;
  double* XL = T;                    // openings of the window, ascending
  double* XU = T + (size_t)nc;       // closings
  int NL = 0, NU = 0;
  if (act && tl == 0) s_redi[0] = 0;  // (dead since the barrier above) 1: not ascending
  if (med) {
    const int e0 = mi - ncard + 1;
    const double* cnt = T + 5 * (size_t)nc;
    const int lbase = e0 > 0 ? (e0 + (int)cnt[e0 - 1]) / 2 : 0;  // openings before the window
    NL = (mi + 1 + (int)cnt[mi]) / 2 - lbase;
    NU = ncard - NL;
    for (int e = e0 + tl; e <= mi; e += 256) {
      const int lo_e = (e + 1 + (int)cnt[e]) / 2;  // openings up to and including e
      if (spos[e] & 1) XU[(e + 1 - lo_e) - (e0 - lbase) - 1] = sxv[e];
      else XL[lo_e - lbase - 1] = sxv[e];
    }
  }
  __syncthreads();
  if (med) {
    auto bounds = [&](const double* a, int n, double v, int& lb, int& ub) __attribute__((always_inline)) {
      int lo = 0, hi = n;  // first index with a[i] >= v
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
      }
      lb = lo;
      hi = n;  // first index with a[i] > v
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1;
        else hi = mid;
      }
      ub = lo;
    };
    bool bad = false;
    for (int j = tl; j < ncard; j += 256) {
      const bool isl = j < NL;
      const double* own = isl ? XL : XU;
      const int k = isl ? j : j - NL, n = isl ? NL : NU;
      const double v = own[k];
      bad = bad || (k + 1 < n && !(v <= own[k + 1]));
      int l1, u1, l2, u2;
      bounds(XL, NL, v, l1, u1);
      bounds(XU, NU, v, l2, u2);
      if (l1 + l2 <= ra && ra < u1 + u2) s_bcast[0] = v;
      if (l1 + l2 <= rb && rb < u1 + u2) s_bcast[1] = v;
    }
    if (bad) s_redi[0] = 1;
  }
  __syncthreads();
  if (med && s_redi[0]) {
    for (int j = tl; j < ncard; j += 256) {  // (spreading one value's comparisons over a wavefront was measured 4x
      const double vj = sxv[mi - j];          //  slower: the cross-lane reduction costs more than the serial walk)
      int rk = 0;
      for (int q = 0; q < ncard; ++q) {
        const double vq = sxv[mi - q];
        rk += ((vq < vj) | ((vq == vj) & (q < j))) ? 1 : 0;
      }
      if (rk == ra) s_bcast[0] = vj;
      if (rk == rb) s_bcast[1] = vj;
    }
  }
  __syncthreads();
  if (act && median_sel) {
    if (ncard >= 2)
      est = (s_bcast[0] + s_bcast[1]) / 2.0;
    else if (ncard == 1)
      est = sxv[mi];
  }
  COTE_TICK(4)
#undef COTE_TICK
  CoteOut o;
  o.est = est;
  o.ncard = ncard;
  return o;
}

// Clique members of the finished search -> bitset in ORIGINAL labels -> ascending id list in `clique`;
// *s_M (LDS) receives the member count.  Called by every thread of a 256-thread workgroup.
__device__ __forceinline__ void clique_members(const SolverState* st, u64* member_bits, const int* picks,
                                               const int* perm, int* clique, int W, int* s_M) {
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int mc = st->mc;
  if (st->best_r != -2) {
    for (int w = tid; w < W; w += nthr) member_bits[w] = 0;
    __syncthreads();
    {
      // the winning start's picks were saved by k_clique_scan: depth-1 picks + the start vertex itself
      const int depth = (st->best_r >= 0) ? mc : 0;
      for (int i = tid; i < depth; i += nthr) {
        const int rr = (i == depth - 1) ? st->best_r : picks[i];
        const int v = perm[rr];
        atomicOr(&member_bits[v >> 6], 1ULL << (v & 63));
      }
    }
    __syncthreads();
  }
  if (tid == 0) *s_M = 0;
  __syncthreads();
  if (wave == 0) {
    int base = 0;
    for (int w0 = 0; w0 < W; w0 += 64) {
      const int w = w0 + lane;
      u64 x = (w < W) ? member_bits[w] : 0ULL;
      int tot;
      int off = wave_excl_scan_i32(__popcll(x), &tot);
      int o = base + off;
      while (x) {
        const int b = __ffsll((long long)x) - 1;
        x &= x - 1;
        clique[o++] = w * 64 + b;
      }
      base += tot;
    }
    if (lane == 0) *s_M = base;
  }
  __syncthreads();
}

// Result record and solver state straight into the slot's pinned host mailbox: the host needs one stream
// synchronisation and no copy launches.  Called by every thread of the workgroup at kernel exit.
__device__ __forceinline__ void export_result(int* __restrict__ mail, const qtr_result* res, const SolverState* st,
                                              int seq) {
  if (!mail) return;
  __syncthreads();
  constexpr int NR = (int)(sizeof(qtr_result) / 4), NS = (int)(sizeof(SolverState) / 4);
  const int t = threadIdx.x;
  // wavefront 0 carries the record, wavefront 1 the state; each folds its own tag (common.h: the host recomputes both
  // before trusting the payload) — word 63 for the record, word 96 for the state
  static_assert(NR <= 63 && NS <= 32, "tag words 63 and 96 of the solver area");
  if (t < 64) {
    const int v = (t < NR) ? ((const int*)res)[t] : 0;
    int x = v;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if (t < NR) mail[MAIL_SOLVER + t] = v;
    if (t == 63) mail[MAIL_SOLVER + 63] = seq ^ x ^ MAIL_TAG_SALT;
  } else if (t < 128) {
    const int i = t - 64;
    const int v = (i < NS) ? ((const int*)st)[i] : 0;
    int x = v;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if (i < NS) mail[MAIL_SOLVER + 64 + i] = v;
    if (i == 32) mail[MAIL_SOLVER + 96] = seq ^ x ^ MAIL_TAG_SALT;
  }
  __threadfence_system();
  __syncthreads();
  if (t == 0) {
    __threadfence_system();
    mail[MAIL_SEQ_SOLVE] = seq;
  }
}

// Stand-alone clique extraction for qtr_max_clique (teaser::MaxCliqueSolver::findMaxClique boundary,
// reference src/graph.cc:15-98): res->n_clique / max_core / n_edges are filled, nothing else is estimated.
__global__ __launch_bounds__(256) void k_clique_only(SolverState* st, u64* member_bits, const int* picks,
                                                     const int* perm, int* clique, int W, qtr_result* res,
                                                     int* mail, int seq) {
  __shared__ int s_M;
  clique_members(st, member_bits, picks, perm, clique, W, &s_M);
  if (threadIdx.x == 0) {
    res->n_clique = s_M;
    res->max_core = st->max_core;
    res->n_edges = st->n_edges2 / 2;
    res->status = QTR_OK;
  }
  export_result(mail, res, st, seq);
}

// qtr_max_clique input hygiene + degrees, one wavefront per row: the diagonal bit and the bits past L are
// cleared in our copy of the matrix (a teaser::Graph has no self loops, include/teaser/graph.h:96-105), then
// deg[i] = popcount(row i).  Symmetry is the caller's contract.
__global__ __launch_bounds__(256) void k_row_degrees(u64* __restrict__ bm, int L, int W, int* __restrict__ deg) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= L) return;
  int c = 0;
  for (int w = lane; w < W; w += 64) {
    u64 x = bm[(size_t)row * W + w], y = x;
    if (w == (row >> 6)) y &= ~(1ULL << (row & 63));
    if (w == W - 1 && (L & 63)) y &= (1ULL << (L & 63)) - 1;
    if (y != x) bm[(size_t)row * W + w] = y;
    c += __popcll(y);
  }
  c = wave_sum_i32(c);
  if (lane == 0) deg[row] = c;
}

template <bool EXT>
__global__ __launch_bounds__(FIN_THREADS) void k_finalize(ViewExt<SolverView> x, SolverView one, qtr_params prm,
                                                          int scan_batch) {
  const SolverView& A = EXT ? x.ext[blockIdx.z] : one;  // (inline on purpose: see ViewExt)
  extern __shared__ __attribute__((aligned(16))) double fin_lds[];
  __shared__ int s_M, s_N, s_nrot, s_nfinal, s_minidx, s_ncard, s_iters;
  __shared__ double s_R[9], s_cost, s_est, s_bestcost;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int L = A.L, W = A.W;
  SolverState* st = A.st;
  qtr_result* res = A.res;
  // scan_batch: the replay of the clique round launched just before (k_clique_scan's job, without its launch)
  if (scan_batch) {
    if (wave == 0) d_clique_scan(A, scan_batch);
    __syncthreads();
  }
  const int mc = st->mc;
  const long long t_fin0 = clock64();
  if (L > 0 && !st->done) {
    // the clique search is not through (solver_continue will run its remaining starts and launch this kernel again):
    // nothing to estimate from yet — the state goes to the host, which only looks at `done` / `redo_cores` (on the
    // 20 k-correspondence graphs of use_crosscheck = 0 this estimate from a provisional clique took 110 us)
    if (tid == 0) {
      res->n_clique = 0;
      res->valid = 0;
      res->status = QTR_OK;
    }
    __syncthreads();
    export_result(A.mail, res, st, A.seq);
    return;
  }

  clique_members(st, A.member_bits, A.picks, A.perm, A.clique, W, &s_M);
  const int M = s_M;
  if (tid == 0) {
    res->n_clique = M;
    res->max_core = st->max_core;
    res->n_edges = st->n_edges2 / 2;
    res->valid = 0;
    res->cost = INFINITY;
    res->gnc_iters = 0;
    res->n_rot_inliers = 0;
    res->n_final = 0;
    for (int i = 0; i < 16; ++i) res->T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    res->n_card[0] = res->n_card[1] = res->n_card[2] = 0;
    res->status = QTR_OK;
  }
  __syncthreads();
  (void)mc;
  if (M <= 1) {  // reference :809-813
    if (tid == 0) res->status = QTR_ERR_CLIQUE_TOO_SMALL;
    export_result(A.mail, res, st, A.seq);
    return;
  }

  // ---- scratch layout: LDS when it fits (5 arrays of M doubles; 7 in the 3-DoF mode), else global
  double* X0;
  double* X1;
  double* Y0;
  double* Y1;
  double* Wt;
  double* X2 = nullptr;
  double* Y2 = nullptr;
  const bool teaser = prm.reg_mode == QTR_REG_TEASER;
  const bool use_lds = ((size_t)M * (teaser ? 7 : 5) * sizeof(double) <= (size_t)FIN_LDS_BYTES);
  if (use_lds) {
    X0 = fin_lds;
    X1 = X0 + M;
    Y0 = X1 + M;
    Y1 = Y0 + M;
    Wt = Y1 + M;
    if (teaser) {
      X2 = Wt + M;
      Y2 = X2 + M;
    }
  } else {
    X0 = A.f64;
    X1 = X0 + L;
    Y0 = X1 + L;
    Y1 = Y0 + L;
    Wt = Y1 + L;
    if (teaser) {  // the RAW area (3 L doubles) is free until the rotation is known
      X2 = A.f64 + 5 * (size_t)L;
      Y2 = X2 + L;
    }
  }
  // chain TIMs over the sorted clique, XY rows (:817-844, :396-402); scale == 1
  for (int i = tid; i < M; i += nthr) {
    const int root = A.clique[i], leaf = (i != M - 1) ? A.clique[i + 1] : A.clique[0];
    const float4 sr = A.src[root], sl = A.src[leaf], tr = A.tgt[root], tl = A.tgt[leaf];
    X0[i] = (double)sl.x - (double)sr.x;
    X1[i] = (double)sl.y - (double)sr.y;
    Y0[i] = ((double)tl.x - (double)tr.x) * (1 / 1.0);
    Y1[i] = ((double)tl.y - (double)tr.y) * (1 / 1.0);
    Wt[i] = 1.0;
    if (teaser) {
      X2[i] = (double)sl.z - (double)sr.z;
      Y2[i] = ((double)tl.z - (double)tr.z) * (1 / 1.0);
    }
  }
  __syncthreads();

  const long long t_fin1 = clock64();
  // ---- GNC-TLS (wavefront 0), reference :430-572
  if (wave == 0 && teaser) {
    double Rg[9], costg;
    int itersg;
    gnc3_wave(sum64_slot(lane), X0, X1, X2, Y0, Y1, Y2, Wt, M, prm.noise_bound * (2 / 1.0), prm.rotation_gnc_factor,
              prm.rotation_max_iterations, prm.rotation_cost_threshold, Rg, &costg, &itersg);
    if (lane == 0) {
      for (int a = 0; a < 9; ++a) s_R[a] = Rg[a];
      s_cost = costg;
      s_iters = itersg;
    }
  } else if (wave == 0) {
    double Rg[4], costg;
    int itersg;
    gnc_wave(sum64_slot(lane), X0, X1, Y0, Y1, Wt, M, prm.noise_bound * (2 / 1.0), prm.rotation_gnc_factor,
             prm.rotation_max_iterations, prm.rotation_cost_threshold, Rg, &costg, &itersg);
    if (lane == 0) {
      s_R[0] = Rg[0];
      s_R[1] = Rg[1];
      s_R[2] = Rg[2];
      s_R[3] = Rg[3];
      s_cost = costg;
      s_iters = itersg;
    }
  } else if (wave == 1 && lane == 0 && !(A.range_pre && A.range_pre[0] == prm.cote_noise_bound * sqrt(prm.cbar2))) {
    // (only without the host's table of these sums, SolverBufs::range_pre; idle while wave 0 runs GNC) COTE needs the sum of its N ranges in the reference's sequential order (:660) — N is only
    // known after the rotation inliers, so the sums for EVERY n are laid down now: a chain of M additions that used to sit
    // on COTE's critical path (2 us at 250 members).  Behind the GNC arrays in LDS when there is room, else in scratch.
    const double rg = prm.cote_noise_bound * sqrt(prm.cbar2);
    double* pre = use_lds && (size_t)8 * M * sizeof(double) <= (size_t)FIN_LDS_BYTES ? fin_lds + 7 * (size_t)M
                                                                                      : A.f64 + 68 * (size_t)L;
    double r = 0;  // pre[n - 1] = rg + rg + ... (n terms)
    int i = 0;
    for (; i + 4 <= M; i += 4) {
      const double r1 = r + rg, r2 = r1 + rg, r3 = r2 + rg, r4 = r3 + rg;
      pre[i] = r1;
      pre[i + 1] = r2;
      pre[i + 2] = r3;
      pre[i + 3] = r4;
      r = r4;
    }
    for (; i < M; ++i) {
      r += rg;
      pre[i] = r;
    }
  }
  __syncthreads();

  const long long t_fin2 = clock64();
  // ---- rotation (yaw block, optional R * RyRx :419-423), rotation inliers (:857-874)
  double R[9] = {s_R[0], s_R[1], 0, s_R[2], s_R[3], 0, 0, 0, 1};
  if (teaser)
    for (int a = 0; a < 9; ++a) R[a] = s_R[a];
  if (prm.using_pre_estimated_ryrx) {
    double Rn[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        Rn[3 * r + c] =
            (R[3 * r] * prm.ryrx[c] + R[3 * r + 1] * prm.ryrx[3 + c]) + R[3 * r + 2] * prm.ryrx[6 + c];
    for (int i = 0; i < 9; ++i) R[i] = Rn[i];
  }
  if (tid == 0) s_nrot = 0;
  __syncthreads();
  if (wave == 0) {
    int base = 0;
    for (int i0 = 0; i0 < M; i0 += 64) {
      const int i = i0 + lane;
      bool in = false;
      if (i < M) {
        const int prev = (i == 0) ? M - 1 : i - 1;
        in = (Wt[prev] >= 0.4) && (Wt[i] >= 0.4);
      }
      const u64 bal = __ballot(in);
      if (in) A.rot_inl[base + __popcll(bal & lanemask_lt())] = i;
      base += __popcll(bal);
    }
    if (lane == 0) s_nrot = base;
  }
  __syncthreads();
  const int NR = s_nrot;
  const bool use_rot = prm.using_rot_inliers_when_estimating_cote && NR > 0;
  const int N = use_rot ? NR : M;
  __shared__ double s_range_sum;
  if (tid == 0) {
    const double* pre = use_lds && (size_t)8 * M * sizeof(double) <= (size_t)FIN_LDS_BYTES ? fin_lds + 7 * (size_t)M
                                                                                            : A.f64 + 68 * (size_t)L;
    const bool table = A.range_pre && A.range_pre[0] == prm.cote_noise_bound * sqrt(prm.cbar2);
    s_range_sum = N > 0 ? (table ? A.range_pre[N] : pre[N - 1]) : 0.0;
  }
  int* sel = A.i32;  // N selected vertex ids
  for (int i = tid; i < N; i += nthr) sel[i] = use_rot ? A.clique[A.rot_inl[i]] : A.clique[i];
  __syncthreads();

  // raw_translation = dst - R * src'  (src' = RyRx*src only in the non-rot-inlier branch, :879-899)
  double* RAW = A.f64 + 5 * (size_t)L;  // 3 arrays of L
  for (int i = tid; i < N; i += nthr) {
    const float4 s4 = A.src[sel[i]], t4 = A.tgt[sel[i]];
    double x = s4.x, y = s4.y, z = s4.z;
    if (!use_rot && prm.using_pre_estimated_ryrx) {
      const double* Y = prm.ryrx;
      const double nx = (Y[0] * x + Y[1] * y) + Y[2] * z, ny = (Y[3] * x + Y[4] * y) + Y[5] * z,
                   nz = (Y[6] * x + Y[7] * y) + Y[8] * z;
      x = nx;
      y = ny;
      z = nz;
    }
    RAW[i] = (double)t4.x - ((R[0] * x + R[1] * y) + R[2] * z);
    RAW[L + i] = (double)t4.y - ((R[3] * x + R[4] * y) + R[5] * z);
    RAW[2 * (size_t)L + i] = (double)t4.z - ((R[6] * x + R[7] * y) + R[8] * z);
  }
  __syncthreads();

  const long long t_fin3 = clock64();
  // ---- COTE (reference estimate(), :618-747).  The three axes are independent: threads 256a..256a+255
  // handle axis a (see cote_axis4).
  const double range = prm.cote_noise_bound * sqrt(prm.cbar2);
  const int nc = 2 * N;
  const int ax = tid >> 8, tl = tid & 255;
  const bool act = ax < 3;
  const int axc = act ? ax : 0;
  __shared__ double s_axis_est[3], s_bc[3][4], s_redc[3][4];
  __shared__ int s_axis_ncard[3], s_redi[3][4];
  const double* X = RAW + (size_t)axc * L;
  // bytes one axis needs: sorted X nc*8, sorted positions nc*4, six term / running-sum arrays of nc doubles
  const size_t a_arr = (((size_t)nc * 8 + 15) & ~(size_t)15), a_spos = a_arr,
               a_T = a_spos + (((size_t)nc * 4 + 15) & ~(size_t)15), a_total = a_T + 6 * a_arr;
  __syncthreads();  // GNC arrays in LDS are dead from here on
  CoteOut co;
  if (3 * a_total <= (size_t)FIN_LDS_BYTES) {
    char* base = (char*)fin_lds + (size_t)axc * a_total;  // LDS: pointers derive from the shared array
    co = cote_axis4(act, tl, X, N, nc, range, nullptr, prm.cote_median, (double*)base, (int*)(base + a_spos),
                    (double*)(base + a_T), s_bc[axc], s_redc[axc], s_redi[axc], ax == 0 ? st->pad + 6 : nullptr, axc,
                    &s_range_sum);
  } else {
    double* gf = A.f64 + 8 * (size_t)L + (size_t)axc * 20 * (size_t)L;  // 20 L doubles of global scratch per axis
    int* gi = A.i32 + 2 * (size_t)L + (size_t)axc * 6 * (size_t)L;     // 6 L ints per axis
    co = cote_axis4(act, tl, X, N, nc, range, nullptr, prm.cote_median, gf, gi, gf + 2 * (size_t)L, s_bc[axc], s_redc[axc],
                    s_redi[axc], ax == 0 ? st->pad + 6 : nullptr, axc, &s_range_sum);
  }
  if (act && tl == 0) {
    s_axis_est[ax] = co.est;
    s_axis_ncard[ax] = co.ncard;
    res->n_card[ax] = co.ncard;
  }
  __syncthreads();
  double tr[3] = {s_axis_est[0], s_axis_est[1], s_axis_est[2]};
  unsigned char* inl = (unsigned char*)(A.i32 + 1 * (size_t)L);  // N bytes
  for (int i = tid; i < N; i += nthr) {
    bool in = true;
#pragma unroll
    for (int a3 = 0; a3 < 3; ++a3) in = in && (fabs(RAW[(size_t)a3 * L + i] - tr[a3]) <= range);
    inl[i] = in ? 1 : 0;
  }
  __syncthreads();
  // ---- final inliers (:914-930) and the 4x4
  if (wave == 0) {
    int base = 0;
    for (int i0 = 0; i0 < N; i0 += 64) {
      const int i = i0 + lane;
      const bool in = (i < N) && inl[i];
      const u64 bal = __ballot(in);
      if (in) A.final_inl[base + __popcll(bal & lanemask_lt())] = sel[i];
      base += __popcll(bal);
    }
    if (lane == 0) s_nfinal = base;
  }
  __syncthreads();
  if (tid == 0) {
    const long long t_fin4 = clock64();
    st->pad[1] = (int)((t_fin1 - t_fin0) >> 4);
    st->pad[2] = (int)((t_fin2 - t_fin1) >> 4);
    st->pad[3] = (int)((t_fin3 - t_fin2) >> 4);
    st->pad[4] = (int)((t_fin4 - t_fin3) >> 4);
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) res->T[4 * r + c] = R[3 * r + c];
      res->T[4 * r + 3] = tr[r];
    }
    res->T[12] = res->T[13] = res->T[14] = 0;
    res->T[15] = 1;
    res->cost = s_cost;
    res->gnc_iters = s_iters;
    res->n_rot_inliers = NR;
    res->n_final = s_nfinal;
    res->valid = 1;
    res->status = QTR_OK;
  }
  export_result(A.mail, res, st, A.seq);
}

// =================================================================================================
// host-side launchers (called from capi.hip)
#define LAUNCH_SV(kern, a, grid, block, lds, st, ...)                                                     \
  do {                                                                                                    \
    if ((a).ext)                                                                                          \
      hipLaunchKernelGGL((kern<true>), grid, block, lds, st, (ViewExt<SolverView>{(a).ext, {0, 0, 0}}), (a).one, \
                         ##__VA_ARGS__);                                                                  \
    else                                                                                                  \
      hipLaunchKernelGGL((kern<false>), grid, block, lds, st, (ViewExt<SolverView>{nullptr, {0, 0, 0}}), (a).one, \
                         ##__VA_ARGS__);                                                                  \
  } while (0)
#define LAUNCH_SV_T(kern, T, a, grid, block, lds, st)                                                     \
  do {                                                                                                    \
    if ((a).ext)                                                                                          \
      hipLaunchKernelGGL((kern<true, T>), grid, block, lds, st, (ViewExt<SolverView>{(a).ext, {0, 0, 0}}), (a).one); \
    else                                                                                                  \
      hipLaunchKernelGGL((kern<false, T>), grid, block, lds, st, (ViewExt<SolverView>{nullptr, {0, 0, 0}}), (a).one); \
  } while (0)

// k_hcore_async's workgroups wait for one another, so all workgroups of a pair have to be resident together.  A launch
// that has the device to itself may use every compute unit; `share` launches that may run side by side (the lanes of a
// batch, the slots of a handle driven from several threads) each take at most 1 / share of them: with in-order dispatch
// inside a launch at most ONE pair per launch is partly resident at any time, so share x nwg <= compute units means some
// pair is always complete and makes progress.  (If the promise is broken — another process on the device — the kernel's
// own residency timeout sends the pair to the peeling workgroup: slower, never wrong.)
static thread_local int t_hca_share = 1;
void solver_set_hca_share(int share) { t_hca_share = share < 1 ? 1 : share; }

// workgroups of k_hcore_async that are certainly co-resident: one per compute unit
static int hca_max_workgroups() {
  static int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
      cus = 64;
    (void)hipGetLastError();
    return cus;
  }();
  return n;
}

hipError_t solver_init_attributes() {
  hipError_t e;
#define SET_LDS(kern, bytes)                                                                                         \
  if ((e = hipFuncSetAttribute((const void*)kern<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) \
    return e;                                                                                                        \
  if ((e = hipFuncSetAttribute((const void*)kern<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess)  \
    return e;
  SET_LDS(k_finalize, FIN_LDS_BYTES)
#ifdef QTR_TEST_ENGINES
  SET_LDS(k_kcore, 156 * 1024)
#endif
  SET_LDS(k_hcore_async, 152 * 1024)  // (150 KB are used; 4.4 KB are static)
  SET_LDS(k_rank_sort, 156 * 1024)
  SET_LDS(k_clique_first, CF_LDS_BYTES)
  SET_LDS(k_clique_batch_lds, 156 * 1024)
  SET_LDS(k_permute, 64 * 1024)
  SET_LDS(k_permute_scatter, 96 * 1024)
#undef SET_LDS
  return hipSuccess;
}
size_t solver_scratch_bytes(int Lcap) {
  const size_t W = (size_t)(Lcap + 63) / 64;
  size_t b = 0;
  b += 2 * (size_t)Lcap * ((W + 3) & ~(size_t)3) * 8;  // bm (row stride: a multiple of four words), adjP
  b += 8 * (size_t)Lcap * 4;            // deg, core, perm, rankof, Kp, picks, gsz, (spare)
  b += 3 * (size_t)Lcap * 4;            // clique, rot_inl, final_inl
  b += 72 * (size_t)Lcap * 8;           // f64
  b += 24 * (size_t)Lcap * 4;           // i32
  b += W * 8 + 4096;
  b += (size_t)CLIQUE_BATCH * Lcap * 4;  // picks_buf
  b += ((size_t)Lcap + 2) * 8 + 256;     // range_pre
  return b;
}

void solver_carve(SolverBufs& B, void* base, int Lcap) {
  char* p = (char*)base;
  const size_t W = (size_t)(Lcap + 63) / 64;
  auto take = [&](size_t bytes) {
    char* r = p;
    p += (bytes + 255) & ~(size_t)255;
    return (void*)r;
  };
  B.Lcap = Lcap;
  B.bm = (u64*)take((size_t)Lcap * ((W + 3) & ~(size_t)3) * 8);
  B.adjP = (u64*)take((size_t)Lcap * W * 8);
  B.deg = (int*)take((size_t)Lcap * 4);
  B.core = (int*)take((size_t)Lcap * 4);
  B.perm = (int*)take((size_t)Lcap * 4);
  B.rankof = (int*)take((size_t)Lcap * 4);
  B.Kp = (int*)take((size_t)Lcap * 4);
  B.picks = (int*)take((size_t)Lcap * 4);
  B.gsz = (int*)take((size_t)Lcap * 4);
  B.clique = (int*)take((size_t)Lcap * 4);
  B.rot_inl = (int*)take((size_t)Lcap * 4);
  B.final_inl = (int*)take((size_t)Lcap * 4);
  B.f64 = (double*)take(72 * (size_t)Lcap * 8);
  B.i32 = (int*)take(24 * (size_t)Lcap * 4);
  B.member_bits = (u64*)take(W * 8);
  B.picks_buf = (int*)take((size_t)CLIQUE_BATCH * Lcap * 4);
  B.range_pre = (double*)take(((size_t)Lcap + 2) * 8);
  B.range_rg = -1.0;
  B.st = (SolverState*)take(sizeof(SolverState));
  B.res = (qtr_result*)take(sizeof(qtr_result));
}

static SolverView make_solver_view(const SolverBufs& B, const float4* src, const float4* tgt, int L) {
  SolverView V;
  memset(&V, 0, sizeof(V));
  // per-block degree bytes live in the clique search's pick lists (unused until the descents start): ceil(L / 64) x Lp
  // bytes of its 4096 L; a view without points (qtr_max_clique) carries finished degrees in deg instead
  V.Lp = (L + 63) & ~63;
  V.degp = src ? (const unsigned char*)B.picks_buf : nullptr;
  V.src = src;
  V.tgt = tgt;
  V.L = L;
  V.W = (L + 63) / 64;
  // row stride of bm: rows built by k_graph_build start on 32-byte boundaries (it stores four words at a time); a matrix
  // that was handed in (qtr_max_clique) keeps the caller's packed layout
  V.Wb = src ? ((V.W + 3) & ~3) : V.W;
  V.bm = B.bm;
  V.adjP = B.adjP;
  V.deg = B.deg;
  V.core = B.core;
  V.perm = B.perm;
  V.rankof = B.rankof;
  V.Kp = B.Kp;
  V.picks = B.picks;
  V.gsz = B.gsz;
  V.clique = B.clique;
  V.rot_inl = B.rot_inl;
  V.final_inl = B.final_inl;
  V.f64 = B.f64;
  V.i32 = B.i32;
  V.member_bits = B.member_bits;
  V.picks_buf = B.picks_buf;
  V.st = B.st;
  V.res = B.res;
  V.mail = B.mail;
  V.seq = B.mail_seq;
  V.range_pre = (B.range_rg > 0.0) ? B.range_pre : nullptr;
  return V;
}
// one pair in the kernel arguments, or a group through the stage
static hipError_t solver_args(SolverArgs& a, const SolverView* views, int G, ViewStage* stage, hipStream_t st) {
  a.one = views[0];
  a.ext = nullptr;
  if (G > 1) {
    a.ext = (const SolverView*)stage_push(stage, views, sizeof(SolverView) * (size_t)G, st);
    if (!a.ext) return hipErrorOutOfMemory;
  }
  return hipSuccess;
}

// the solver state of every pair starts from zero (was a hipMemsetAsync per pair)
template <bool EXT>
__global__ void k_solver_reset(ViewExt<SolverView> x, SolverView one, int second_run) {
  const SolverView& V = EXT ? x.ext[blockIdx.z] : one;
  int* p = (int*)V.st;
  if (threadIdx.x < (int)(sizeof(SolverState) / 4)) p[threadIdx.x] = 0;
  if (threadIdx.x == 0) V.st->pad[12] = second_run;  // statistics: this is the stage's second run (see solver_continue)
}

static void launch_finalize(const SolverArgs& a, int G, const qtr_params& prm, hipStream_t stream, int scan_batch = 0) {
  LAUNCH_SV(k_finalize, a, dim3(1, 1, G), dim3(FIN_THREADS), (size_t)FIN_LDS_BYTES, stream, prm, scan_batch);
}

// K-core -> rank relabelling -> permuted adjacency -> the first two clique rounds, for the G pairs of `a` (Lmax = the
// largest L among them: grids, LDS sizes and kernel variants are chosen for it; every kernel reads its own pair's L).
// Expects the bit matrices in bm and the degrees in deg; everything stays on `stream`.
// which core-number path a graph of L vertices takes (QTR_KCORE=peel | sweeps: the older ones, kept for comparison)
static bool kcore_peel_only() {
  static const bool v = [] {
    const char* e = QTR_ENGINE_ENV("QTR_KCORE");
    return e && strcmp(e, "peel") == 0;
  }();
  return v;
}
static bool kcore_sweeps() {
  static const bool v = [] {
    const char* e = QTR_ENGINE_ENV("QTR_KCORE");
    return e && strcmp(e, "sweeps") == 0;
  }();
  return v;
}
static int hcore_min_l() {  // (the control words sit in V.perm: L must exceed HCA_CTL_DONE + HCA_MAXWG = 1088 in any case)
  static const int v = [] {
    const char* e = QTR_ENGINE_ENV("QTR_HCORE_MIN_L");
    return max(1280, e ? atoi(e) : 1280);  // (measured, ms per solve, peel against this: 0.174 / 0.159 at L = 1500,
                                            // 0.213 / 0.174 at 2000, 0.259 / 0.199 at 2500, 0.304 / 0.200 at 3000)
  }();
  return v;
}
static bool hcore_planned(int L) { return !kcore_peel_only() && L > hcore_min_l() && L <= 65536; }  // (16-bit ids and values in k_hcore_async)
static bool hcore_async_planned(int L) { return hcore_planned(L) && !kcore_sweeps(); }

// hcore_prepared: k_graph_build has left k_hcore_async's clean slate (values, control words)
// defer_last_scan: the caller's next launch (k_finalize) replays the last round itself; returns that round's next_batch
// argument (0: nothing left to replay)
// exact_cores: k_hcore_async without its floor (the second run of a pair whose first came back empty; the k-core
// heuristic, which reads every core number)
static int clique_stage_launch(const SolverArgs& a, int G, int L, int mode, double kcore_thr, hipStream_t stream,
                               bool hcore_prepared, bool defer_last_scan, bool exact_cores = false) {
  int deferred = 0;
  const int W = (L + 63) / 64;
  static const bool dbg_sync = QTR_ENGINE_ENV("QTR_DEBUG_SYNC") != nullptr;  // name every kernel as it completes
#define CS_DBG(name)                                                                                   \
  do {                                                                                                 \
    if (dbg_sync) {                                                                                    \
      const hipError_t de = hipStreamSynchronize(stream);                                              \
      fprintf(stderr, "[clique stage] %s done (%s)\n", name, hipGetErrorString(de));                   \
    }                                                                                                  \
  } while (0)
  {
    const bool q_in_lds = (size_t)2 * L * sizeof(int) <= (size_t)128 * 1024;
    const size_t kc_lds = (size_t)(q_in_lds ? 2 : 1) * L * sizeof(int);
    const size_t bm_bytes = (size_t)L * W * 8;
    const bool kc_single_wave = (L <= 256 * KCL_VPT);
    // (see the clique rounds below) level-parallel core numbers and rows in LDS: the first round takes CLIQUE_BATCH starts
    const bool merged_first_round = kc_single_wave && ((size_t)L * W * 8 + (size_t)4 * L * sizeof(int) <= (size_t)150 * 1024) &&
                                    QTR_ENGINE_ENV("QTR_CLIQUE_ROUND0") == nullptr;
    if (kc_single_wave) {
      const dim3 kgrid(L > 1 ? L - 1 : 1, 1, G);
      if (W <= 4) LAUNCH_SV_T(k_kcore_levels, 1, a, kgrid, dim3(256), 0, stream);
      else if (W <= 8) LAUNCH_SV_T(k_kcore_levels, 2, a, kgrid, dim3(256), 0, stream);
      else if (W <= 12) LAUNCH_SV_T(k_kcore_levels, 3, a, kgrid, dim3(256), 0, stream);
      else if (W <= 16) LAUNCH_SV_T(k_kcore_levels, 4, a, kgrid, dim3(256), 0, stream);
      else LAUNCH_SV_T(k_kcore_levels, 5, a, kgrid, dim3(256), 0, stream);
      LAUNCH_SV(k_kcore_collect_rank, a, dim3(1, 1, G), dim3(1024), 0, stream, merged_first_round ? CLIQUE_BATCH : 1);
    } else {
      const int lds_bitmap = (q_in_lds && kc_lds + 8 + bm_bytes <= (size_t)150 * 1024) ? 1 : 0;
      const bool hcore = hcore_planned(L), hc_sweeps = kcore_sweeps();
      bool after_async = false;
      if (hcore && !hc_sweeps) {
        // one resident workgroup per compute unit at most (they wait for one another); a group of pairs shares the device
        int nwg = min(min(hca_max_workgroups(), HCA_MAXWG), max(1, (L + 15) / 16));
        nwg = max(1, min(nwg, hca_max_workgroups() / t_hca_share));
        static const bool no_floor = [] {  // (QTR_HCORE_FLOOR=0: comparison runs of the test build)
          const char* e = QTR_ENGINE_ENV("QTR_HCORE_FLOOR");
          return e && atoi(e) == 0;
        }();
        int allow_floor = (exact_cores || no_floor || mode == QTR_INLIER_KCORE_HEU) ? 0 : 1;
        // a single pair's large graph: one more workgroup, the scout (hca_scout) — one of the resident set, so a launch that
        // would fill the device gives it one of its places
        static const bool no_scout = [] {  // (QTR_HCORE_SCOUT=0: comparison runs of the test build)
          const char* e = QTR_ENGINE_ENV("QTR_HCORE_SCOUT");
          return e && atoi(e) == 0;
        }();
        static const int scout_min_l = [] {  // (QTR_HCORE_SCOUT_MIN_L: the test build's sweeps run it on small graphs too)
          const char* e = QTR_ENGINE_ENV("QTR_HCORE_SCOUT_MIN_L");
          return e ? atoi(e) : 8192;
        }();
        const bool scout = allow_floor && !no_scout && G == 1 && L > scout_min_l && L <= 32768 && nwg >= 2;
        if (scout) {
          allow_floor = 2;
          if (nwg + 1 > min(HCA_MAXWG, hca_max_workgroups() / t_hca_share)) nwg -= 1;
        }
        if (G > 1) {
          // A group of pairs: an iteration of a workgroup is one snapshot round trip (~1.2 us, whatever it owns) plus its
          // rows (~0.6 us per sixteen), so the device does the most work per microsecond when ALL pairs of the launch are
          // resident (or nearly) with few, large workgroups each.  Measured on 256 composite pairs (L = 5000, two lanes of
          // sixteen pairs), registrations/s by workgroups per pair: 64 -> 4408, 32 -> 4828, 16 -> 4887, 8 -> 4797, 4 -> 3708
          // (95 us of k_hcore_async per pair at 64, profiles/r4a_batch_kernel_stats.txt).
          int per_pair = max(16, hca_max_workgroups() / (t_hca_share * G));
          if (const char* e = QTR_ENGINE_ENV("QTR_HCA_GROUP_WGS")) per_pair = max(1, atoi(e));
          nwg = max(1, min(nwg, per_pair));
        }
        const int Lp = (L + 63) & ~63, R = (L + nwg - 1) / nwg;
        const int Rp = (R + 3) & ~3;
        const size_t fixed = (size_t)2 * Lp + (size_t)4 * (Rp + 4) + (size_t)4 * Rp;
        // the pool of neighbour lists takes what the compute unit's LDS has left (the workgroups run one per unit anyway)
        size_t lds_budget = (size_t)150 * 1024;
        if (G > 1) {
          // In a group the launch of one lane shares the device with the OTHER lane's chain: a workgroup that claims the
          // unit's whole LDS keeps every kernel that needs some out of its unit for as long as it runs (k_rank_sort of the
          // other lane showed 626 us in the trace, all of it waiting for a unit).  Its rows' lists need ~2 bytes per edge
          // endpoint; what does not fit the smaller pool keeps its bit row (slower, same result).  256 composite pairs,
          // registrations/s by budget: 150 KB 4771, 128 KB 4833, 104 KB 4854, 88 KB 4891, 72 KB 4886.
          lds_budget = (size_t)88 * 1024;
          if (const char* e = QTR_ENGINE_ENV("QTR_HCA_GROUP_LDS_KB")) lds_budget = (size_t)max(32, atoi(e)) * 1024;
          lds_budget = max(lds_budget, fixed + 4096);
        }
        const int pool_entries = (int)((lds_budget - fixed) / 2) & ~7;
        if (!hcore_prepared) LAUNCH_SV(k_hcore_async_init, a, dim3((max(L, 4096) + 255) / 256, 1, G), dim3(256), 0, stream);
        LAUNCH_SV(k_hcore_async, a, dim3(nwg + (scout ? 1 : 0), 1, G), dim3(HCA_THREADS), fixed + (size_t)2 * pool_entries, stream,
                  pool_entries, allow_floor);
        after_async = true;
      }
#ifdef QTR_TEST_ENGINES
      else if (hcore) {
        LAUNCH_SV(k_hcore_init, a, dim3((L + 255) / 256, 1, G), dim3(256), 0, stream);
        for (int it = 0; it < HC_MAXIT; ++it) LAUNCH_SV(k_hcore_sweep, a, dim3((L + 3) / 4, 1, G), dim3(256), 0, stream, it);
        LAUNCH_SV(k_hcore_finish, a, dim3(1, 1, G), dim3(1024), 0, stream, 0);
      }
#endif
      static const bool rank_quadratic = QTR_ENGINE_ENV("QTR_RANK_QUADRATIC") != nullptr;
      const size_t kc_bytes = kc_lds + (lds_bitmap ? bm_bytes + 8 : 0);
      // the peeling workgroup rides in k_rank_sort's launch (it is the fallback behind k_hcore_async, and the core-number
      // kernel of the graphs in between); the older chains keep its own launch
      const bool own_kcore_launch = rank_quadratic || (hcore && !after_async);
#ifdef QTR_TEST_ENGINES
      if (own_kcore_launch)
        LAUNCH_SV(k_kcore, a, dim3(1, 1, G), dim3(1024), kc_bytes, stream, q_in_lds ? 0 : 1, lds_bitmap, hcore ? 1 : 0);
#endif
      CS_DBG("k_kcore");
      int slices = (L + 1023) / 1024;
      if (slices > 32) slices = 32;
#ifdef QTR_TEST_ENGINES
      if (rank_quadratic) {
        LAUNCH_SV(k_rank_partial, a, dim3((L + 255) / 256, slices, G), dim3(256), 0, stream);
        LAUNCH_SV(k_rank_finish, a, dim3((L + 255) / 256, 1, G), dim3(256), 0, stream);
        LAUNCH_SV(k_clique_init, a, dim3(1, 1, G), dim3(64), 0, stream);
      } else
#else
      (void)slices;
#endif
      {
        const int kcore_mode = own_kcore_launch ? 0 : after_async ? 2 : 1;
        LAUNCH_SV(k_rank_sort, a, dim3(1, 1, G), dim3(RS_THREADS), max((size_t)16 * RS_BINS * 4, kcore_mode ? kc_bytes : (size_t)0),
                  stream, after_async ? 1 : 0, kcore_mode, q_in_lds ? 0 : 1, lds_bitmap);
      }
      CS_DBG("rank");
    }
    if (L <= PM2_MAXL)
      LAUNCH_SV(k_permute_scatter, a, dim3((L + PM2_ROWS - 1) / PM2_ROWS, 1, G), dim3(256),
                (size_t)4 * W * 8 + (size_t)2 * ((L + 63) & ~63), stream);
    else
      LAUNCH_SV(k_permute, a, dim3((L + PM_ROWS - 1) / PM_ROWS, 1, G), dim3(256), (size_t)PM_ROWS * W * 8, stream);
    CS_DBG("k_permute");
    if (mode == QTR_INLIER_KCORE_HEU) LAUNCH_SV(k_kcore_heu, a, dim3(1, 1, G), dim3(256), 0, stream, kcore_thr);
    {
      const int BATCH = CLIQUE_BATCH;
      // round 0: the single top-ranked start; round 1..: BATCH starts each
      const size_t cl_lds = (size_t)L * W * 8 + (size_t)4 * L * sizeof(int);  // matrix + four per-wave pick lists
      const bool lds_rows = cl_lds <= (size_t)150 * 1024;
      if (merged_first_round) {
        // Small graphs (rows in LDS): the single start of round 0 is speculated together with the next BATCH - 1 — the
        // replay in k_clique_scan gives the sequential result whatever bound the descents were started with (a descent
        // that strays below the current bound cannot return more than that bound: every member of a clique of size s
        // has K >= s), and a round of parallel descents costs what the one descent does.  Two launches instead of four.
        LAUNCH_SV(k_clique_batch_lds, a, dim3(BATCH / 4, 1, G), dim3(256), cl_lds, stream, 0);
        if (L > BATCH) {  // (a second round only exists for more than BATCH vertices)
          LAUNCH_SV(k_clique_scan, a, dim3(1, 1, G), dim3(64), 0, stream, BATCH);
          LAUNCH_SV(k_clique_batch_lds, a, dim3(BATCH / 4, 1, G), dim3(256), cl_lds, stream, 0);
        }
        if (defer_last_scan) deferred = BATCH;
        else LAUNCH_SV(k_clique_scan, a, dim3(1, 1, G), dim3(64), 0, stream, BATCH);
      } else {
      // (round 0 is one workgroup: its replay rides in the same launch)
      if (lds_rows)
        LAUNCH_SV(k_clique_batch_lds, a, dim3(1, 1, G), dim3(256), cl_lds, stream, BATCH);
      else
        LAUNCH_SV(k_clique_first, a, dim3(1, 1, G), dim3(CF_THREADS), (size_t)CF_LDS_BYTES, stream, BATCH);
      CS_DBG("clique round 0");
      // Rounds 0 and 1 are enqueued unconditionally: the heuristic nearly always terminates within them (the
      // first start finds the large clique, the second batch only confirms that no start can beat it).  The
      // host checks `done` once, together with the result record; solver_continue() handles the rare rest.
      // (Round 5 measured leaving round 1 to solver_continue: the headline's planted clique is settled by round 0 and
      // saves the 4.7 us dispatch, but a matcher's own correspondences — use_tuple_test = 0, L = 2104, clique 87 — need
      // round 1 and paid a host round trip for it, 0.637 -> 0.680 ms, and at L = 20 k the sweep then started from a
      // small bound and re-ran a long descent per accepted start: 3.3 -> 3.7 ms.  Kept unconditional.)
      if (lds_rows)
        LAUNCH_SV(k_clique_batch_lds, a, dim3(BATCH / 4, 1, G), dim3(256), cl_lds, stream, 0);
      else
        LAUNCH_SV(k_clique_batch, a, dim3(BATCH / 4, 1, G), dim3(256), 0, stream);
      CS_DBG("clique batch 1");
      if (defer_last_scan) deferred = BATCH;
      else LAUNCH_SV(k_clique_scan, a, dim3(1, 1, G), dim3(64), 0, stream, BATCH);
      }
      CS_DBG("clique scan 1");
    }
  }
#undef CS_DBG
  return deferred;
}

// Rare path: the two unconditional clique rounds did not finish the search.  Runs further rounds (one host
// check per round) and the finalisation again.
// redo_cores (the state's flag of that name): the search ran under k_hcore_async's floor and found nothing it can vouch
// for — the whole stage again from the bit matrix, with exact core numbers, before any further round.
hipError_t solver_continue(const SolverBufs& B, const float4* src, const float4* tgt, int L, const qtr_params& prm,
                           hipStream_t stream, int* pinned_state, int redo_cores) {
  hipError_t e;
  int guard = 0;
  SolverView V = make_solver_view(B, src, tgt, L);
  V.degp = nullptr;  // (the degrees are finished and in deg; the per-block counts lay where the pick lists are now)
  SolverArgs a;
  if ((e = solver_args(a, &V, 1, nullptr, stream)) != hipSuccess) return e;
  bool redo = redo_cores != 0;
  while (true) {
    if (redo) {  // (at most once: the second run has no floor)
      redo = false;
      LAUNCH_SV(k_solver_reset, a, dim3(1, 1, 1), dim3(64), 0, stream, 1);
      clique_stage_launch(a, 1, L, QTR_INLIER_PMC_HEU, 0.0, stream, false, false, true);
    } else {
      // every remaining start in one speculated round (see k_clique_sweep)
      LAUNCH_SV(k_clique_sweep, a, dim3(max(1, min((L + 3) / 4, 2048)), 1, 1), dim3(256), 0, stream);
      LAUNCH_SV(k_clique_scan_all, a, dim3(1, 1, 1), dim3(CSA_THREADS), 0, stream);
    }
    if ((e = hipMemcpyAsync(pinned_state, B.st, sizeof(SolverState), hipMemcpyDeviceToHost, stream)) != hipSuccess)
      return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    const SolverState* hs = (const SolverState*)pinned_state;
    if (hs->done) break;
    if (hs->redo_cores) {  // the rounds under the floor's bound ended empty-handed
      redo = true;
      guard = 0;
      continue;
    }
    // The sweep + scan-all pair goes over EVERY remaining start in one round and ends with done = 1 (or redo_cores), and a
    // second run of the stage is followed by at most one more sweep: a state that is still open after that is an internal
    // error, and it is reported as one — k_finalize exports valid = 0 with status QTR_OK while the search is not through,
    // which a caller could not tell from "no clique"
    if (++guard > 3) return hipErrorUnknown;
  }
  if (src) launch_finalize(a, 1, prm, stream);
  return hipGetLastError();
}

// PMC_EXACT replaced the clique after the first finalisation: estimate again from the state as it is now.
hipError_t solver_refinalize(const SolverBufs& B, const float4* src, const float4* tgt, int L, const qtr_params& prm,
                             hipStream_t stream) {
  (void)hipGetLastError();
  const SolverView V = make_solver_view(B, src, tgt, L);
  SolverArgs a;
  hipError_t e = solver_args(a, &V, 1, nullptr, stream);
  if (e != hipSuccess) return e;
  launch_finalize(a, 1, prm, stream);
  return hipGetLastError();
}

// qtr_max_clique: bit matrix (device, L x ceil(L/64) words) -> degrees -> clique search.
hipError_t clique_only_enqueue(const SolverBufs& B, const u64* d_adj, int L, int mode, double kcore_thr,
                               hipStream_t stream) {
  const int W = (L + 63) / 64;
  hipError_t e;
  (void)hipGetLastError();
  if ((e = hipMemsetAsync(B.st, 0, sizeof(SolverState), stream)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(B.res, 0, sizeof(qtr_result), stream)) != hipSuccess) return e;
  if (L > 0) {
    if (d_adj != B.bm &&
        (e = hipMemcpyAsync(B.bm, d_adj, (size_t)L * W * 8, hipMemcpyDeviceToDevice, stream)) != hipSuccess)
      return e;
    hipLaunchKernelGGL(k_row_degrees, dim3((L + 3) / 4), dim3(256), 0, stream, B.bm, L, W, B.deg);
    const SolverView V = make_solver_view(B, nullptr, nullptr, L);
    SolverArgs a;
    if ((e = solver_args(a, &V, 1, nullptr, stream)) != hipSuccess) return e;
    clique_stage_launch(a, 1, L, mode, kcore_thr, stream, false, false);
  }
  return hipGetLastError();
}

hipError_t clique_only_finish(const SolverBufs& B, int L, hipStream_t stream) {
  const int W = (L + 63) / 64;
  hipLaunchKernelGGL(k_clique_only, dim3(1), dim3(256), 0, stream, B.st, B.member_bits, B.picks, B.perm, B.clique, W,
                     B.res, B.mail, B.mail_seq);
  return hipGetLastError();
}

// the table of SolverBufs::range_pre for the range of `prm`, laid down when the range differs from the one it holds (the
// buffers are idle then: a slot runs one call at a time).  Host arithmetic = device arithmetic (IEEE additions, a correctly
// rounded square root); k_finalize still compares [0] with its own range and runs the chain itself if they differ.
static hipError_t ensure_range_table(const SolverBufs& B, const qtr_params& prm) {
  const double rg = prm.cote_noise_bound * sqrt(prm.cbar2);
  if (!B.range_pre || !(rg > 0.0) || B.range_rg == rg) return hipSuccess;
  std::vector<double> t((size_t)B.Lcap + 1);
  t[0] = rg;
  double r = 0;
  for (int n = 1; n <= B.Lcap; ++n) {
    r += rg;
    t[(size_t)n] = r;
  }
  const hipError_t e = hipMemcpy(B.range_pre, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) B.range_rg = rg;
  return e;
}

// The whole back end of the G pairs of `views` on `stream` (L known on the host).  The clique heuristic normally
// terminates after the first two batches (see k_clique_batch); the host checks `done` with the result record.
static hipError_t solver_launch(const SolverView* views, int G, const qtr_params& prm, ViewStage* stage,
                                hipStream_t stream, hipEvent_t ev_graph, hipEvent_t ev_clique, bool reset_done = false) {
  hipError_t e;
  (void)hipGetLastError();  // a stale sticky error (e.g. timing query on an unrecorded event) is not ours
  SolverArgs a;
  if ((e = solver_args(a, views, G, stage, stream)) != hipSuccess) return e;
  int L = 0;
  for (int g = 0; g < G; ++g) L = max(L, views[g].L);
  // (the state's clean slate rides on k_graph_build when there is one: a launch fewer on the chain)
  if (!reset_done && L <= 0) LAUNCH_SV(k_solver_reset, a, dim3(1, 1, G), dim3(64), 0, stream, 0);
  if (L <= 0 && ev_graph) hipEventRecord(ev_graph, stream);
  const bool prep_hcore = L > 0 && hcore_async_planned(L);
  int scan_batch = 0;
  if (L > 0) {
    const double beta = 2 * prm.noise_bound * sqrt(prm.cbar2);
    {
      const int nb = (L + 63) / 64, nsb = (nb + 3) / 4;
      const int prep = (prep_hcore ? 1 : 0) | (reset_done ? 0 : 2);
      // Two forms of one computation (identical bit matrices).  Tiles: 64 x 64 per workgroup, four waves x 16 rows — many
      // short workgroups, the lower latency while the whole graph is one wave of workgroups (same-box A/B at L = 5000 and
      // at the matcher's L ~ 300: 10 us per registration in its favour).  Strips: 64 x 256 per workgroup, one wave per
      // tile — 30 % fewer vector instructions and sector-sized row stores, the higher throughput once the device is
      // full (L = 20000: 107 against 132 us).
      // Round 6: from GBM_MIN_L correspondences on the squared lengths come off the matrix pipe (k_graph_build_mfma: 14.2 vector
      // instructions per 64 predicates against the strips' 17.3 and the tiles' 24 — 92 against 110 us at L = 20000, 16 against
      // 18.5 at L = 5000); the strips stay as the test build's comparison engine (QTR_GRAPH=strips).
      bool mfma = L >= GBM_MIN_L, tiles = !mfma;
      if (const char* e = QTR_ENGINE_ENV("QTR_GRAPH")) {
        tiles = strcmp(e, "tiles") == 0;
        mfma = strcmp(e, "mfma") == 0;
      }
      if (mfma)
        LAUNCH_SV(k_graph_build_mfma, a, dim3(nsb, nb, G), dim3(GB2_THREADS), 0, stream, beta, prep);
      else if (tiles)
        LAUNCH_SV(k_graph_build_tiles, a, dim3(nb * (nb + 1) / 2, 1, G), dim3(256), 0, stream, beta, graph_margin(beta), prep);
      else  // (grid: column group x row block; the lower-left half returns at once)
        LAUNCH_SV(k_graph_build, a, dim3(nsb, nb, G), dim3(GB2_THREADS), 0, stream, beta, graph_margin(beta), prep);
    }
    if (ev_graph) hipEventRecord(ev_graph, stream);
    scan_batch = clique_stage_launch(a, G, L, prm.inlier_selection_mode, prm.kcore_heuristic_threshold, stream, prep_hcore, true);
  }
  if (ev_clique) hipEventRecord(ev_clique, stream);
  launch_finalize(a, G, prm, stream, scan_batch);
  return hipGetLastError();
}

hipError_t solver_enqueue(const SolverBufs& B, const float4* src, const float4* tgt, int L, const qtr_params& prm,
                          hipStream_t stream, int* pinned_state /* unused */, hipEvent_t ev_graph, hipEvent_t ev_clique,
                          bool reset_done) {
  (void)pinned_state;
  const hipError_t e = ensure_range_table(B, prm);
  if (e != hipSuccess) return e;
  const SolverView V = make_solver_view(B, src, tgt, L);
  return solver_launch(&V, 1, prm, nullptr, stream, ev_graph, ev_clique, reset_done);
}
// the state reset of a run, for callers that can issue it early (the whole-path driver: beside the FPFH chain)
__global__ void k_state_reset(SolverState* st) {
  int* p = (int*)st;
  if (threadIdx.x < (int)(sizeof(SolverState) / 4)) p[threadIdx.x] = 0;
}
hipError_t solver_reset_enqueue(const SolverBufs& B, hipStream_t stream) {
  hipLaunchKernelGGL(k_state_reset, dim3(1), dim3(64), 0, stream, B.st);
  return hipGetLastError();
}

hipError_t solver_enqueue_group(SolverBufs* const* B, int G, const float4* const* src, const float4* const* tgt,
                                const int* L, const qtr_params& prm, ViewStage* stage, hipStream_t stream) {
  std::vector<SolverView> v((size_t)G);
  for (int g = 0; g < G; ++g) {
    const hipError_t e = ensure_range_table(*B[g], prm);
    if (e != hipSuccess) return e;
    v[g] = make_solver_view(*B[g], src[g], tgt[g], L[g]);
  }
  return solver_launch(v.data(), G, prm, stage, stream, nullptr, nullptr);
}

#ifdef QTR_HCA_PROF
extern "C" int qtr_debug_hca_prof(unsigned* out, int n_wg) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hca_prof), (size_t)n_wg * 32);
}
#endif

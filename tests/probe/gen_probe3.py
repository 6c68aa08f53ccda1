#!/usr/bin/env python3
"""gen_probe3.py — writes nn_probe3.hip: hand-scheduled (inline asm, fixed registers) variants of one k_nn_mfma tile
(68 x v_mfma_f32_32x32x2_f32 + the top-2 fold of 64 accumulator values), to find out what the matrix pipe and the VALU
can overlap on gfx950.  Not part of the product.   python gen_probe3.py > nn_probe3.hip

register map (per lane):  v[64:127] accumulator set A (4 x 16), v[128:191] set B, v[40:56] the 17 base-tile operands,
v[192:208] .. query operands (re-used for the four column blocks: timing does not depend on the values),
v20-23 best, v24-27 second, v28-31 "before", v32-35 winning tile, v36-39 temporaries, a[0:127] accumulators of the AGPR
variants.  s20 = pack mask, s21 = tile counter.
"""
import sys

ACC = {"A": 64, "B": 128}


def qreg(agpr):
    return 64 if agpr else 192


def mfma(dst_set, c, kk, agpr=False, first=False):
    base = (0 if dst_set == "A" else 64) + 16 * c if agpr else ACC[dst_set] + 16 * c
    r = ("a[%d:%d]" if agpr else "v[%d:%d]") % (base, base + 15)
    src_c = "0" if first else r
    return "v_mfma_f32_32x32x2_f32 %s, v%d, v%d, %s" % (r, 40 + kk, qreg(agpr) + (kk + 3 * c) % 17, src_c)


def fold(src_set, c, r, nops, agpr=False, plain=False):
    """fold value r of column block c of accumulator set src_set; nops = VALU ops per value (2, 3 or 4)"""
    out = []
    if plain:
        src = "v%d" % (210 + r)
    elif agpr:
        out.append("v_accvgpr_read_b32 v36, a%d" % ((0 if src_set == "A" else 64) + 16 * c + r))
        src = "v36"
    else:
        src = "v%d" % (ACC[src_set] + 16 * c + r)
    if nops >= 3:
        out.append("v_and_or_b32 v37, %s, s20, %d" % (src, r))
        src = "v37"
    out.append("v_med3_f32 v%d, v%d, v%d, %s" % (24 + c, 20 + c, 24 + c, src))
    if nops >= 4:
        out.append("v_max_f32 v38, %s, %s" % (src, src))
        src = "v38"
    out.append("v_min_f32 v%d, v%d, %s" % (20 + c, 20 + c, src))
    return out


def tile_tail(c):
    return ["v_cmp_neq_f32 vcc, v%d, v%d" % (20 + c, 28 + c), "v_cndmask_b32 v%d, v%d, v39, vcc" % (32 + c, 32 + c),
            "v_mov_b32 v%d, v%d" % (28 + c, 20 + c)]


def body_sequential(nops, agpr=False):
    """the product kernel's order: 68 MFMAs (four chains round-robin), wait, then the whole fold"""
    L = []
    for kk in range(17):
        for c in range(4):
            L.append(mfma("A", c, kk, agpr, first=(kk == 0)))
    L += ["s_nop 15", "s_nop 3"]
    for c in range(4):
        for r in range(16):
            L += fold("A", c, r, nops, agpr)
        L += tile_tail(c)
    return L, 1


def body_interleaved(nops, agpr=False, plain=False, per_slot=1):
    """two accumulator sets: the MFMAs of this tile go to one set while the other set (previous tile) is folded, one value
    after every MFMA starting behind the second one.  Two tiles per loop trip (A<-mfma/B folded, then B<-mfma/A folded)."""
    L = []
    for dst, src in (("A", "B"), ("B", "A")):
        vals = [(c, r) for c in range(4) for r in range(16)]
        vi = 0
        n = 0
        for kk in range(17):
            for c in range(4):
                L.append(mfma(dst, c, kk, agpr, first=(kk == 0)))
                n += 1
                if n >= 2:
                    for _ in range(per_slot):
                        if vi < len(vals):
                            cc, rr = vals[vi]
                            L += fold(src, cc, rr, nops, agpr, plain)
                            if rr == 15:
                                L += tile_tail(cc)
                            vi += 1
        while vi < len(vals):
            cc, rr = vals[vi]
            L += fold(src, cc, rr, nops, agpr, plain)
            if rr == 15:
                L += tile_tail(cc)
            vi += 1
    return L, 2


def body_mfma_only():
    L = []
    for kk in range(17):
        for c in range(4):
            L.append(mfma("A", c, kk, False, first=(kk == 0)))
    return L, 1


def body_fold_only(nops):
    L = []
    for c in range(4):
        for r in range(16):
            L += fold("A", c, r, nops)
        L += tile_tail(c)
    return L, 1


def mfma16(dst_set, c, j, first=False):
    base = ACC[dst_set] + 16 * c
    r = "v[%d:%d]" % (base, base + 15)
    a0, b0 = 192 + 4 * j, 220 + 4 * ((j + 2 * c) % 7)
    return "v_mfma_f32_32x32x16_f16 %s, v[%d:%d], v[%d:%d], %s" % (r, a0, a0 + 3, b0, b0 + 3, "0" if first else r)


def body16_mfma_only():
    return [mfma16("A", c, j, first=(j == 0)) for j in range(7) for c in range(4)], 1


def body16_sequential(nops):
    L = [mfma16("A", c, j, first=(j == 0)) for j in range(7) for c in range(4)]
    L += ["s_nop 15"]
    for c in range(4):
        for r in range(16):
            L += fold("A", c, r, nops)
        L += tile_tail(c)
    return L, 1


def body16_interleaved(nops):
    L = []
    for dst, src in (("A", "B"), ("B", "A")):
        vals = [(c, r) for c in range(4) for r in range(16)]
        vi = 0
        n = 0
        for j in range(7):
            for c in range(4):
                L.append(mfma16(dst, c, j, first=(j == 0)))
                n += 1
                if n >= 2:
                    for _ in range(3 if n < 27 else 0):
                        if vi < len(vals):
                            cc, rr = vals[vi]
                            L += fold(src, cc, rr, nops)
                            if rr == 15:
                                L += tile_tail(cc)
                            vi += 1
        while vi < len(vals):
            cc, rr = vals[vi]
            L += fold(src, cc, rr, nops)
            if rr == 15:
                L += tile_tail(cc)
            vi += 1
    return L, 2


VARIANTS = [
    ("f16 x16: 28 mfma only (28 x 32 = 896)", body16_mfma_only()),
    ("f16 x16: seq 3op", body16_sequential(3)),
    ("f16 x16: seq 2op", body16_sequential(2)),
    ("f16 x16: interleaved 3op", body16_interleaved(3)),
    ("f16 x16: interleaved 2op", body16_interleaved(2)),
    ("mfma_only", body_mfma_only()),
    ("fold_only_3op", body_fold_only(3)),
    ("fold_only_4op", body_fold_only(4)),
    ("seq_4op (product order)", body_sequential(4)),
    ("seq_3op", body_sequential(3)),
    ("seq_2op", body_sequential(2)),
    ("interleaved_3op vgpr acc", body_interleaved(3)),
    ("interleaved_4op vgpr acc", body_interleaved(4)),
    ("interleaved_2op vgpr acc", body_interleaved(2)),
    ("interleaved_3op agpr acc (+accvgpr_read)", body_interleaved(3, agpr=True), True),
    ("interleaved_3op plain regs (no acc reads)", body_interleaved(3, plain=True)),
    ("seq_3op agpr acc", body_sequential(3, agpr=True), True),
]

VARIANTS = [(v[0], v[1], len(v) > 2) for v in VARIANTS]


def clobber(agpr):
    if agpr:
        return ", ".join('"v%d"' % i for i in range(20, 84)) + ", " + ", ".join('"a%d"' % i for i in range(0, 128))
    return ", ".join('"v%d"' % i for i in range(20, 248))


print("// generated by gen_probe3.py — do not edit")
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n")
for i, (name, (lines, tiles), agpr) in enumerate(VARIANTS):
    asm = ["s_mov_b32 s20, 0xfffffff0", "v_mov_b32 v39, 7"]
    for r in list(range(20, 39)) + list(range(40, 57)) + list(range(64, 84 if agpr else 248)):
        asm.append("v_mov_b32 v%d, 1.0" % r)
    if agpr:
        for r in range(0, 128):
            asm.append("v_accvgpr_write_b32 a%d, v40" % r)
    asm += ["s_memtime %[c0]", "s_waitcnt lgkmcnt(0)", "L_loop_%d:" % i]
    asm += lines
    asm += ["s_sub_u32 %[it], %[it], 1", "s_cmp_lg_u32 %[it], 0", "s_cbranch_scc1 L_loop_%d" % i, "s_nop 15", "s_nop 15",
            "s_memtime %[c1]", "s_waitcnt lgkmcnt(0)"]
    # keep results alive
    asm += ["v_add_f32 %[r], v20, v24", "v_add_f32 %[r], %[r], v64", "v_add_f32 %[r], %[r], v32"]
    asm += ["v_accvgpr_read_b32 v36, a0", "v_add_f32 %[r], %[r], v36"] if agpr else ["v_add_f32 %[r], %[r], v128"]
    print("__global__ __launch_bounds__(256) void k_v%d(long long* clk, float* out, int iters) {" % i)
    print("  long long c0, c1;\n  float r;\n  int it = iters;")
    print("  asm volatile(")
    for a in asm:
        print('      "%s\\n"' % a)
    print('      : [c0] "=&s"(c0), [c1] "=&s"(c1), [r] "=&v"(r), [it] "+s"(it)\n      :\n      : %s, "s20", "vcc", "scc", "memory");' % clobber(agpr))
    print("  out[blockIdx.x * blockDim.x + threadIdx.x] = r;")
    print("  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;\n}\n")

print(r"""
// ---- numerics of v_mfma_f32_32x32x16_f16: how far is the f32 accumulation of the 16 (x7) exact products from the exact sum?
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k_num(const _Float16* A, const _Float16* B, float* C, int nk) {  // A[32][16*nk], B[32][16*nk] (row = i resp. j)
  const int lane = threadIdx.x, i = lane & 31, g = lane >> 5;
  f32x16 acc = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int m = 0; m < nk; ++m) {
    h8 a, b;
    for (int e = 0; e < 8; ++e) {
      a[e] = A[(size_t)i * 16 * nk + 16 * m + 8 * g + e];
      b[e] = B[(size_t)i * 16 * nk + 16 * m + 8 * g + e];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) C[(size_t)(8 * (r >> 2) + 4 * g + (r & 3)) * 32 + i] = acc[r];  // C[row of A][row of B]
}
#include <cmath>
#include <random>
#include <vector>
static void numerics() {
  for (int mode = 0; mode < 3; ++mode)
    for (int nk = 1; nk <= 7; nk += 6) {
      const int K = 16 * nk;
      std::vector<_Float16> A(32 * K), B(32 * K);
      std::mt19937 rng(7 + mode);
      std::uniform_real_distribution<float> U(0.f, 1.f);
      for (int t = 0; t < 32 * K; ++t) {
        float a, b;
        if (mode == 0) { a = U(rng) * 100.f; b = -U(rng) * 100.f; }                     // same sign, like -2ab of histograms
        else if (mode == 1) { a = (U(rng) - .5f) * 200.f; b = (U(rng) - .5f) * 200.f; }  // cancellation
        else { a = std::ldexp(U(rng) + 1.f, (int)(U(rng) * 24) - 12); b = std::ldexp(U(rng) + 1.f, (int)(U(rng) * 24) - 12); if (t & 1) b = -b; }  // wide exponents
        A[t] = (_Float16)a;
        B[t] = (_Float16)b;
      }
      _Float16 *dA, *dB;
      float* dC;
      hipMalloc(&dA, A.size() * 2);
      hipMalloc(&dB, B.size() * 2);
      hipMalloc(&dC, 32 * 32 * 4);
      hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_num, dim3(1), dim3(64), 0, 0, dA, dB, dC, nk);
      std::vector<float> C(32 * 32);
      hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
      double worst = 0, worst_rel = 0;
      int exact_rn = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double s = 0, sa = 0;
          for (int k = 0; k < K; ++k) {
            const double p = (double)(float)A[i * K + k] * (double)(float)B[j * K + k];
            s += p;
            sa += std::fabs(p);
          }
          const double err = std::fabs((double)C[i * 32 + j] - s);
          worst = std::fmax(worst, err / (sa * 5.9604645e-08));
          if (s != 0) worst_rel = std::fmax(worst_rel, err / (std::fabs(s) * 5.9604645e-08));
          exact_rn += ((float)s == C[i * 32 + j]);
        }
      printf("f16 mfma numerics mode %d K %3d: max |err| = %.3f u*sum|terms|, %.3f u*|sum|; %d/1024 equal to RN(exact)\n", mode, K, worst,
             worst_rel, exact_rn);
    }
}
""")
print("template <typename K>\nstatic void run(const char* name, K kern, int tiles_per_trip, int blocks) {")
print("""  long long* clk;
  float* out;
  hipMalloc(&clk, 16);
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, clk, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  long long h = 0;
  hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  const double tiles = (double)iters * tiles_per_trip;
  printf("%-46s blocks %4d: clk/tile %7.0f   wall us/tile/wg %.3f   (68 x 64 = 4352)\\n", name, blocks, (double)h / tiles,
         ms * 1e3 / tiles / ((blocks + 255) / 256));
  hipFree(clk);
  hipFree(out);
}
int main() {""")
for i, (name, (lines, tiles), agpr) in enumerate(VARIANTS):
    print('  run("%s", k_v%d, %d, 256);' % (name, i, tiles))
for i, (name, (lines, tiles), agpr) in enumerate(VARIANTS):
    if i not in (0, 5, 6, 7):
        print('  run("%s", k_v%d, %d, 512);' % (name, i, tiles))
print("  numerics();\n  return 0;\n}")

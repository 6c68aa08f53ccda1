#!/usr/bin/env python3
"""gen_nn_f16_core_packed.py — writes nn_f16_core_packed.inc: the inner loop of k_nn_f16 as rounds 2-4 ran it (the row index of
a value packed into its low mantissa bits: three instructions per value), kept for the BATCHED launches — there the
finish of the index-free loop (gen_nn_f16_core.py: k_nn_finish_f16 stages 16 candidate rows per query in LDS, one workgroup
per compute unit) costs a group of sixteen pairs more than the loop saves.     python gen_nn_f16_core_packed.py > nn_f16_core_packed.inc

Why by hand: the loop needs 28 x v_mfma_f32_32x32x16_f16 per 32-row tile with the 3-VALU-per-value top-2 fold of the
PREVIOUS tile issued in their shadow, two accumulator sets in architectural VGPRs (the fold must read them without
v_accvgpr_read) and the 112 dwords of stationary query fragments in AGPRs (MFMA reads them in place).  The compiler's
register allocation put the accumulators in AGPRs and re-packed the f16 fragments with v_perm; the schedule is the whole
point of the kernel, so it is written out.  Timing of this schedule in isolation: tests/probe/gen_probe3.py.

Register map (per lane)
  a[0:111]    query fragments: q[c][m] = a[(7c+m)*4 .. +3]      (c = column block 0..3, m = MFMA 0..6)
  v[64:127]   accumulator set A (16 per column block), v[128:191] set B
  v[192:219]  base tile buffer 0 (7 fragments of 4 dwords), v[220:247] buffer 1
  v20-23 best, v24-27 second, v28-31 best before the tile, v32-35 tile of the best, v36 temp, v38 tile being folded,
  v39 lane byte offset inside a chunk pair, v40 = v39 + 4096
  s[40:41] base-table cursor (next tile to load), s[44:45] query table, s42 tiles left to start, s43 pack mask
Wait states that the assembler will not insert for us (gfx940/950): a VALU read of an MFMA result needs the MFMA to
be 11 wait states old (8-pass) — every fold starts behind two MFMAs of the next tile; v_cmp -> v_cndmask through VCC
needs 2 (s_nop 1).
"""

TILE_BYTES = 14 * 32 * 16  # 7168: one tile of either operand table
ACC = {"A": 64, "B": 128}
MBUF = {0: 192, 1: 220}


def vr(lo, n):
    return "v[%d:%d]" % (lo, lo + n - 1)


def load_tile(buf):
    """7 fragment loads of the tile at the base cursor into buffer buf, then advance the cursor"""
    L = []
    for j in range(7):
        va, off = ("v39", 1024 * j) if j < 4 else ("v40", 1024 * (j - 4))
        L.append("global_load_dwordx4 %s, %s, s[40:41] offset:%d" % (vr(MBUF[buf] + 4 * j, 4), va, off))
    L += ["s_add_u32 s40, s40, %d" % TILE_BYTES, "s_addc_u32 s41, s41, 0"]
    return L


def mfma(dst, c, j, buf):
    acc = vr(ACC[dst] + 16 * c, 16)
    q = "a[%d:%d]" % ((7 * c + j) * 4, (7 * c + j) * 4 + 3)
    return "v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (acc, vr(MBUF[buf] + 4 * j, 4), q, "0" if j == 0 else acc)


def fold_value(src, c, r):
    a = ACC[src] + 16 * c + r
    return ["v_and_or_b32 v36, v%d, s43, %d" % (a, r), "v_med3_f32 v%d, v%d, v%d, v36" % (24 + c, 20 + c, 24 + c),
            "v_min_f32 v%d, v%d, v36" % (20 + c, 20 + c)]


def fold_tail(c):
    return ["v_cmp_neq_f32 vcc, v%d, v%d" % (20 + c, 28 + c), "s_nop 1", "v_cndmask_b32 v%d, v%d, v38, vcc" % (32 + c, 32 + c),
            "v_mov_b32 v%d, v%d" % (28 + c, 20 + c)]


def fold_groups(src):
    """the fold of one accumulator set as a list of small instruction groups (one value each, tails after r = 15)"""
    G = []
    for c in range(4):
        for r in range(16):
            g = fold_value(src, c, r)
            if r == 15:
                g += fold_tail(c)
            G.append(g)
    G.append(["v_add_u32 v38, 1, v38"])
    return G


def phase(dst, buf, fold_src):
    """28 MFMAs of the tile in buffer buf into set dst; the fold of set fold_src (or None) dealt between them, starting
    behind the second MFMA"""
    L = []
    G = fold_groups(fold_src) if fold_src else []
    gi = 0
    n = 0
    for j in range(7):
        for c in range(4):
            L.append(mfma(dst, c, j, buf))
            n += 1
            if n >= 2:
                take = 3 if n % 2 == 0 else 2  # 13 x 3 + 13 x 2 = the 65 groups, spread over MFMAs 2..27
                for _ in range(take):
                    if gi < len(G):
                        L += G[gi]
                        gi += 1
    while gi < len(G):
        L += G[gi]
        gi += 1
    return L


def fold_only(src):
    L = ["s_nop 15"]
    for g in fold_groups(src):
        L += g
    return L


asm = []
A = asm.append
# ---- prologue
asm += ["s_mov_b32 s40, %[blo]", "s_mov_b32 s41, %[bhi]", "s_mov_b32 s44, %[qlo]", "s_mov_b32 s45, %[qhi]", "s_mov_b32 s42, %[nt]",
        "s_mov_b32 s43, 0xfffffff0", "v_mov_b32 v39, %[frag]", "v_add_u32 v40, 0x1000, v39", "v_mov_b32 v38, %[t0]"]
for c in range(4):  # query fragments straight into AGPRs; the lane's row of column block c starts at byte %[qc] of the table
    asm.append("v_add_u32 v36, 0x1000, %%[q%d]" % c)
    for m in range(7):
        va, off = ("%%[q%d]" % c, 1024 * m) if m < 4 else ("v36", 1024 * (m - 4))
        asm.append("global_load_dwordx4 a[%d:%d], %s, s[44:45] offset:%d" % ((7 * c + m) * 4, (7 * c + m) * 4 + 3, va, off))
for c in range(4):
    asm += ["v_mov_b32 v%d, 0x7f800000" % (20 + c), "v_mov_b32 v%d, 0x7f800000" % (24 + c), "v_mov_b32 v%d, 0x7f800000" % (28 + c),
            "v_mov_b32 v%d, -1" % (32 + c)]
asm += load_tile(0)  # tile 0 of the slice
asm += load_tile(1)  # tile 1 (the tables are padded by two tiles: prefetching past the slice is harmless)
asm += ["s_waitcnt vmcnt(7)"]
asm += phase("A", 0, None)
asm += ["s_sub_u32 s42, s42, 1", "s_cmp_eq_u32 s42, 0", "s_cbranch_scc1 L_f16p_tailA_%="]
A("L_f16p_loop_%=:")
# A holds an unfolded tile, buffer 1 holds (or is receiving) the next one
asm += load_tile(0)
asm += ["s_waitcnt vmcnt(7)"]
asm += phase("B", 1, "A")
asm += ["s_sub_u32 s42, s42, 1", "s_cmp_eq_u32 s42, 0", "s_cbranch_scc1 L_f16p_tailB_%="]
asm += load_tile(1)
asm += ["s_waitcnt vmcnt(7)"]
asm += phase("A", 0, "B")
asm += ["s_sub_u32 s42, s42, 1", "s_cmp_eq_u32 s42, 0", "s_cbranch_scc0 L_f16p_loop_%="]
A("L_f16p_tailA_%=:")
asm += fold_only("A")
A("s_branch L_f16p_done_%=")
A("L_f16p_tailB_%=:")
asm += fold_only("B")
A("L_f16p_done_%=:")
asm += ["s_waitcnt vmcnt(0)"]
for c in range(4):
    asm += ["v_mov_b32 %%[b1%d], v%d" % (c, 20 + c), "v_mov_b32 %%[b2%d], v%d" % (c, 24 + c), "v_mov_b32 %%[it%d], v%d" % (c, 32 + c)]

clob = ['"v%d"' % i for i in list(range(20, 41)) + list(range(64, 248))] + ['"a%d"' % i for i in range(112)]
clob += ['"s40"', '"s41"', '"s42"', '"s43"', '"s44"', '"s45"', '"vcc"', '"scc"', '"memory"']

print("// generated by gen_nn_f16_core_packed.py — do not edit (see that file for the register map and the schedule)")
print("// One item of k_nn_f16: 4 x 32 query columns of this wave (the lane's row of column block c starts at byte qoff[c] of")
print("// the query table: any row, so a list of rows needs no gathered copy) against `ntiles` base tiles starting at `base`;")
print("// running best / second best (scaled, row index packed in the low mantissa bits) and the tile of the best, per block.")
print("__device__ __forceinline__ void nn_f16_core_packed(const uint4* query, const u32 (&qoff)[4], const uint4* base, int ntiles, int t_begin,")
print("                                            u32 frag_bytes,")
print("                                            float (&b1)[4], float (&b2)[4], int (&it1)[4]) {")
print("  const u32 qlo = __builtin_amdgcn_readfirstlane((u32)(uintptr_t)query), qhi = __builtin_amdgcn_readfirstlane((u32)((uintptr_t)query >> 32));")
print("  const u32 blo = __builtin_amdgcn_readfirstlane((u32)(uintptr_t)base), bhi = __builtin_amdgcn_readfirstlane((u32)((uintptr_t)base >> 32));")
print("  const int nt = __builtin_amdgcn_readfirstlane(ntiles), t0 = __builtin_amdgcn_readfirstlane(t_begin);")
print("  asm volatile(")
for a in asm:
    print('      "%s\\n"' % a)
outs = ", ".join('[b1%d] "=&v"(b1[%d]), [b2%d] "=&v"(b2[%d]), [it%d] "=&v"(it1[%d])' % (c, c, c, c, c, c) for c in range(4))
print("      : %s" % outs)
print('      : [qlo] "s"(qlo), [qhi] "s"(qhi), [blo] "s"(blo), [bhi] "s"(bhi), [nt] "s"(nt), [t0] "s"(t0), [frag] "v"(frag_bytes),')
print('        [q0] "v"(qoff[0]), [q1] "v"(qoff[1]), [q2] "v"(qoff[2]), [q3] "v"(qoff[3])')
print("      : %s);" % ", ".join(clob))
print("}")

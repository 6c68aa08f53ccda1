#!/usr/bin/env python3
"""gen_nn_f16_core.py — writes nn_f16_core.inc: the hand-scheduled inner loop of k_nn_f16 (match.hip) as one inline-asm
block with a fixed register map.       python gen_nn_f16_core.py > nn_f16_core.inc

Why by hand: the loop needs 28 x v_mfma_f32_32x32x16_f16 per 32-row tile with the top-2 fold of the PREVIOUS tile issued
in their shadow, two accumulator sets in architectural VGPRs (the fold must read them without v_accvgpr_read) and the
112 dwords of stationary query fragments in AGPRs (MFMA reads them in place).  The compiler's register allocation put
the accumulators in AGPRs and re-packed the f16 fragments with v_perm; the schedule is the whole point of the kernel, so
it is written out.  Timing of the round-2 schedule in isolation: tests/probe/gen_probe3.py.

The fold (round 5).  One wave per SIMD hides about five single-issue instructions behind a v_mfma_f32_32x32x16 (32
clocks of matrix pipe = 8 issue slots; MI355X_MICROARCH.md), i.e. ~140 per tile of 28 MFMAs; rounds 2-4 folded every
value with three instructions (row index packed into the low mantissa bits, v_med3, v_min: 209 per tile with the
per-tile bookkeeping) and ran at 1186 clocks per tile against 896 of matrix pipe.  Now the lane's running (best,
second best) takes the 16 values of a column block two at a time:
    t  = med3(best, x, y)        the second smallest of the three
    best = min3(best, x, y)
    second = min3(second, t, t') once per two pairs
— five instructions per FOUR values, 89 per tile with the bookkeeping: the loop is bound by the matrix pipe.  (Invariant:
second >= best; the two smallest of {best, second, x, y} are min3(best, x, y) and min(second, med3(best, x, y)).)  No
index travels with a value any more: the lane notes the tile and QUAD of accumulator registers (four consecutive rows
of the base cloud) in which its best last changed — the running best moves through five registers per column block
(in, three temporaries, out; in / out ping-pong between two banks from tile to tile), one compare and one conditional
move per quad, 116 instructions per tile all told (131 issue slots with loads and loop control: still inside the ~140 the
matrix pipe hides) — so a partial result names 4 candidate rows, one aligned 528-byte run of the row-major descriptor
table, and k_nn_finish_f16 picks the one with the smallest EXACT distance (match.hip).

Register map (per lane)
  a[0:111]    query fragments: q[c][m] = a[(7c+m)*4 .. +3]      (c = column block 0..3, m = MFMA 0..6)
  v[64:127]   accumulator set A (16 per column block), v[128:191] set B
  v[192:219]  base tile buffer 0 (7 fragments of 4 dwords), v[220:247] buffer 1
  v20-23 / v28-31 best (ping-pong: the fold of set A reads 20.. and writes 28.., the fold of set B the other way),
  v24-27 second, v32-35 4 * tile + quad of the best, v36 v37 temporaries, v41-43 the best between quads,
  v44-47 4 * (tile being folded) + quad,
  v39 lane byte offset inside a chunk pair, v40 = v39 + 4096
  s[40:41] base-table cursor (next tile to load), s[44:45] query table, s42 tiles left to start, s[46:61] compare masks
Wait states that the assembler will not insert for us (gfx940/950): a VALU read of an MFMA result needs the MFMA to
be 11 wait states old (8-pass) — every fold starts behind two MFMAs of the next tile; a v_cmp's mask is read by its
v_cndmask three instructions later.
"""

import sys

LDS = "--lds" in sys.argv  # experiment (round 6, VERDICT item 4 ii): the base tile staged ONCE per workgroup through LDS —
# every wave fetches two of the tile's eight 1 KB pieces with global_load_lds_dwordx4 (the eighth is the next tile's first:
# it lands in a spare slot), a barrier, then seven ds_read_b128 per wave; three tiles in flight (fabric -> LDS two ahead,
# LDS -> registers one ahead).  Measured against the register-direct loop in profiles/r6_ab.txt; nn_f16_core_lds.inc is what
# this flag writes, built into the library only with -DQTR_NN_LDS_STAGE.
TILE_BYTES = 14 * 32 * 16  # 7168: one tile of either operand table
LBUF_BYTES = 8192          # an LDS buffer: the tile + the spare piece
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


def dma_tile(lbuf):
    """this wave's two pieces of the tile at the base cursor -> LDS buffer lbuf, then advance the cursor
    (s36 / s37: LDS addresses of the wave's pieces in buffer 0, s38 / s39 in buffer 1; v48 / v49 their byte offsets)"""
    a, b = (36, 37) if lbuf == 0 else (38, 39)
    return ["s_mov_b32 m0, s%d" % a, "global_load_lds_dwordx4 v48, s[40:41]", "s_mov_b32 m0, s%d" % b,
            "global_load_lds_dwordx4 v49, s[40:41]", "s_add_u32 s40, s40, %d" % TILE_BYTES, "s_addc_u32 s41, s41, 0"]


def lds_to_regs(lbuf, buf):
    """seven fragments of the tile in LDS buffer lbuf -> register buffer buf (v50 / v51: the lane's address in buffer 0 / 1)"""
    va = "v50" if lbuf == 0 else "v51"
    return ["ds_read_b128 %s, %s offset:%d" % (vr(MBUF[buf] + 4 * j, 4), va, 1024 * j) for j in range(7)]


def top(p_next_buf):
    """top of a phase that computes from register buffer 1 - p_next_buf: the tile after it goes LDS -> registers, the one after
    that fabric -> LDS (into the LDS buffer the phase's own tile came from: every wave has read it, see the barrier)"""
    nb = p_next_buf
    return ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"] + lds_to_regs(nb, nb) + dma_tile(1 - nb)


def mfma(dst, c, j, buf):
    acc = vr(ACC[dst] + 16 * c, 16)
    q = "a[%d:%d]" % ((7 * c + j) * 4, (7 * c + j) * 4 + 3)
    return "v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (acc, vr(MBUF[buf] + 4 * j, 4), q, "0" if j == 0 else acc)


BEST = {"A": (20, 28), "B": (28, 20)}  # the fold of a set reads the running best from the first bank, writes the second


TMP = [41, 42, 43]   # the running best between the four quads of a column block (in -> T0 -> T1 -> T2 -> out)
TQ = [44, 45, 46, 47]  # 4 * tile + quad of the tile being folded


def fold_ops(src):
    """the fold of one accumulator set as a flat list of instructions (see the module docstring)"""
    pin, pout = BEST[src]
    ops = []
    pending = None  # the v_cndmask of the previous quad: issued a quad later than its v_cmp (the mask travels in SGPRs)
    n = 0
    for c in range(4):
        x = ["v%d" % (ACC[src] + 16 * c + r) for r in range(16)]
        sec = "v%d" % (24 + c)
        stage = ["v%d" % (pin + c)] + ["v%d" % t for t in TMP] + ["v%d" % (pout + c)]
        for q in range(4):
            r = 4 * q
            cur, out = stage[q], stage[q + 1]
            ops.append("v_med3_f32 v36, %s, %s, %s" % (cur, x[r], x[r + 1]))
            ops.append("v_min3_f32 %s, %s, %s, %s" % (out, cur, x[r], x[r + 1]))
            ops.append("v_med3_f32 v37, %s, %s, %s" % (out, x[r + 2], x[r + 3]))
            ops.append("v_min3_f32 %s, %s, %s, %s" % (out, out, x[r + 2], x[r + 3]))
            ops.append("v_min3_f32 %s, %s, v36, v37" % (sec, sec))
            sp = 46 + 2 * (n % 8)
            n += 1
            # did the best change in this QUAD (rows 8 q + 4 half + {0..3} of the tile)?
            ops.append("v_cmp_neq_f32_e64 s[%d:%d], %s, %s" % (sp, sp + 1, out, cur))
            if pending:
                ops.append(pending)
            pending = "v_cndmask_b32_e64 v%d, v%d, v%d, s[%d:%d]" % (32 + c, 32 + c, TQ[q], sp, sp + 1)
    ops += ["v_nop", "v_nop", pending]
    ops += ["v_add_u32 v%d, 4, v%d" % (t, t) for t in TQ]
    return ops


def phase(dst, buf, fold_src):
    """28 MFMAs of the tile in buffer buf into set dst; the fold of set fold_src (or None) dealt evenly between them,
    starting behind the second MFMA"""
    L = []
    G = fold_ops(fold_src) if fold_src else []
    gaps, gi, n = 26, 0, 0
    for j in range(7):
        for c in range(4):
            L.append(mfma(dst, c, j, buf))
            n += 1
            if n >= 2 and n - 2 < gaps:
                upto = (len(G) * (n - 1) + gaps - 1) // gaps
                L += G[gi:upto]
                gi = max(gi, upto)
    L += G[gi:]
    return L


def first_phase():
    """tile 0 into set A, column block by column block: a block's MFMAs start as soon as ITS query fragments have landed
    (the loads were issued q0, tile 0, q1, q2, q3, tile 1: 42 in flight), not when all 28 have"""
    L = []
    for c in range(4):
        # (LDS form: in flight behind q_c are the later query blocks and the two pieces of tile 2)
        L.append("s_waitcnt vmcnt(%d)" % ((23 - 7 * c) if LDS else (28 - 7 * c)))
        for j in range(7):
            L.append(mfma("A", c, j, 0))
    return L


def fold_only(src, move_best):
    L = ["s_nop 15"] + fold_ops(src)
    if move_best:  # the results are read from v20-23
        L += ["v_mov_b32 v%d, v%d" % (20 + c, 28 + c) for c in range(4)]
    return L


asm = []
A = asm.append


def query_loads(c):
    L = ["v_add_u32 v36, 0x1000, %%[q%d]" % c]
    for m in range(7):
        va, off = ("%%[q%d]" % c, 1024 * m) if m < 4 else ("v36", 1024 * (m - 4))
        L.append("global_load_dwordx4 a[%d:%d], %s, s[44:45] offset:%d" % ((7 * c + m) * 4, (7 * c + m) * 4 + 3, va, off))
    return L


# ---- prologue
asm += ["s_mov_b32 s40, %[blo]", "s_mov_b32 s41, %[bhi]", "s_mov_b32 s44, %[qlo]", "s_mov_b32 s45, %[qhi]", "s_mov_b32 s42, %[nt]",
        "v_mov_b32 v39, %[frag]", "v_add_u32 v40, 0x1000, v39", "v_lshlrev_b32 v44, 2, %[t0v]", "v_add_u32 v45, 1, v44",
        "v_add_u32 v46, 2, v44", "v_add_u32 v47, 3, v44"]
# query fragments straight into AGPRs; the lane's row of column block c starts at byte %[qc] of the table
if not LDS:
    asm += query_loads(0)
    asm += load_tile(0)  # tile 0 of the slice
    for c in range(1, 4):
        asm += query_loads(c)
    asm += load_tile(1)  # tile 1 (the tables are padded by two tiles: prefetching past the slice is harmless)
else:
    asm += ["s_mov_b32 s36, %[l0]", "s_add_u32 s37, s36, 4096", "s_add_u32 s38, s36, %d" % LBUF_BYTES, "s_add_u32 s39, s37, %d" % LBUF_BYTES,
            "v_mov_b32 v48, %[dma]", "v_add_u32 v49, 0x1000, v48", "v_mov_b32 v50, %[lrd]", "v_add_u32 v51, %d, v50" % LBUF_BYTES]
    asm += dma_tile(0) + dma_tile(1)  # tiles 0 and 1 on their way to LDS before anything else
    for c in range(4):
        asm += query_loads(c)
    asm += ["s_waitcnt vmcnt(28)", "s_barrier"]  # the four pieces this wave asked for have landed, and everybody's
    asm += lds_to_regs(0, 0) + lds_to_regs(1, 1)
    asm += ["s_waitcnt lgkmcnt(0)", "s_barrier"]  # every wave has tile 0 in registers: its LDS buffer is free
    asm += dma_tile(0)  # tile 2
for c in range(4):
    asm += ["v_mov_b32 v%d, 0x7f800000" % (20 + c), "v_mov_b32 v%d, 0x7f800000" % (24 + c), "v_mov_b32 v%d, 0x7f800000" % (28 + c),
            "v_mov_b32 v%d, -1" % (32 + c)]
asm += first_phase()
asm += ["s_sub_u32 s42, s42, 1", "s_cmp_eq_u32 s42, 0", "s_cbranch_scc1 L_f16_tailA_%="]
A("L_f16_loop_%=:")
# A holds an unfolded tile, buffer 1 holds (or is receiving) the next one
if not LDS:
    asm += load_tile(0)
    asm += ["s_waitcnt vmcnt(7)"]
else:
    asm += top(0)  # tile (p + 1) LDS buffer 0 -> register buffer 0, tile (p + 2) -> LDS buffer 1
asm += phase("B", 1, "A")
asm += ["s_sub_u32 s42, s42, 1", "s_cmp_eq_u32 s42, 0", "s_cbranch_scc1 L_f16_tailB_%="]
if not LDS:
    asm += load_tile(1)
    asm += ["s_waitcnt vmcnt(7)"]
else:
    asm += top(1)
asm += phase("A", 0, "B")
asm += ["s_sub_u32 s42, s42, 1", "s_cmp_eq_u32 s42, 0", "s_cbranch_scc0 L_f16_loop_%="]
A("L_f16_tailA_%=:")
asm += fold_only("A", True)
A("s_branch L_f16_done_%=")
A("L_f16_tailB_%=:")
asm += fold_only("B", False)
A("L_f16_done_%=:")
asm += ["s_waitcnt vmcnt(0)"]
for c in range(4):
    asm += ["v_mov_b32 %%[b1%d], v%d" % (c, 20 + c), "v_mov_b32 %%[b2%d], v%d" % (c, 24 + c), "v_mov_b32 %%[it%d], v%d" % (c, 32 + c)]

clob = ['"v%d"' % i for i in list(range(20, 48)) + list(range(64, 248))] + ['"a%d"' % i for i in range(112)]
clob += ['"s%d"' % i for i in [40, 41, 42, 44, 45] + list(range(46, 62))] + ['"vcc"', '"scc"', '"memory"']
if LDS:
    clob += ['"s%d"' % i for i in (36, 37, 38, 39)] + ['"v%d"' % i for i in (48, 49, 50, 51)] + ['"m0"']

print("// generated by gen_nn_f16_core.py — do not edit (see that file for the register map and the schedule)")
print("// One item of k_nn_f16: 4 x 32 query columns of this wave (the lane's row of column block c starts at byte qoff[c] of")
print("// the query table: any row, so a list of rows needs no gathered copy) against `ntiles` base tiles starting at `base`;")
print("// running best / second best (scaled) and the tile in which the best last changed, per column block.")
print("__device__ __forceinline__ void nn_f16_core%s(const uint4* query, const u32 (&qoff)[4], const uint4* base, int ntiles, int t_begin," % ("_lds" if LDS else ""))
print("                                            u32 frag_bytes,%s" % (" u32 lds_base, int wave," if LDS else ""))
print("                                            float (&b1)[4], float (&b2)[4], int (&it1)[4]) {")
print("  const u32 qlo = __builtin_amdgcn_readfirstlane((u32)(uintptr_t)query), qhi = __builtin_amdgcn_readfirstlane((u32)((uintptr_t)query >> 32));")
print("  const u32 blo = __builtin_amdgcn_readfirstlane((u32)(uintptr_t)base), bhi = __builtin_amdgcn_readfirstlane((u32)((uintptr_t)base >> 32));")
print("  const int nt = __builtin_amdgcn_readfirstlane(ntiles), t0 = __builtin_amdgcn_readfirstlane(t_begin);")
if LDS:
    print("  // (LDS addresses: two buffers of 8 KB at lds_base; this wave's pieces are wave and wave + 4 of a tile's eight KB)")
    print("  const u32 l0 = __builtin_amdgcn_readfirstlane(lds_base + 1024u * (u32)wave);")
    print("  const u32 dma = frag_bytes + 1024u * (u32)wave, lrd = lds_base + frag_bytes;")
print("  asm volatile(")
for a in asm:
    print('      "%s\\n"' % a)
outs = ", ".join('[b1%d] "=&v"(b1[%d]), [b2%d] "=&v"(b2[%d]), [it%d] "=&v"(it1[%d])' % (c, c, c, c, c, c) for c in range(4))
print("      : %s" % outs)
print('      : [qlo] "s"(qlo), [qhi] "s"(qhi), [blo] "s"(blo), [bhi] "s"(bhi), [nt] "s"(nt), [t0v] "v"(t0), [frag] "v"(frag_bytes),')
print('        [q0] "v"(qoff[0]), [q1] "v"(qoff[1]), [q2] "v"(qoff[2]), [q3] "v"(qoff[3])%s' % (', [l0] "s"(l0), [dma] "v"(dma), [lrd] "v"(lrd)' if LDS else ""))
print("      : %s);" % ", ".join(clob))
print("}")

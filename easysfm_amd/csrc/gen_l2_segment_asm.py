#!/usr/bin/env python3
"""Generates l2_segment_gfx950.inc: the hand-scheduled main loop of l2_knn_bf16_kernel (match_kernels.hip) as ONE inline-asm
block with fixed registers -- the distance GEMM of one SEGMENT (<= 16 tiles of 128 train rows = 64 steps of 32) for a wave's two
sets of 32 queries, with the fused two-level top-3 fold.

Why asm: hipcc would not keep a software-pipelined version of this loop in 256 VGPRs (33-69 spills, B operands reloaded from
scratch inside the loop, slower than the unpipelined loop), and its own placement of the LDS reads -- right in front of their
first use -- leaves every MFMA waiting for LDS (1.41 ms of main loop per 300 pairs against ~1.1 ms of matrix-pipe time).  Here
every instruction has its slot:

  K-step ks of a 32-train step (6 MFMAs: hi*lo, lo*hi, hi*hi for the two query sets)
      s_waitcnt lgkmcnt(n)                  operands of THIS K-step (issued two K-steps ago) have landed
      2 MFMA   n[s] = Ahi(ks) * Blo[s](ks) + (ks == 0 ? |t|^2 start values : n[s])
      6 VALU   fold of group ks of query set 0 of the PREVIOUS step (2 v_min3, v_and_or, 3 v_med3)
      2 MFMA   n[s] += Alo(ks) * Bhi[s](ks)
      6 VALU   fold of group ks of query set 1
      2 MFMA   n[s] += Ahi(ks) * Bhi[s](ks)
      2-6 ds_read_b128: the A fragments of K-step ks + 2 (this step's, or the next step's once ks >= 2; K-step 2 also fetches
               the next step's 16 start values) into the registers K-step ks + 2 of the step before last used

so the fragments of K-step ks always live in the same eight registers and the pipelining costs none.  A tile's hand-over sits
between K-steps 1 and 2 of its last step: by then the wave's last reads of the buffer have been issued (and are waited for), the
barrier tells the others -- and that tile + 1, whose LDS-DMA was issued a whole tile earlier, has landed -- and the DMA of tile + 2
goes into the buffer just released.

Register map (all clobbered):
  v[0:15] v[16:31]   accumulators "a" of query set 0 / 1      v[32:47] v[48:63]  accumulators "b"
  v[64:95]           A fragments: K-step ks hi = v[64+8ks : +3], lo = v[68+8ks : +3]
  v[96:111]          start values (|t|^2 of the step's 32 train rows, this lane's 16)
  v112-v119 a_addr   v120 n_addr   v121-v124 DMA source offsets   v125 norm source offset   v126 norm LDS address
  v127-v129, v135 scratch   v130-v133 fold temporaries   v134 key mask 0xFFFFFF00
  s40 tile  s41 code base of the step being folded  s42 step index  s43 saved M0  s44-s51 scratch  s52 total tiles  s53 1 if wave < 2
Operands: %0-%5 out: segment keys k0 k1 k2 of set 0, then set 1;  %6-%13 in: Bhi[s][ks] (s major);  %14-%21 in: Blo[s][ks];
  %22 tile0  %23 tile_end  %24 nt  %25 train image buffer descriptor  %26 train norms buffer descriptor  %27 LDS address of the
  tile buffers (norms behind them)  %28 wave index.
"""
import os
import sys

# timing-only experiment knobs (the shipped file is generated with the defaults): products per K-step, keys kept by the fold
PRODUCTS = int(os.environ.get("ESFM_GEN_PRODUCTS", "3"))
KEEP = int(os.environ.get("ESFM_GEN_KEEP", "3"))
DMA_SPREAD = int(os.environ.get("ESFM_GEN_DMA_SPREAD", "1"))   # 0: all eight pieces back to back after the barrier (measured 2 % slower)
KBIG = 0x7F61B1E6          # 3.0e38f
NKBIG = 0xFF61B1E6
TT, SLOT_BYTES = 128, 256  # rows per tile, bytes per row image
TILE_BYTES = TT * SLOT_BYTES

OP_K = lambda s, i: f"%{3 * s + i}"           # i: 0 k0, 1 k1, 2 k2
OP_BHI = lambda s, ks: f"%{6 + 4 * s + ks}"
OP_BLO = lambda s, ks: f"%{14 + 4 * s + ks}"
OP_TILE0, OP_TILE_END, OP_NT, OP_TRSRC, OP_NRSRC, OP_LDS, OP_WAVE = "%22", "%23", "%24", "%25", "%26", "%27", "%28"

ACC = {"a": (0, 16), "b": (32, 48)}


def vr(base, n=16):
    return f"v[{base}:{base + n - 1}]"


def fr(ks, lo):
    b = 64 + 8 * ks + (4 if lo else 0)
    return f"v[{b}:{b + 3}]"


class Emit:
    def __init__(self):
        self.lines = []

    def __call__(self, s):
        self.lines.append(s)


def gen():
    e = Emit()
    # ------------------------------------------------------------------ set-up
    e("s_mov_b32 s43, m0")
    e("v_mbcnt_lo_u32_b32 v128, -1, 0")
    e("v_mbcnt_hi_u32_b32 v128, -1, v128")                  # lane
    e("v_and_b32 v129, 31, v128")                            # j
    e("v_lshrrev_b32 v135, 5, v128")                         # h
    e("v_and_b32 v127, 15, v129")                            # j & 15
    e(f"v_lshlrev_b32 v126, 8, v129")                        # j * 256
    e(f"v_add_u32 v126, {OP_LDS}, v126")                     # row j of buffer 0
    for ks in range(4):
        for lo in (0, 1):
            e(f"v_or_b32 v130, {4 * ks + 2 * lo}, v135")      # logical slot 4 ks + 2 lo + h
            e("v_xor_b32 v130, v130, v127")                   # swizzle
            e(f"v_lshl_add_u32 v{112 + 2 * ks + lo}, v130, 4, v126")
    e("v_lshlrev_b32 v120, 4, v135")
    e(f"v_add_u32 v120, {OP_LDS}, v120")
    e(f"v_add_u32 v120, {2 * TILE_BYTES}, v120")              # n_addr = lds_norm + 16 h
    # LDS-DMA source offsets: wave w stages rows [32 w, 32 w + 32) of a tile, 4 rows per instruction
    e(f"s_lshl_b32 s44, {OP_WAVE}, 5")
    e("v_lshrrev_b32 v129, 4, v128")                          # lane >> 4
    e("v_and_b32 v127, 15, v128")                             # lane & 15
    for i in range(4):
        e(f"v_add_u32 v130, s44, v129")
        e(f"v_add_u32 v130, {4 * i}, v130")                   # row
        e("v_and_b32 v131, 15, v130")
        e("v_xor_b32 v131, v131, v127")
        e("v_lshlrev_b32 v131, 4, v131")
        e(f"v_lshl_add_u32 v{121 + i}, v130, 8, v131")
    # norms: threads 0..127 (waves 0, 1) move tile + 2's |t|^2 through a register into LDS
    e(f"s_lshl_b32 s44, {OP_WAVE}, 6")
    e("v_add_u32 v130, s44, v128")                            # tid
    e("v_lshlrev_b32 v125, 2, v130")                          # byte offset inside a tile's norms
    e(f"v_add_u32 v126, {OP_LDS}, v125")
    e(f"v_add_u32 v126, {2 * TILE_BYTES}, v126")              # lds_norm + 4 tid (buffer 0)
    e(f"s_cmp_lt_u32 {OP_WAVE}, 2")
    e("s_cselect_b32 s53, 1, 0")
    e(f"s_add_u32 s52, {OP_NT}, {TT - 1}")
    e("s_lshr_b32 s52, s52, 7")                               # total tiles
    e(f"s_mov_b32 s46, 0x{KBIG:08x}")
    e(f"s_mov_b32 s47, 0x{NKBIG:08x}")
    e("v_mov_b32 v134, 0xffffff00")
    for s in range(2):
        for i in range(3):
            e(f"v_mov_b32 {OP_K(s, i)}, s46")
    for r in range(32, 64):
        e(f"v_mov_b32 v{r}, s46")                             # placeholders for the step before the first: never live
    e(f"s_mov_b32 s40, {OP_TILE0}")
    e("s_mov_b32 s41, 0")
    e("s_mov_b32 s42, 0")
    # tiles tile0 and tile0 + 1 were issued by the caller (first segment) or by the previous segment's last two tiles
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier")
    # pipeline fill: K-steps 0 and 1 and the start values of step 0 (buffer 0: segments start on even tiles)
    # (start values first: the first K-step waits for everything but the last two reads)
    for g in range(4):
        e(f"ds_read_b128 v[{96 + 4 * g}:{99 + 4 * g}], v120 offset:{32 * g}")
    for ks in (0, 1):
        e(f"ds_read_b128 {fr(ks, 0)}, v{112 + 2 * ks}")
        e(f"ds_read_b128 {fr(ks, 1)}, v{113 + 2 * ks}")

    def fold(s, ks, p):
        b = p[s] + 4 * ks
        t, key = 130 + 2 * s, 131 + 2 * s
        extra = [f"v_med3_f32 v{136 + 4 * s + i}, v{136 + 4 * s + i}, {OP_K(s, 2)}, v{key}" for i in range(KEEP - 3)]   # timing only
        return [f"v_min3_f32 v{t}, v{b}, s46, v{b + 1}",
                f"v_min3_f32 v{t}, v{t}, v{b + 2}, v{b + 3}",
                f"v_and_or_b32 v{key}, v{t}, v134, s44"] + extra + [
                f"v_med3_f32 {OP_K(s, 2)}, {OP_K(s, 1)}, {OP_K(s, 2)}, v{key}",
                f"v_med3_f32 {OP_K(s, 1)}, {OP_K(s, 0)}, {OP_K(s, 1)}, v{key}",
                f"v_med3_f32 {OP_K(s, 0)}, {OP_K(s, 0)}, v{key}, s47"]

    def loads(ks, buf, base, nbuf, nbase):
        out = []
        if ks < 2:
            off = buf * TILE_BYTES + base * SLOT_BYTES
            k2 = ks + 2
            out.append(f"ds_read_b128 {fr(k2, 0)}, v{112 + 2 * k2} offset:{off}")
            out.append(f"ds_read_b128 {fr(k2, 1)}, v{113 + 2 * k2} offset:{off}")
        else:
            off = nbuf * TILE_BYTES + nbase * SLOT_BYTES
            k2 = ks - 2
            if ks == 2:        # the next step's start values first: they are needed by its very first MFMA
                for g in range(4):
                    out.append(f"ds_read_b128 v[{96 + 4 * g}:{99 + 4 * g}], v120 offset:{(nbuf * TT + nbase + 8 * g) * 4}")
            out.append(f"ds_read_b128 {fr(k2, 0)}, v{112 + 2 * k2} offset:{off}")
            out.append(f"ds_read_b128 {fr(k2, 1)}, v{113 + 2 * k2} offset:{off}")
        return out

    WAIT = {0: 2, 1: 2, 2: 2, 3: 6}   # reads issued by the K-step before this one may stay in flight

    def kstep(ks, accs, buf, base, nbuf, nbase, wait=True, extra=((), (), ())):
        n = ACC[accs]
        p = ACC["b" if accs == "a" else "a"]
        if wait:
            e(f"s_waitcnt lgkmcnt({WAIT[ks]})")
        e(f"s_add_u32 s44, s41, {ks}")
        c = [vr(96), vr(96)] if ks == 0 else [vr(n[0]), vr(n[1])]
        if PRODUCTS >= 3:
            for s in range(2):
                e(f"v_mfma_f32_32x32x16_bf16 {vr(n[s])}, {fr(ks, 0)}, {OP_BLO(s, ks)}, {c[s]}")
            c = [vr(n[0]), vr(n[1])]
        for x in extra[0]:
            e(x)
        if PRODUCTS == 1:          # one MFMA per set: each set's fold behind the other set's MFMA
            e(f"v_mfma_f32_32x32x16_bf16 {vr(n[0])}, {fr(ks, 0)}, {OP_BHI(0, ks)}, {c[0]}")
        for x in fold(0, ks, p):
            e(x)
        second, third = ((0, 1) if os.environ.get("ESFM_GEN_A_ORDER", "hl") == "hh" else (1, 0))   # which A half goes second
        if PRODUCTS >= 2:
            for s in range(2):
                e(f"v_mfma_f32_32x32x16_bf16 {vr(n[s])}, {fr(ks, second)}, {OP_BHI(s, ks)}, {c[s]}")
            c = [vr(n[0]), vr(n[1])]
        else:
            e(f"v_mfma_f32_32x32x16_bf16 {vr(n[1])}, {fr(ks, 0)}, {OP_BHI(1, ks)}, {c[1]}")
        for x in extra[1]:
            e(x)
        for x in fold(1, ks, p):
            e(x)
        if PRODUCTS >= 2:
            for s in range(2):
                e(f"v_mfma_f32_32x32x16_bf16 {vr(n[s])}, {fr(ks, third)}, {OP_BHI(s, ks)}, {vr(n[s])}")
        for x in extra[2]:
            e(x)
        for x in loads(ks, buf, base, nbuf, nbase):
            e(x)

    def end_step():
        e("s_lshl_b32 s41, s42, 2")
        e("s_add_u32 s42, s42, 1")

    def dma_piece(i, buf):
        dst = buf * TILE_BYTES + 4 * i * SLOT_BYTES           # + lds base + wave * 32 rows, in s49
        return [f"s_add_u32 s50, s49, {dst}",
                "s_mov_b32 m0, s50",
                f"s_add_u32 s51, s48, {16 * SLOT_BYTES if i >= 4 else 0}",
                f"buffer_load_dwordx4 v{121 + (i & 3)}, {OP_TRSRC}, s51 offen lds"]

    def tile(buf, tag):
        # tile + 2 exists?  (s45 = 1) -- its norms start their way now, its DMA goes out after the hand-over
        e("s_add_u32 s44, s40, 2")
        e("s_cmp_lt_u32 s44, s52")
        e("s_cselect_b32 s45, 1, 0")
        e("s_and_b32 s51, s45, s53")
        e("s_cmp_eq_u32 s51, 0")
        e(f"s_cbranch_scc1 L_nonorm_{tag}_%=")
        e("s_lshl_b32 s48, s44, 9")                            # (tile + 2) * 128 * 4
        e(f"buffer_load_dword v127, v125, {OP_NRSRC}, s48 offen")
        e(f"L_nonorm_{tag}_%=:")
        for st, accs in ((0, "a"), (1, "b"), (2, "a")):
            for ks in range(4):
                kstep(ks, accs, buf, 32 * st, buf, 32 * (st + 1))
            end_step()
        # step 3
        kstep(0, "b", buf, 96, buf ^ 1, 0)
        kstep(1, "b", buf, 96, buf ^ 1, 0)
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_barrier")
        e("s_cmp_eq_u32 s45, 0")
        e(f"s_cbranch_scc1 L_nodma_{tag}_%=")
        e("s_cmp_eq_u32 s53, 0")
        e(f"s_cbranch_scc1 L_nostore_{tag}_%=")
        # rows past nt carry kBig
        e("s_add_u32 s44, s40, 2")
        e("s_lshl_b32 s44, s44, 7")
        e("v_lshrrev_b32 v128, 2, v125")
        e("v_add_u32 v128, s44, v128")                         # train row of this thread's norm
        e(f"v_cmp_gt_u32 vcc, {OP_NT}, v128")
        e("v_mov_b32 v129, s46")
        e("v_cndmask_b32 v127, v129, v127, vcc")
        e(f"ds_write_b32 v126, v127 offset:{buf * TT * 4}")
        e(f"L_nostore_{tag}_%=:")
        e("s_add_u32 s44, s40, 2")
        e("s_lshl_b32 s48, s44, 15")                           # (tile + 2) * 128 rows * 256 B
        e(f"s_lshl_b32 s49, {OP_WAVE}, 13")                    # wave * 32 rows * 256 B
        e(f"s_add_u32 s49, s49, {OP_LDS}")
        if not DMA_SPREAD:
            for i in range(8):
                for x in dma_piece(i, buf):
                    e(x)
        e(f"L_nodma_{tag}_%=:")
        if DMA_SPREAD:
            # the eight pieces ride behind the MFMA pairs of the step's last two K-steps (2 + 1 + 1 each) instead of going out
            # back to back; a tile that does not exist reads zeros through the descriptor into a buffer nobody reads again, so
            # they are issued unconditionally
            e("s_add_u32 s44, s40, 2")
            e("s_lshl_b32 s48, s44, 15")
            e(f"s_lshl_b32 s49, {OP_WAVE}, 13")
            e(f"s_add_u32 s49, s49, {OP_LDS}")
            pcs = [dma_piece(i, buf) for i in range(8)]
            kstep(2, "b", buf, 96, buf ^ 1, 0, wait=False, extra=(pcs[0] + pcs[1], pcs[2], pcs[3]))
            kstep(3, "b", buf, 96, buf ^ 1, 0, wait=False, extra=(pcs[4] + pcs[5], pcs[6], pcs[7]))
        else:
            kstep(2, "b", buf, 96, buf ^ 1, 0, wait=False)
            kstep(3, "b", buf, 96, buf ^ 1, 0, wait=False)
        end_step()
        e("s_add_u32 s40, s40, 1")

    e("L_top_%=:")
    tile(0, "e")
    e(f"s_cmp_ge_u32 s40, {OP_TILE_END}")
    e("s_cbranch_scc1 L_done_%=")
    tile(1, "o")
    e(f"s_cmp_lt_u32 s40, {OP_TILE_END}")
    e("s_cbranch_scc1 L_top_%=")
    e("L_done_%=:")
    # the last step's results (accumulators "b") are folded here: every read of them is > 4 MFMA times old... no: the last
    # MFMAs were just issued -- wait for the matrix pipe before the VALU reads (16 passes)
    e("s_nop 15")
    e("s_nop 15")
    e("s_waitcnt lgkmcnt(0)")                                   # the prefetches past the segment's end land in dead registers
    for ks in range(4):
        e(f"s_add_u32 s44, s41, {ks}")
        for s in range(2):
            for x in fold(s, ks, ACC["b"]):
                e(x)
    e("s_mov_b32 m0, s43")
    return e.lines


def main():
    lines = gen()
    clob = [f"v{i}" for i in range(136 if KEEP == 3 else 144)] + [f"s{i}" for i in range(40, 54)] + ["scc", "vcc", "memory"]
    out = ["// GENERATED by gen_l2_segment_asm.py -- do not edit; see that file for the schedule and the register map",
           "#define ESFM_L2_SEGMENT_ASM \\"]
    for l in lines:
        out.append(f'    "{l}\\n" \\')
    out.append('    ""')
    out.append("#define ESFM_L2_SEGMENT_CLOBBERS " + ", ".join(f'"{c}"' for c in clob))
    open(sys.argv[1] if len(sys.argv) > 1 else "l2_segment_gfx950.inc", "w").write("\n".join(out) + "\n")
    n_mfma = sum("v_mfma" in l for l in lines)
    print(f"{len(lines)} instructions, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()

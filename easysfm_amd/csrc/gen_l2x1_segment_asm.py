#!/usr/bin/env python3
"""Generates l2x1_segment_gfx950.inc: the hand-scheduled main loop of l2_knn_bf16x1_kernel (match_kernels.hip), the ONE-product
bf16 distance pass -- q.t ~ bf16(q).bf16(t), 4 instead of 12 v_mfma_f32_32x32x16_bf16 per 32 x 32 x 64 tile -- as one inline-asm
block per SEGMENT (<= 16 tiles of 128 train rows = 64 steps of 32) for a wave's two sets of 32 queries, with the fused two-level
top-K fold (K = KEEP group keys per lane and set; the certificate of the one-product pass needs K > 3).

Differences from gen_l2_segment_asm.py (the three-product pass, which now only sees the queries this pass cannot certify):
  * the train image is bf16(t) only: 128-B rows, 16-B slot 2 ks + h holds a lane's A fragment of K-step ks; in LDS slot s of row r
    sits at physical slot s ^ ((r >> 1) & 7) -- every 16-lane group of a ds_read_b128 then covers the 64 banks exactly once;
  * a RING of four 16-KiB tile buffers instead of two 32-KiB ones (a tile is a third of the matrix-pipe time it used to be: the
    LDS-DMA of tile t + 4 is issued when tile t hands its buffer over and has three tiles' time to land); a tile's |t|^2 travel
    through one VGPR per ring slot (in/out operands: they live across segments) and are written to LDS -- kBig for rows past nt --
    at the hand-over that publishes the tile;
  * the A fragments are prefetched a whole 32-train step ahead (two register sets of four fragments): with 2 instead of 6 MFMAs
    per K-step the two-K-step distance of the three-product schedule is shorter than the LDS latency;
  * K keys per set: 3 + K VALU per group of four results (2 v_min3, v_and_or, K v_med3).

  K-step ks of a 32-train step (2 MFMAs: one per query set)
      s_waitcnt lgkmcnt(n)            the fragment of THIS K-step (read one step ago) has landed
      MFMA  n[0] (+)= A(ks) * B[0](ks)      (ks == 0: on top of the |t|^2 start values)
      3 + K VALU   fold of group ks of set 0 of the PREVIOUS step
      MFMA  n[1] (+)= A(ks) * B[1](ks)
      3 + K VALU   fold of group ks of set 1
      ds_read_b128  A(ks) of the next step (ks == 2: its 16 start values first)
  hand-over of tile t = the top of its last step: every read of the tile has been issued a step ago and is waited for, the wave's own
  transfers of tile t + 1 are waited for (vmcnt) and its norms written, barrier, then the norm load and the four DMA pieces of
  tile t + 4 go out (LDS-DMA destinations stay below 64 KiB: M0 carries the address).

Register map (all clobbered):
  v[0:15] v[16:31]   accumulators "a" of query set 0 / 1      v[32:47] v[48:63]  accumulators "b"
  v[64:79] v[80:95]  A fragments of even / odd steps: K-step ks = v[64 + 16 par + 4 ks : +3]
  v[96:111]          start values (|t|^2 of the step's 32 train rows, this lane's 16)
  v112-v115 a_addr   v120 n_addr   v121-v124 DMA source offsets   v125 norm source offset   v126 norm LDS address
  v127-v129, v135 scratch   v130-v133 fold temporaries   v134 key mask 0xFFFFFF00
  s40 tile  s41 code base of the step being folded  s42 step index  s43 saved M0  s44-s51 scratch
Operands: %0..%(2K-1) out: segment keys of set 0, then set 1 (ascending);  then in/out: the four norm registers (ring slot 0..3);
  then in: B[s][ks] (s major, 8 operands);  then tile0, tile_end, nt, train image buffer descriptor, train norms buffer
  descriptor, LDS address of the ring (norms behind it), wave index.
"""
import os
import sys

KEEP = int(os.environ.get("ESFM_GEN_KEEP", "4"))
NOFOLD = int(os.environ.get("ESFM_GEN_NOFOLD", "0"))      # timing experiments only: no fold / no MFMA
NOMFMA = int(os.environ.get("ESFM_GEN_NOMFMA", "0"))
KBIG = 0x7F61B1E6          # 3.0e38f
NKBIG = 0xFF61B1E6
TT, ROW_BYTES, RING = 128, 128, 4
TILE_BYTES = TT * ROW_BYTES          # 16 KiB
NORM_BASE = RING * TILE_BYTES        # the ring's norms sit behind the tiles: RING x TT floats

OP_K = lambda s, i: f"%{KEEP * s + i}"
OP_NR = lambda b: f"%{2 * KEEP + b}"
OP_B = lambda s, ks: f"%{2 * KEEP + 4 + 4 * s + ks}"
_o = 2 * KEEP + 12
OP_TILE0, OP_TILE_END, OP_NT, OP_TRSRC, OP_NRSRC, OP_LDS, OP_WAVE = (f"%{_o + i}" for i in range(7))

ACC = {"a": (0, 16), "b": (32, 48)}


def vr(base, n=16):
    return f"v[{base}:{base + n - 1}]"


def fr(ks, par):
    b = 64 + 16 * par + 4 * ks
    return f"v[{b}:{b + 3}]"


def gen():
    L = []
    e = L.append
    # ------------------------------------------------------------------ set-up
    e("s_mov_b32 s43, m0")
    e("v_mbcnt_lo_u32_b32 v128, -1, 0")
    e("v_mbcnt_hi_u32_b32 v128, -1, v128")                  # lane
    e("v_and_b32 v129, 31, v128")                            # j
    e("v_lshrrev_b32 v135, 5, v128")                         # h
    e("v_bfe_u32 v127, v129, 1, 3")                          # (j >> 1) & 7: the row's swizzle
    e("v_lshlrev_b32 v126, 7, v129")                         # j * 128
    e(f"v_add_u32 v126, {OP_LDS}, v126")                     # row j of buffer 0, step 0
    for ks in range(4):
        e(f"v_or_b32 v130, {2 * ks}, v135")                   # logical slot 2 ks + h
        e("v_xor_b32 v130, v130, v127")
        e(f"v_lshl_add_u32 v{112 + ks}, v130, 4, v126")
    e("v_lshlrev_b32 v120, 4, v135")
    e(f"v_add_u32 v120, {OP_LDS}, v120")
    e(f"v_add_u32 v120, {NORM_BASE}, v120")                   # n_addr = lds_norm + 16 h
    # LDS-DMA source offsets: wave w stages rows [32 w, 32 w + 32) of a tile, 8 rows (1 KiB) per instruction
    e(f"s_lshl_b32 s44, {OP_WAVE}, 5")
    e("v_lshrrev_b32 v129, 3, v128")                          # lane >> 3
    e("v_and_b32 v127, 7, v128")                              # lane & 7: physical slot
    for i in range(4):
        e("v_add_u32 v130, s44, v129")
        e(f"v_add_u32 v130, {8 * i}, v130")                   # row in the tile
        e("v_bfe_u32 v131, v130, 1, 3")
        e("v_xor_b32 v131, v131, v127")                       # logical slot fetched into this physical slot
        e("v_lshlrev_b32 v131, 4, v131")
        e(f"v_lshl_add_u32 v{121 + i}, v130, 7, v131")
    # norms: lane l of wave w moves |t|^2 of row 32 w + (l & 31) (both halves of the wave the same row: no masking needed)
    e("v_and_b32 v130, 31, v128")
    e("v_add_u32 v130, s44, v130")                            # row in the tile
    e("v_lshlrev_b32 v125, 2, v130")                          # byte offset inside a tile's norms
    e(f"v_add_u32 v126, {OP_LDS}, v125")
    e(f"v_add_u32 v126, {NORM_BASE}, v126")                   # lds_norm + 4 row (ring slot 0)
    e(f"s_mov_b32 s46, 0x{KBIG:08x}")
    e(f"s_mov_b32 s47, 0x{NKBIG:08x}")
    e("v_mov_b32 v134, 0xffffff00")
    for s in range(2):
        for i in range(KEEP):
            e(f"v_mov_b32 {OP_K(s, i)}, s46")
    for r in range(32, 64):
        e(f"v_mov_b32 v{r}, s46")                             # placeholders for the step before the first: never live
    e(f"s_mov_b32 s40, {OP_TILE0}")
    e("s_mov_b32 s41, 0")
    e("s_mov_b32 s42, 0")
    # tile0 has landed for every wave: the caller's barrier (first segment) or the previous segment's last hand-over.
    # pipeline fill: the start values and the four fragments of step 0 (buffer 0: segments start on multiples of the ring)
    for g in range(4):
        e(f"ds_read_b128 v[{96 + 4 * g}:{99 + 4 * g}], v120 offset:{32 * g}")
    for ks in range(4):
        e(f"ds_read_b128 {fr(ks, 0)}, v{112 + ks}")

    def fold(s, ks, p):
        if NOFOLD:
            return []
        b = p[s] + 4 * ks
        t, key = 130 + 2 * s, 131 + 2 * s
        out = [f"v_min3_f32 v{t}, v{b}, s46, v{b + 1}",
               f"v_min3_f32 v{t}, v{t}, v{b + 2}, v{b + 3}",
               f"v_and_or_b32 v{key}, v{t}, v134, s44"]
        for i in range(KEEP - 1, 0, -1):
            out.append(f"v_med3_f32 {OP_K(s, i)}, {OP_K(s, i - 1)}, {OP_K(s, i)}, v{key}")
        out.append(f"v_med3_f32 {OP_K(s, 0)}, {OP_K(s, 0)}, v{key}, s47")
        return out

    WAIT = {0: 2, 1: None, 2: 3, 3: 7}   # younger reads that may stay in flight (see the module docstring's queue order)

    def kstep(ks, accs, par, nbuf, nstep, first_wait=None, extra=((), ())):
        """K-step ks of a step accumulating into `accs` with the fragments of parity `par`; prefetches the next step's
        fragment ks (and at ks == 2 its start values) from buffer nbuf, step nstep."""
        n = ACC[accs]
        p = ACC["b" if accs == "a" else "a"]
        w = WAIT[ks] if first_wait is None else first_wait
        if w is not None:
            e(f"s_waitcnt lgkmcnt({w})")
        e(f"s_add_u32 s44, s41, {ks}")
        c = [vr(96), vr(96)] if ks == 0 else [vr(n[0]), vr(n[1])]
        for s in range(2):
            if not NOMFMA:
                e(f"v_mfma_f32_32x32x16_bf16 {vr(n[s])}, {fr(ks, par)}, {OP_B(s, ks)}, {c[s]}")
            for x in extra[s]:
                e(x)
            for x in fold(s, ks, p):
                e(x)
        if ks == 2:
            for g in range(4):
                e(f"ds_read_b128 v[{96 + 4 * g}:{99 + 4 * g}], v120 offset:{(nbuf * TT + nstep * 32 + 8 * g) * 4}")
        e(f"ds_read_b128 {fr(ks, par ^ 1)}, v{112 + ks} offset:{nbuf * TILE_BYTES + nstep * 32 * ROW_BYTES}")

    def end_step():
        e("s_lshl_b32 s41, s42, 2")
        e("s_add_u32 s42, s42, 1")

    def dma_piece(i, buf):
        return [f"s_add_u32 s50, s49, {buf * TILE_BYTES + i * 1024}",
                "s_mov_b32 m0, s50",
                f"buffer_load_dwordx4 v{121 + i}, {OP_TRSRC}, s48 offen lds"]

    def norm_piece(buf):
        return [f"buffer_load_dword {OP_NR(buf)}, v125, {OP_NRSRC}, s51 offen"]

    def tile(buf, tag):
        nb = (buf + 1) % RING
        for st, accs in ((0, "a"), (1, "b"), (2, "a")):
            for ks in range(4):
                kstep(ks, accs, st & 1, buf, st + 1)
            end_step()
        # ---- step 3: hand-over first
        e("s_waitcnt lgkmcnt(0)")                               # every read of this tile has landed
        e(f"s_waitcnt vmcnt({2 * 5})")                          # this wave's five transfers of tile + 1 have landed (tile + 2, + 3 may fly)
        # norms of tile + 1 -> LDS, rows past nt as kBig
        e("s_add_u32 s44, s40, 1")
        e("s_lshl_b32 s44, s44, 7")
        e("v_lshrrev_b32 v128, 2, v125")
        e("v_add_u32 v128, s44, v128")                         # train row of this lane's norm
        e(f"v_cmp_gt_u32 vcc, {OP_NT}, v128")
        e("v_mov_b32 v129, s46")
        e(f"v_cndmask_b32 v129, v129, {OP_NR(nb)}, vcc")
        e(f"ds_write_b32 v126, v129 offset:{nb * TT * 4}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        e("s_add_u32 s44, s40, 4")
        e("s_lshl_b32 s51, s44, 9")                             # (tile + 4) * 128 * 4
        e("s_lshl_b32 s48, s44, 14")                            # (tile + 4) * 128 rows * 128 B
        e(f"s_lshl_b32 s49, {OP_WAVE}, 12")                     # wave * 32 rows * 128 B
        e(f"s_add_u32 s49, s49, {OP_LDS}")
        pcs = [dma_piece(i, buf) for i in range(4)]
        npc = norm_piece(buf)
        # (a tile that does not exist reads zeros through the descriptors into a buffer nobody reads again: unconditional)
        kstep(0, "b", 1, nb, 0, extra=(npc + pcs[0], pcs[1]))
        kstep(1, "b", 1, nb, 0, extra=(pcs[2], pcs[3]))
        kstep(2, "b", 1, nb, 0)
        kstep(3, "b", 1, nb, 0)
        end_step()
        e("s_add_u32 s40, s40, 1")

    e("L_top_%=:")
    for buf in range(RING):
        tile(buf, f"t{buf}")
        if buf < RING - 1:
            e(f"s_cmp_ge_u32 s40, {OP_TILE_END}")
            e("s_cbranch_scc1 L_done_%=")
    e(f"s_cmp_lt_u32 s40, {OP_TILE_END}")
    e("s_cbranch_scc1 L_top_%=")
    e("L_done_%=:")
    # the last step's results (accumulators "b"): wait for the matrix pipe before the VALU reads them
    e("s_nop 15")
    e("s_nop 15")
    e("s_waitcnt lgkmcnt(0)")                                   # the prefetches past the segment's end land in dead registers
    for ks in range(4):
        e(f"s_add_u32 s44, s41, {ks}")
        for s in range(2):
            for x in fold(s, ks, ACC["b"]):
                e(x)
    e("s_mov_b32 m0, s43")
    return L


def main():
    lines = gen()
    clob = [f"v{i}" for i in range(136)] + [f"s{i}" for i in range(40, 52)] + ["scc", "vcc", "memory"]
    out = ["// GENERATED by gen_l2x1_segment_asm.py -- do not edit; see that file for the schedule and the register map",
           f"#define ESFM_L2X1_KEEP {KEEP}",
           "#define ESFM_L2X1_SEGMENT_ASM \\"]
    for l in lines:
        out.append(f'    "{l}\\n" \\')
    out.append('    ""')
    out.append("#define ESFM_L2X1_SEGMENT_CLOBBERS " + ", ".join(f'"{c}"' for c in clob))
    open(sys.argv[1] if len(sys.argv) > 1 else "l2x1_segment_gfx950.inc", "w").write("\n".join(out) + "\n")
    n_mfma = sum("v_mfma" in l for l in lines)
    print(f"{len(lines)} instructions, {n_mfma} MFMAs, KEEP = {KEEP}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generates l2x1_segment_gfx950.inc: the hand-scheduled main loop of l2_knn_bf16x1_kernel (match_kernels.hip), the ONE-product
bf16 distance pass -- q.t ~ bf16(q).bf16(t), 4 instead of 12 v_mfma_f32_32x32x16_bf16 per 32 x 32 x 64 tile -- as ONE inline-asm
block for a wave's FOUR sets of 32 queries over the whole train set (<= 65536 rows), with the fused top-K fold (K = KEEP group
keys per lane and set; the certificate of the one-product pass needs K > 3).

Differences from gen_l2_segment_asm.py (the three-product pass, which now only sees the queries this pass cannot certify):
  * four query sets per wave (512 queries per workgroup): with a third of the MFMAs per train row the L2 -> LDS stream, the LDS
    reads and the per-tile barrier of the 256-query shape were as long as the matrix work (measured: 0.26 ms of 0.66 with all
    arithmetic removed).  Two accumulators per set do not fit beside 64 registers of B operands: six register sets and a skewed
    MFMA order (below) let every fold ride behind an MFMA all the same (a first version with four sets folded each set right in
    front of the MFMA that overwrote it: no overlap at all, fold and matrix time simply added up -- 0.47 + 0.21 = 0.68 ms);
  * the train image is bf16(t) only: 128-B rows, 16-B slot 2 ks + h holds a lane's A fragment of K-step ks; in LDS slot s of row r
    sits at physical slot s ^ ((r >> 1) & 7) -- every 16-lane group of a ds_read_b128 then covers the 64 banks exactly once;
  * a RING of tile buffers, 64 KiB in all: the LDS-DMA of tile t + RING is issued when tile t hands its buffer over; a tile's |t|^2
    travel through one VGPR per ring slot and are written to LDS -- kBig for rows past nt -- at the hand-over that publishes the
    tile.  Shipped: two tiles of 256 rows (one hand-over -- drain of the LDS queue, barrier, nine transfers to issue -- per 128
    MFMA slots; four tiles of 128 rows, measured on the same box: 0.91 - 0.92 ms against 0.89);
  * the A fragments are prefetched a whole 32-train step ahead (two register sets of four fragments);
  * a 13-bit position code (11 bits step, 2 bits group) in the low mantissa bits of a group key instead of 8: no segments, no
    master list -- the keys the block ends with name their rows directly.  The coarser keys (2^-10 relative) are noise next to
    the one-product pass's operand rounding (2^-8); train sets beyond 65536 rows skip this pass;
  * K keys per set: 3 + K VALU per group of four results (2 v_min3, v_and_or, K v_med3).  The 4 K keys leave through LDS (the
    ring is dead by then): key i of set s of thread tid at float (K s + i) * 256 + tid.

  step (32 train rows x 128 queries, 16 MFMAs).  Six accumulator register sets for the four query sets: sets 2 and 3 have ONE
  each, sets 0 and 1 alternate between two (even / odd steps).  MFMA order inside a step
      slot   0      1      2      3      4      5      6      7      8      9      10     11     12     13     14     15
      MFMA  (0,0)  (1,0)  (2,0)  (3,0)  (2,1)  (3,1)  (2,2)  (3,2)  (2,3)  (3,3)  (0,1)  (1,1)  (0,2)  (1,2)  (0,3)  (1,3)     (set, K-step)
  so that the single-buffered sets finish at slots 8 / 9 and are overwritten at slots 2 / 3 of the next step: their folds (two
  slots after the last MFMA: the results must have left the pipe) take the eight gaps behind slots 10 .. 15, 0', 1'; the folds
  of sets 0 and 1 (whose registers are not touched by the next step) the eight gaps behind slots 2' .. 9'.  Every gap carries
  (3 + K) VALU of one group of one set and a K-step's worth of LDS reads at most: uniform, 7 VALU per MFMA at K = 4.
  hand-over of tile t = the top of its last step: every read of the tile has been issued a step ago and is waited for, the wave's own
  transfers of tile t + 1 are waited for (vmcnt) and its norms written, barrier, then the norm load and the four DMA pieces of
  tile t + 4 go out (LDS-DMA destinations stay below 64 KiB: M0 carries the address).

Register map (all clobbered; b = 112 + 4 K):
  v[0:15] v[16:31]   accumulators of query sets 0, 1 in even steps      v[b+24 : b+56)  ... in odd steps
  v[32:47] v[48:63]  accumulators of query sets 2, 3
  v[64:79] v[80:95]  A fragments of even / odd steps: K-step ks = v[64 + 16 par + 4 ks : +3]
  v[96:111]          start values (|t|^2 of the step's 32 train rows, this lane's 16)
  v[112 : b)         keys: set s, rank i = v[112 + K s + i]
  b..b+3 a_addr   b+4 n_addr   b+5..b+8 DMA source offsets   b+9 norm source offset   b+10 norm LDS address
  b+11..b+14 ring norm registers   b+15..b+18 scratch   b+19..b+22 fold temporaries   b+23 key mask
  s40 tile  s41 code base of the step being folded  s42 step index  s43 saved M0  s44-s47 codes of the four groups
  s48-s51 scratch  s52 kBig  s53 -kBig
Operands (inputs only): %0-%15 B[s][ks] (s major);  %16 tile count;  %17 nt;  %18 train image buffer descriptor;  %19 train
  norms buffer descriptor;  %20 LDS address of the ring (norms behind it: the caller has written those of the first four tiles);
  %21 wave index.
"""
import os
import sys

KEEP = int(os.environ.get("ESFM_GEN_KEEP", "4"))
NOFOLD = int(os.environ.get("ESFM_GEN_NOFOLD", "0"))      # timing experiments only: no fold / no MFMA
NOMFMA = int(os.environ.get("ESFM_GEN_NOMFMA", "0"))
NOBAR = int(os.environ.get("ESFM_GEN_NOBAR", "0"))        # timing experiments only (wrong results): no barrier / no wait for the
NOVMWAIT = int(os.environ.get("ESFM_GEN_NOVMWAIT", "0"))  # wave's transfers at a hand-over, no waits for LDS reads inside a step,
NOLGKM = int(os.environ.get("ESFM_GEN_NOLGKM", "0"))      # no LDS reads at all
NOLDSRD = int(os.environ.get("ESFM_GEN_NOLDSRD", "0"))
MFMA = os.environ.get("ESFM_GEN_MFMA", "bf16")            # "fp4": v_mfma_f32_32x32x64_f8f6f4 on e2m1 nibbles -- the 256-bit Hamming matcher: a row of
assert MFMA in ("bf16", "fp4")                            # 256 nibbles is 128 B, four K-steps of 64, a lane's 16-B slot 2 ks + h its fragment: the same shapes
PREFIX = os.environ.get("ESFM_GEN_PREFIX", "ESFM_L2X1")   # macro prefix of the generated file
NS = 4
GRP = int(os.environ.get("ESFM_GEN_GRP", "8"))            # results per fold group: 4 (round 3) or 8 (round 4: 4.5 instead of 7 VALU per MFMA at K = 4)
assert GRP in (4, 8, 16)                                  # (16: the Hamming form only -- a kept group costs the exact tail 16 rows of 32 B there, of 256 B in the L2 pass)
NG = 16 // GRP                                            # groups per lane and 32-train step
STEP_BITS = int(os.environ.get("ESFM_GEN_STEP_BITS", "11"))   # (the FP4 Hamming form's scores leave 14 zero mantissa bits: 13 step bits = 262 144 rows)
CODE_BITS = STEP_BITS + {4: 2, 8: 1, 16: 0}[GRP]          # position code: step, 2 / 1 / 0 bits group
KBIG = 0x7F61B1E6          # 3.0e38f
NKBIG = 0xFF61B1E6
TT = int(os.environ.get("ESFM_GEN_TT", "256"))            # train rows per tile (128 or 256)
RING = int(os.environ.get("ESFM_GEN_RING", "2"))          # tile buffers (TT * RING = 512: 64 KiB of LDS)
ROW_BYTES = 128
assert TT in (128, 256) and TT * RING == 512
STEPS = TT // 32                      # 32-train steps per tile
NP = TT // 32                         # LDS-DMA pieces (8 rows = 1 KiB) per wave and tile
WROWS = TT // 4                       # rows of a tile staged by one wave
TILE_BYTES = TT * ROW_BYTES          # 16 / 32 KiB
NORM_BASE = RING * TILE_BYTES        # the ring's norms sit behind the tiles: RING x TT floats
assert (2 if os.environ.get('ESFM_GEN_MFMA', 'bf16') == 'fp4' else 3) <= KEEP <= 6      # (exact scores need no certificate: two keys per lane half are enough)

OP_B = lambda s, ks: f"%{4 * s + ks}"
OP_NTILES, OP_NT, OP_TRSRC, OP_NRSRC, OP_LDS, OP_WAVE = (f"%{16 + i}" for i in range(6))

KEY = lambda s, i: f"v{112 + KEEP * s + i}"
_b = 112 + 4 * KEEP            # (registers are packed: beside 64 registers of B operands every one counts)
A_ADDR, N_ADDR, DMA_OFF, NSRC_OFF, NLDS = _b, _b + 4, _b + 5, _b + 9, _b + 10
NR = lambda b: f"v{_b + 11 + b}"
SCR = _b + 15                  # 4 scratch registers
FTMP = _b + 19                 # 4: (t, key) pairs, alternating
MASK = _b + 23
ACC_ODD = _b + 24              # accumulators of sets 0, 1 in odd steps: 32 registers
N_CLOBBER = ACC_ODD + 32


def acc(s, par=0):
    return 16 * s if (s >= 2 or par == 0) else ACC_ODD + 16 * s


def accr(s, par=0):
    b = acc(s, par)
    return f"v[{b}:{b + 15}]"


def fr(ks, par):
    b = 64 + 16 * par + 4 * ks
    return f"v[{b}:{b + 3}]"


def gen():
    L = []
    e = L.append
    # ------------------------------------------------------------------ set-up
    e("s_mov_b32 s43, m0")
    e(f"v_mbcnt_lo_u32_b32 v{SCR}, -1, 0")
    e(f"v_mbcnt_hi_u32_b32 v{SCR}, -1, v{SCR}")             # lane
    e(f"v_and_b32 v{SCR + 1}, 31, v{SCR}")                   # j
    e(f"v_lshrrev_b32 v{SCR + 2}, 5, v{SCR}")                # h
    e(f"v_bfe_u32 v{SCR + 3}, v{SCR + 1}, 1, 3")             # (j >> 1) & 7: the row's swizzle
    e(f"v_lshlrev_b32 v{NLDS}, 7, v{SCR + 1}")               # j * 128
    e(f"v_add_u32 v{NLDS}, {OP_LDS}, v{NLDS}")               # row j of buffer 0, step 0
    for ks in range(4):
        e(f"v_or_b32 v{FTMP}, {2 * ks}, v{SCR + 2}")          # logical slot 2 ks + h
        e(f"v_xor_b32 v{FTMP}, v{FTMP}, v{SCR + 3}")
        e(f"v_lshl_add_u32 v{A_ADDR + ks}, v{FTMP}, 4, v{NLDS}")
    e(f"v_lshlrev_b32 v{N_ADDR}, 4, v{SCR + 2}")
    e(f"v_add_u32 v{N_ADDR}, {OP_LDS}, v{N_ADDR}")
    e(f"v_add_u32 v{N_ADDR}, {NORM_BASE}, v{N_ADDR}")         # n_addr = lds_norm + 16 h
    # LDS-DMA source offsets: wave w stages rows [WROWS w, WROWS (w + 1)) of a tile, 8 rows (1 KiB) per instruction; pieces i and
    # i + 4 share the swizzle (32 rows apart) and the register
    e(f"s_mul_i32 s48, {OP_WAVE}, {WROWS}")
    e(f"v_lshrrev_b32 v{SCR + 1}, 3, v{SCR}")                 # lane >> 3
    e(f"v_and_b32 v{SCR + 3}, 7, v{SCR}")                     # lane & 7: physical slot
    for i in range(4):
        e(f"v_add_u32 v{FTMP}, s48, v{SCR + 1}")
        e(f"v_add_u32 v{FTMP}, {8 * i}, v{FTMP}")             # row in the tile
        e(f"v_bfe_u32 v{FTMP + 1}, v{FTMP}, 1, 3")
        e(f"v_xor_b32 v{FTMP + 1}, v{FTMP + 1}, v{SCR + 3}")  # logical slot fetched into this physical slot
        e(f"v_lshlrev_b32 v{FTMP + 1}, 4, v{FTMP + 1}")
        e(f"v_lshl_add_u32 v{DMA_OFF + i}, v{FTMP}, 7, v{FTMP + 1}")
    # norms: lane l of wave w moves |t|^2 of row WROWS w + (l & (WROWS - 1)) (lanes that share a row move the same value)
    e(f"v_and_b32 v{FTMP}, {WROWS - 1}, v{SCR}")
    e(f"v_add_u32 v{FTMP}, s48, v{FTMP}")                     # row in the tile
    e(f"v_lshlrev_b32 v{NSRC_OFF}, 2, v{FTMP}")               # byte offset inside a tile's norms
    e(f"v_add_u32 v{NLDS}, {OP_LDS}, v{NSRC_OFF}")
    e(f"v_add_u32 v{NLDS}, {NORM_BASE}, v{NLDS}")             # lds_norm + 4 row (ring slot 0)
    e(f"s_mov_b32 s52, 0x{KBIG:08x}")
    e(f"s_mov_b32 s53, 0x{NKBIG:08x}")
    e(f"v_mov_b32 v{MASK}, 0x{(0xFFFFFFFF << CODE_BITS) & 0xFFFFFFFF:08x}")
    for s in range(NS):
        for i in range(KEEP):
            e(f"v_mov_b32 {KEY(s, i)}, s52")
    for r in list(range(0, 64)) + list(range(ACC_ODD, ACC_ODD + 32)) + [FTMP + 1, FTMP + 3]:
        e(f"v_mov_b32 v{r}, s52")                             # placeholders for the step before the first: never live (the fold temporaries:
        #                                                       GRP = 8 runs the second half of a fold of that step in the first step's slots 0 / 1)
    e("s_mov_b32 s40, 0")
    e("s_mov_b32 s41, 0")
    e("s_mov_b32 s42, 0")
    for g in range(NG):
        e(f"s_mov_b32 s{44 + g}, {g}")
    # tile 0 has landed for every wave (the caller's barrier).  Pipeline fill: the start values and the four fragments of step 0
    for g in range(4):
        e(f"ds_read_b128 v[{96 + 4 * g}:{99 + 4 * g}], v{N_ADDR} offset:{32 * g}")
    for ks in range(4):
        e(f"ds_read_b128 {fr(ks, 0)}, v{A_ADDR + ks}")

    cnt = [0]

    def fold_group(s, g, par):
        """Quarter g (0..3) of the fold of set s' 16 results of the step with parity `par`.
        GRP = 4: the whole fold of group g (accumulator registers 4 g .. 4 g + 3): 3 + K VALU.
        GRP = 8: groups are register octets (0..7 = rows {0..3, 8..11} + 4 h of the step, 8..15 = rows {16..19, 24..27} + 4 h); quarter
        2 G is the first half of group G's fold -- the minimum of its eight scores and the position code, 5 VALU, into the set
        parity's temporary -- quarter 2 G + 1 the second half: the K v_med3 (it reads the temporary only, so it may run after the
        octet has been overwritten).  4.5 VALU per MFMA at K = 4 instead of 7."""
        if NOFOLD:
            return []
        if GRP == 4:
            b = acc(s, par) + 4 * g
            cnt[0] ^= 1
            t, key = FTMP + 2 * cnt[0], FTMP + 1 + 2 * cnt[0]
            out = [f"v_min3_f32 v{t}, v{b}, s52, v{b + 1}",
                   f"v_min3_f32 v{t}, v{t}, v{b + 2}, v{b + 3}",
                   f"v_and_or_b32 v{key}, v{t}, v{MASK}, s{44 + g}"]
            for i in range(KEEP - 1, 0, -1):
                out.append(f"v_med3_f32 {KEY(s, i)}, {KEY(s, i - 1)}, {KEY(s, i)}, v{key}")
            out.append(f"v_med3_f32 {KEY(s, 0)}, {KEY(s, 0)}, v{key}, s53")
            return out
        if GRP == 16:
            # one group per lane, set and step: quarters 0 / 1 the minimum of its registers 0..7 / 8..15 (4 v_min3 each), quarter 2 the code
            # and the K v_med3, quarter 3 nothing: 9 + K VALU per 16 results
            t, key = FTMP + 2 * (s & 1), FTMP + 1 + 2 * (s & 1)
            b = acc(s, par) + 8 * g
            if g == 0:
                return [f"v_min3_f32 v{t}, v{b}, s52, v{b + 1}", f"v_min3_f32 v{t}, v{t}, v{b + 2}, v{b + 3}",
                        f"v_min3_f32 v{t}, v{t}, v{b + 4}, v{b + 5}", f"v_min3_f32 v{t}, v{t}, v{b + 6}, v{b + 7}"]
            if g == 1:
                return [f"v_min3_f32 v{t}, v{t}, v{b}, v{b + 1}", f"v_min3_f32 v{t}, v{t}, v{b + 2}, v{b + 3}",
                        f"v_min3_f32 v{t}, v{t}, v{b + 4}, v{b + 5}", f"v_min3_f32 v{t}, v{t}, v{b + 6}, v{b + 7}"]
            if g == 3:
                return []
            out = [f"v_and_or_b32 v{key}, v{t}, v{MASK}, s44"]
            for i in range(KEEP - 1, 0, -1):
                out.append(f"v_med3_f32 {KEY(s, i)}, {KEY(s, i - 1)}, {KEY(s, i)}, v{key}")
            out.append(f"v_med3_f32 {KEY(s, 0)}, {KEY(s, 0)}, v{key}, s53")
            return out
        G, half = g >> 1, g & 1
        t, key = FTMP + 2 * (s & 1), FTMP + 1 + 2 * (s & 1)        # sets of equal parity never have a fold in flight at the same time
        if half == 0:
            b = acc(s, par) + 8 * G
            return [f"v_min3_f32 v{t}, v{b}, s52, v{b + 1}",
                    f"v_min3_f32 v{t}, v{t}, v{b + 2}, v{b + 3}",
                    f"v_min3_f32 v{t}, v{t}, v{b + 4}, v{b + 5}",
                    f"v_min3_f32 v{t}, v{t}, v{b + 6}, v{b + 7}",
                    f"v_and_or_b32 v{key}, v{t}, v{MASK}, s{44 + G}"]
        out = []
        for i in range(KEEP - 1, 0, -1):
            out.append(f"v_med3_f32 {KEY(s, i)}, {KEY(s, i - 1)}, {KEY(s, i)}, v{key}")
        out.append(f"v_med3_f32 {KEY(s, 0)}, {KEY(s, 0)}, v{key}, s53")
        return out

    # slot -> (set, K-step)
    ORDER = [(0, 0), (1, 0), (2, 0), (3, 0), (2, 1), (3, 1), (2, 2), (3, 2), (2, 3), (3, 3), (0, 1), (1, 1), (0, 2), (1, 2), (0, 3), (1, 3)]
    # fold of the group behind each slot: ("cur", set, group) = this step's results, ("prev", ...) = the previous step's
    GAP = {10: ("cur", 2, 0), 11: ("cur", 3, 0), 12: ("cur", 2, 1), 13: ("cur", 3, 1), 14: ("cur", 2, 2), 15: ("cur", 3, 2),
           0: ("prev", 2, 3), 1: ("prev", 3, 3),
           2: ("prev", 0, 0), 3: ("prev", 1, 0), 4: ("prev", 0, 1), 5: ("prev", 1, 1), 6: ("prev", 0, 2), 7: ("prev", 1, 2), 8: ("prev", 0, 3), 9: ("prev", 1, 3)}
    # LDS reads behind a slot: fragment ks of the next step (into the other parity's registers) and, behind slot 9, the next
    # step's 16 start values (this step's were last read by slot 3)
    READS = {1: ("frag", 0), 5: ("frag", 1), 9: ("start", None), 11: ("frag", 2), 13: ("frag", 3)}
    # Waits in front of a slot: the fragment it reads (issued one step ago) must have landed.  Queue order per step:
    # F0 (slot 1) | F1 (slot 5) | N x 4 (slot 9) | F2 (slot 11) | F3 (slot 13).  Slot 0 needs F0 and the start values: the younger
    # F2, F3 may fly (F1 is older: landed as well, nothing to wait for at its first use, slot 4); slot 6 needs F2: younger are F3 and
    # this step's F0', F1'; slot 8 needs F3: younger are F0', F1'.
    WAITS = {0: 2, 6: 3, 8: 2}

    def step(par, spar, nbuf, nstep, extra=None):
        """One 32-train step: fragments of parity `par`, accumulators of sets 0 / 1 of parity `spar`; prefetches the next step
        (buffer nbuf, step nstep).  extra: instruction lists issued behind slots 4 .. 8 (the hand-over's transfers)."""
        extra = list(extra or [])
        for slot, (s, ks) in enumerate(ORDER):
            if slot in WAITS and not NOLGKM and not NOLDSRD:
                e(f"s_waitcnt lgkmcnt({WAITS[slot]})")
            if slot == 10:
                # the step being folded from now on is this one
                e(f"s_lshl_b32 s41, s42, {CODE_BITS - STEP_BITS}")
                e("s_add_u32 s42, s42, 1")
                for g in range(NG):
                    e(f"s_add_u32 s{44 + g}, s41, {g}")
            if not NOMFMA:
                a = accr(s, spar)
                c_in = 'v[96:111]' if ks == 0 else a
                if MFMA == "fp4":
                    e(f"v_mfma_f32_32x32x64_f8f6f4 {a}, {fr(ks, par)}, {OP_B(s, ks)}, {c_in} cbsz:4 blgp:4")
                else:
                    e(f"v_mfma_f32_32x32x16_bf16 {a}, {fr(ks, par)}, {OP_B(s, ks)}, {c_in}")
            if slot >= 4 and extra:
                for x in extra.pop(0):
                    e(x)
            which, fs, fg = GAP[slot]
            for x in fold_group(fs, fg, spar if which == "cur" else spar ^ 1):
                e(x)
            if slot in READS and not NOLDSRD:
                kind, k = READS[slot]
                if kind == "start":
                    for g in range(4):
                        e(f"ds_read_b128 v[{96 + 4 * g}:{99 + 4 * g}], v{N_ADDR} offset:{(nbuf * TT + nstep * 32 + 8 * g) * 4}")
                else:
                    e(f"ds_read_b128 {fr(k, par ^ 1)}, v{A_ADDR + k} offset:{nbuf * TILE_BYTES + nstep * 32 * ROW_BYTES}")
        assert not extra

    def dma_piece(i, buf):
        out = [f"s_add_u32 s50, s49, {buf * TILE_BYTES + i * 1024}",
               "s_mov_b32 m0, s50"]
        if i >= 4:
            out.append(f"s_add_u32 s50, s48, {(i >> 2) * 32 * ROW_BYTES}")
        out.append(f"buffer_load_dwordx4 v{DMA_OFF + (i & 3)}, {OP_TRSRC}, {'s50' if i >= 4 else 's48'} offen lds")
        return out

    def tile(buf, tag):
        nb = (buf + 1) % RING
        for st in range(STEPS - 1):
            step(st & 1, st & 1, buf, st + 1)
        # ---- last step: hand-over first
        e("s_waitcnt lgkmcnt(0)")                               # every read of this tile has landed
        if not NOVMWAIT:
            e(f"s_waitcnt vmcnt({(RING - 2) * NP})")           # this wave's transfers of tile + 1 have landed (younger tiles may fly;
        #                                                         counted in pieces only: the caller's first tiles come without norm loads)
        # norms of tile + 1 -> LDS, rows past nt as kBig (the first RING tiles' norms were written by the caller)
        e(f"s_cmp_lt_u32 s40, {RING - 1}")
        e(f"s_cbranch_scc1 L_nonorm_{tag}_%=")
        e("s_add_u32 s48, s40, 1")
        e(f"s_mul_i32 s48, s48, {TT}")
        e(f"v_lshrrev_b32 v{SCR}, 2, v{NSRC_OFF}")
        e(f"v_add_u32 v{SCR}, s48, v{SCR}")                    # train row of this lane's norm
        e(f"v_cmp_gt_u32 vcc, {OP_NT}, v{SCR}")
        e(f"v_mov_b32 v{SCR + 1}, s52")
        e(f"v_cndmask_b32 v{SCR + 1}, v{SCR + 1}, {NR(nb)}, vcc")
        e(f"ds_write_b32 v{NLDS}, v{SCR + 1} offset:{nb * TT * 4}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"L_nonorm_{tag}_%=:")
        if not NOBAR:
            e("s_barrier")
        e(f"s_add_u32 s48, s40, {RING}")
        e(f"s_mul_i32 s51, s48, {TT * 4}")                      # (tile + RING) * TT * 4: its norms
        e(f"s_mul_i32 s48, s48, {TILE_BYTES}")                  # ... its rows
        e(f"s_mul_i32 s49, {OP_WAVE}, {WROWS * ROW_BYTES}")     # this wave's rows inside a tile
        e(f"s_add_u32 s49, s49, {OP_LDS}")
        # (a tile that does not exist reads zeros through the descriptors into a buffer nobody reads again: unconditional)
        pcs = [[f"buffer_load_dword {NR(buf)}, v{NSRC_OFF}, {OP_NRSRC}, s51 offen"]] + [dma_piece(i, buf) for i in range(NP)]
        step(1, 1, nb, 0, extra=pcs)
        e("s_add_u32 s40, s40, 1")

    e("L_top_%=:")
    for buf in range(RING):
        tile(buf, f"t{buf}")
        if buf < RING - 1:
            e(f"s_cmp_ge_u32 s40, {OP_NTILES}")
            e("s_cbranch_scc1 L_done_%=")
    e(f"s_cmp_lt_u32 s40, {OP_NTILES}")
    e("s_cbranch_scc1 L_top_%=")
    e("L_done_%=:")
    # what the last step (parity 1) left unfolded: group 3 of its sets 2 and 3, all of its sets 0 and 1 -- after the matrix pipe
    # has delivered them
    e("s_nop 15")
    e("s_nop 15")
    e("s_waitcnt lgkmcnt(0)")                                   # the prefetches past the end land in dead registers
    for fs, fg in [(2, 3), (3, 3)] + [(s_, g_) for s_ in (0, 1) for g_ in range(4)]:
        for x in fold_group(fs, fg, 1):
            e(x)
    # keys -> LDS (the ring is dead once every wave is here and every transfer has landed)
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    e(f"v_mbcnt_lo_u32_b32 v{SCR}, -1, 0")
    e(f"v_mbcnt_hi_u32_b32 v{SCR}, -1, v{SCR}")
    e(f"s_lshl_b32 s48, {OP_WAVE}, 6")
    e(f"v_add_u32 v{SCR}, s48, v{SCR}")                         # tid
    e(f"v_lshlrev_b32 v{SCR}, 2, v{SCR}")
    e(f"v_add_u32 v{SCR}, {OP_LDS}, v{SCR}")
    for s in range(NS):
        for i in range(KEEP):
            e(f"ds_write_b32 v{SCR}, {KEY(s, i)} offset:{(KEEP * s + i) * 1024}")
    e("s_waitcnt lgkmcnt(0)")
    e("s_mov_b32 m0, s43")
    return L


def main():
    lines = gen()
    clob = [f"v{i}" for i in range(N_CLOBBER)] + [f"s{i}" for i in range(40, 54)] + ["scc", "vcc", "memory"]
    out = ["// GENERATED by gen_l2x1_segment_asm.py -- do not edit; see that file for the schedule and the register map",
           f"#define {PREFIX}_KEEP {KEEP}",
           f"#define {PREFIX}_SETS {NS}",
           f"#define {PREFIX}_CODE_BITS {CODE_BITS}",
           f"#define {PREFIX}_GRP {GRP}",
           f"#define {PREFIX}_TT {TT}",
           f"#define {PREFIX}_RING {RING}",
           f"#define {PREFIX}_SEGMENT_ASM \\"]
    for l in lines:
        out.append(f'    "{l}\\n" \\')
    out.append('    ""')
    out.append(f"#define {PREFIX}_SEGMENT_CLOBBERS " + ", ".join(f'"{c}"' for c in clob))
    open(sys.argv[1] if len(sys.argv) > 1 else "l2x1_segment_gfx950.inc", "w").write("\n".join(out) + "\n")
    n_mfma = sum("v_mfma" in l for l in lines)
    print(f"{len(lines)} instructions, {n_mfma} MFMAs, KEEP = {KEEP}, GRP = {GRP}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generates potrf16_gfx950.inc: the one-wave 16 x 16 "Cholesky factor + inverse of the factor" of the reduced camera solve as ONE
inline-asm block for gfx950 (ba_chol_large.hip: potrf16_fused_inv).

Why asm: the n pivots of the reduced system are a serial chain, and the bulk of the work (column updates of the block and of the
inverse, 2 instructions per (pivot, column)) has to be issued in the shadow of that chain's latencies.  hipcc does not do that: with
v_readlane broadcasts it parks the scalars in VGPR lanes (v_writelane + s_nop hazards, 525 cycles per pivot), with DPP broadcasts it
hoists all of them and spills to scratch.  Here a small list scheduler places the instructions (critical path first) and inserts the
wait states the ISA asks for by hand.

Layout: lane r (= lane & 15; the four 16-lane rows of the wave work redundantly) holds row r of the block in x[0..15] and solves
L y = e_r in t[0..15].  Once column c is final, L[i][c] -- row_newbcast:i of x[c] -- updates x[i] -= x[c] L[i][c] and
t[i] -= L[i][c] t[c], each ONE v_fmac_f64_dpp (the broadcast rides in the multiply-add).  1 / sqrt(pivot): v_rsq_f64 + one Newton
step (relative error <= 1.5 * 2^-52).

History per 16 x 16 block inside the 64 x 64 tile factorisation (ba_chol_large.hip, s_memrealtime): v_mov_b64_dpp + two v_fma_f64 per
(pivot, column): 1.74 us; v_fmac_f64_dpp (18 % fewer instructions): 1.65 us.  Both fit "a DPP f64 instruction issues in 8 cycles, any
other in 4, nothing else matters" (168 x 8 + 572 x 4 = 3.6 k cycles; 352 x 8 + 200 x 4 = 3.5 k).  Tried and dropped: a five-deep
instead of seven-deep pivot chain (v_rsq on every lane's own element so that no broadcast precedes it, the column's scale folded into
the Newton step's last multiply as v_fmac_f64_dpp onto a zeroed register): the same 1.66 us -- the chain is not what bounds the block
-- and a different rounding of the factor, which moves BA-25's iteration log.

Operands:  %0 (out, v) bad-pivot flag;  %1 (in, v) LDS byte address of row r of the block;  %2 (in, v) LDS byte address of
column r of the inverse (written column-wise: lane r holds column r; row pitch PITCH bytes).
The block's rows are only read: nothing uses a factored diagonal block again (panels and back-substitution go through the inverse).
"""
import os
import sys

FMAC_DPP = os.environ.get("ESFM_GEN_POTRF_FMAC_DPP", "1") == "1"   # v_fmac_f64_dpp: the broadcast rides in the multiply-add (round 3)
B64DPP = os.environ.get("ESFM_GEN_POTRF_B64DPP", "1") == "1"   # one v_mov_b64_dpp per broadcast (15.3 us per solve at n = 150) instead of two v_mov_b32_dpp (17.5 us)

SB = 16
PITCH = 18 * 8          # VLD doubles

def X(c): return 2 * c
def T(c): return 32 + 2 * c
PIV, Y, H, W, RINV = 64, 66, 68, 70, 72
BC0, NBC = 74, 6
C15 = 86                # 1.5
NV = 88                 # v0..v87
S_MASK, S_CMP, S_BAD = 4, 6, 8

def pair(v): return "v[%d:%d]" % (v, v + 1)

class Ins:
    __slots__ = ("text", "reads", "writes", "kind", "prio", "succ", "npred", "ready_at", "idx", "dpp_reads")
    def __init__(self, text, reads, writes, kind):
        self.text, self.reads, self.writes, self.kind = text, set(reads), set(writes), kind
        self.succ, self.npred, self.prio, self.ready_at = [], 0, 0, 0
        self.dpp_reads = self.reads if kind == "dpp" else set()      # operands read THROUGH the DPP network (the ones with the hazard)

def regs(v): return (v, v + 1)

prog = []
def emit(text, reads, writes, kind="dp"):
    prog.append(Ins(text, reads, writes, kind))

def dpp_bcast(dst, src, lane):
    if B64DPP:
        emit("v_mov_b64_dpp %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % (pair(dst), pair(src), lane), regs(src), regs(dst), "dpp")
        return
    # two 32-bit row broadcasts (bound_ctrl: no dependence on the old destination)
    emit("v_mov_b32_dpp v%d, v%d row_newbcast:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % (dst, src, lane), [src], [dst], "dpp")
    emit("v_mov_b32_dpp v%d, v%d row_newbcast:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % (dst + 1, src + 1, lane), [src + 1], [dst + 1], "dpp")

def fmac_dpp(dst, bsrc, lane, src1):
    """dst += -row_newbcast:lane(bsrc) * src1"""
    emit("v_fmac_f64_dpp %s, -%s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf" % (pair(dst), pair(bsrc), pair(src1), lane),
         regs(bsrc) + regs(src1) + regs(dst), regs(dst), "dppfma")
    prog[-1].dpp_reads = set(regs(bsrc))

def build():
    del prog[:]
    bc_next = [0]
    def new_bc():
        v = BC0 + 2 * (bc_next[0] % NBC); bc_next[0] += 1; return v
    dpp_bcast(PIV, X(0), 0)
    for c in range(SB):
        # pivot checks (off the chain): anything but a positive finite number is bad
        emit("v_cmp_class_f64_e64 s[%d:%d], %s, s%d" % (S_CMP, S_CMP + 1, pair(PIV), S_MASK), regs(PIV), ["scmp"], "cmp")
        emit("s_or_b64 s[%d:%d], s[%d:%d], s[%d:%d]" % (S_BAD, S_BAD + 1, S_BAD, S_BAD + 1, S_CMP, S_CMP + 1), ["scmp", "sbad"], ["sbad"], "salu")
        emit("v_rsq_f64_e32 %s, %s" % (pair(Y), pair(PIV)), regs(PIV), regs(Y), "trans")
        emit("v_mul_f64 %s, %s, -0.5" % (pair(H), pair(PIV)), regs(PIV), regs(H))
        emit("v_mul_f64 %s, %s, %s" % (pair(W), pair(Y), pair(Y)), regs(Y), regs(W))
        emit("v_fma_f64 %s, %s, %s, %s" % (pair(W), pair(H), pair(W), pair(C15)), regs(H) + regs(W) + regs(C15), regs(W))
        emit("v_mul_f64 %s, %s, %s" % (pair(RINV), pair(Y), pair(W)), regs(Y) + regs(W), regs(RINV))
        emit("v_mul_f64 %s, %s, %s" % (pair(X(c)), pair(X(c)), pair(RINV)), regs(X(c)) + regs(RINV), regs(X(c)))
        emit("v_mul_f64 %s, %s, %s" % (pair(T(c)), pair(T(c)), pair(RINV)), regs(T(c)) + regs(RINV), regs(T(c)))
        for i in range(c + 1, SB):
            if FMAC_DPP:
                # broadcast and multiply-add in one instruction: D += -bcast_i(x[c]) * src1  (v_fmac_f64 is VOP2 on gfx90a+, and its
                # DPP form takes row_newbcast like v_mov_b64_dpp): 2 instead of 3 instructions per (pivot, column)
                fmac_dpp(X(i), X(c), i, X(c))
                if i == c + 1:
                    dpp_bcast(PIV, X(i), i)
                fmac_dpp(T(i), X(c), i, T(c))
                continue
            bc = new_bc()
            dpp_bcast(bc, X(c), i)
            emit("v_fma_f64 %s, -%s, %s, %s" % (pair(X(i)), pair(X(c)), pair(bc), pair(X(i))), regs(X(c)) + regs(bc) + regs(X(i)), regs(X(i)))
            if i == c + 1:
                dpp_bcast(PIV, X(i), i)
            emit("v_fma_f64 %s, -%s, %s, %s" % (pair(T(i)), pair(bc), pair(T(c)), pair(T(i))), regs(bc) + regs(T(c)) + regs(T(i)), regs(T(i)))

LAT = {"dp": 10, "trans": 20, "dpp": 6, "dppfma": 12, "cmp": 8, "salu": 2, "lds": 4}     # issue-to-use estimates (cycles); only the ORDER depends on them
ISSUE = {"dp": 4, "trans": 8, "dpp": 8, "dppfma": 8, "cmp": 4, "salu": 1, "lds": 4}

def schedule():
    n = len(prog)
    last_w, readers = {}, {}
    for k, ins in enumerate(prog):
        ins.idx = k
        preds = set()
        for r in ins.reads:
            if r in last_w: preds.add(last_w[r])
        for w in ins.writes:
            if w in last_w: preds.add(last_w[w])                      # WAW
            for rd in readers.get(w, []): preds.add(rd)                # WAR
        preds.discard(k)
        for p in preds: prog[p].succ.append(k)
        ins.npred = len(preds)
        for r in ins.reads: readers.setdefault(r, []).append(k)
        for w in ins.writes: last_w[w] = k; readers[w] = []
    for ins in reversed(prog):
        ins.prio = LAT[ins.kind] + max([prog[s].prio for s in ins.succ], default=0)
    ready = [k for k in range(n) if prog[k].npred == 0]
    order, t = [], 0
    done_at = {}
    while ready:
        # an instruction whose inputs are available now, longest path first; otherwise the one available soonest
        avail = [k for k in ready if prog[k].ready_at <= t]
        pick = max(avail, key=lambda k: (prog[k].prio, -k)) if avail else min(ready, key=lambda k: (prog[k].ready_at, -prog[k].prio))
        ready.remove(pick)
        ins = prog[pick]
        t = max(t, ins.ready_at) + ISSUE[ins.kind]
        order.append(pick)
        for s in ins.succ:
            nxt = prog[s]
            # true dependence: wait for the result; WAR / WAW: issue order is enough
            lat = LAT[ins.kind] if (ins.writes & nxt.reads) else 0
            nxt.ready_at = max(nxt.ready_at, t - ISSUE[ins.kind] + lat)
            nxt.npred -= 1
            if nxt.npred == 0: ready.append(s)
    return order, t

def hazards(order):
    """Wait states the hardware does not interlock (MI300 ISA 4.5): a DPP instruction reading a VGPR written by one of the two
    previous VALU instructions needs 2 wait states; a trans result consumed by the very next VALU instruction 1."""
    out = []
    recent = []            # (writes, kind) of the last issued instructions, newest last; s_nop counts as that many slots
    for k in order:
        ins = prog[k]
        need = 0
        if ins.dpp_reads:
            for back, (w, kind) in enumerate(reversed(recent[-2:])):
                if w is not None and (ins.dpp_reads & w): need = max(need, 2 - back)
        if recent and recent[-1][0] is not None and recent[-1][1] == "trans" and (ins.reads & recent[-1][0]) and ins.kind != "salu":
            need = max(need, 1)
        if ins.kind == "salu" and recent and recent[-1][1] == "cmp":
            need = max(need, 0)        # s_or of a VALU-written SGPR pair: interlocked
        if need:
            out.append("s_nop %d" % (need - 1))
            recent += [(None, "nop")] * need
        out.append(ins.text)
        recent.append((set(x for x in ins.writes if isinstance(x, int)), ins.kind))
    return out

def generate():
    build()
    order, cycles = schedule()
    body = hazards(order)
    pre = []
    # rows in: 8 x ds_read_b128 (row pitch a multiple of 16 B)
    for c in range(0, SB, 2):
        pre.append("ds_read_b128 v[%d:%d], %%1 offset:%d" % (X(c), X(c) + 3, 8 * c))
    # t[c] = (lane & 15) == c ? 1.0 : 0.0 -- the lane mask walks through an SGPR pair (2 VALU instructions per column, not 3)
    pre.append("v_mov_b32 v%d, 0x3ff00000" % H)                      # H holds the high word of 1.0 until the first pivot
    pre.append("s_mov_b32 s%d, 0x00010001" % S_CMP)
    pre.append("s_mov_b32 s%d, 0x00010001" % (S_CMP + 1))
    for c in range(SB):
        pre.append("v_mov_b32 v%d, 0" % T(c))
        pre.append("v_cndmask_b32_e64 v%d, 0, v%d, s[%d:%d]" % (T(c) + 1, H, S_CMP, S_CMP + 1))
        if c + 1 < SB: pre.append("s_lshl_b64 s[%d:%d], s[%d:%d], 1" % (S_CMP, S_CMP + 1, S_CMP, S_CMP + 1))
    pre.append("v_mov_b32 v%d, 0" % C15)
    pre.append("v_mov_b32 v%d, 0x3ff80000" % (C15 + 1))
    pre.append("s_movk_i32 s%d, 0x27f" % S_MASK)                     # v_cmp_class mask: NaN, -anything, +-0, +inf
    pre.append("s_mov_b64 s[%d:%d], 0" % (S_BAD, S_BAD + 1))
    pre.append("s_waitcnt lgkmcnt(0)")
    post = []
    for i in range(SB):
        post.append("ds_write_b64 %%2, %s offset:%d" % (pair(T(i)), PITCH * i))   # column r of the inverse: exactly +-0 above the diagonal
    post.append("v_cndmask_b32_e64 %%0, 0, 1, s[%d:%d]" % (S_BAD, S_BAD + 1))
    post.append("s_waitcnt lgkmcnt(0)")
    return pre + body + post, cycles

def main(path):
    with open(path, "w") as f:
        f.write("// GENERATED by gen_potrf16_asm.py -- do not edit.\n")
        lines, cycles = generate()
        f.write("// ESFM_POTRF16_ASM: %d instructions, scheduler estimate %d cycles\n" % (len(lines), cycles))
        f.write("#define ESFM_POTRF16_ASM \\\n")
        for ln in lines:
            f.write('    "%s\\n\\t" \\\n' % ln)
        f.write('    ""\n')
        print("ESFM_POTRF16_ASM: %d instructions, estimated %d cycles" % (len(lines), cycles))
        clob = ['"v%d"' % v for v in range(NV)] + ['"s%d"' % s for s in (S_MASK, S_CMP, S_CMP + 1, S_BAD, S_BAD + 1)] + ['"vcc"', '"memory"']
        f.write("#define ESFM_POTRF16_CLOBBERS " + ", ".join(clob) + "\n")

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "potrf16_gfx950.inc")

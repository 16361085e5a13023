"""Launch-shape split of l2_knn_bf16x1_kernel: the same kernel on pair lists that differ in how they fill the chip
(512 resident workgroup slots = 2 per CU; a 512-query block per workgroup) and in how much main loop a block's set-up is spread over."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easysfm_amd as E
from easysfm_amd import synth, _lib

def run(name, sets, pairs, reps=30):
    pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), pairs)
    for _ in range(5): pm.match(0.5)
    pm.ctx.synchronize(); pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_L2_KNN); pm.ctx.kernel_time(_lib.K_L2_SECOND)
    for _ in range(reps): pm.match(0.5)
    pm.ctx.synchronize()
    k = pm.ctx.kernel_time(_lib.K_L2_KNN); pm.ctx.set_kernel_timing(False)
    ms = k[0] / max(k[1], 1)
    fl = sum(2.0 * len(sets[i]) * len(sets[j]) * 64 for i, j in pairs)
    blocks = sum((len(sets[i]) + 511) // 512 for i, j in pairs)
    print(f"{name:58s} {len(pairs):5d} pairs {blocks:6d} blocks = {blocks / 512:6.2f} rounds of 512 slots: pass {ms:8.4f} ms, {fl / ms / 1e9:7.1f} TFLOP/s = {fl / ms / 1e9 / 2500:.3f} of the bf16 peak", flush=True)
    pm.close()

s25 = synth.surf_like_sets(27, 4096, pool=16384, seed_base=1000)
allp = synth.all_pairs(27)
run("M-SURF-4k (300 pairs)", s25[:25], synth.all_pairs(25))
run("256 pairs: 4 full rounds", s25, allp[:256])
run("320 pairs: 5 full rounds", s25, allp[:320])
run("640 pairs (the list twice: 10 full rounds)", s25, np.concatenate([allp[:320], allp[:320]]))
run("64 pairs: one full round", s25, allp[:64])
run("38 pairs: one rank's share of the 300 at N = 8 (0.59 round)", s25, allp[:38])
# 4096-row query sets against 8192-row train sets: twice the main loop per block set-up, the same blocks
big = synth.surf_like_sets(13, 8192, pool=65536, seed_base=2000)
mixed = big + s25[:25]
pq = np.array([(13 + (k % 25), k % 13) for k in range(320)], np.int32)
run("320 pairs 4096 q x 8192 t (5 full rounds, 2 x loop per block)", mixed, pq)
run("300 pairs 4096 q x 8192 t", mixed, pq[:300])
pb = synth.all_pairs(13)
run("78 pairs 8192 x 8192 (config 4's shape)", big, pb)
run("80 pairs 8192 x 8192 (the list + 2: 2.5 rounds)", big, np.concatenate([pb, pb[:2]]))

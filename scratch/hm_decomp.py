# hipEvent kernel time of hamming_fp4_kernel on M-ORB-4k (for the timing-only variants: ESFM_LIB=scratch/variants/libesfm_hm_*.so)
import sys; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
pairs = synth.all_pairs(25)
sets = synth.orb_like_sets(25, 4096, pool=16384, seed_base=3000)
pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_HAMMING), pairs)
out = []
for rep in range(2):
    for _ in range(3): pm.match(0.8)
    pm.ctx.synchronize(); pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_HAMMING_KNN)
    for _ in range(20): pm.match(0.8)
    pm.ctx.synchronize(); k = pm.ctx.kernel_time(_lib.K_HAMMING_KNN); pm.ctx.set_kernel_timing(False)
    out.append(k[0] / max(k[1], 1))
print("  ".join(f"{x:.4f}" for x in out))

# L2 matcher kernel time on M-SURF-4k (300 pairs of 4096 x 4096 x 64 f32)
import sys, time; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
pairs = synth.all_pairs(25)
sets = synth.surf_like_sets(25, 4096, pool=16384, seed_base=1000)
bank = E.DescriptorBank(sets, E.ESFM_L2_F32)
pm = E.PairMatcher(bank, pairs)
for _ in range(3): pm.match(0.5)
pm.ctx.synchronize(); pm.ctx.set_kernel_timing(True)
t = time.perf_counter()
for _ in range(10): pm.match(0.5)
pm.ctx.synchronize(); el = time.perf_counter() - t
ms, n = pm.ctx.kernel_time(_lib.K_L2_KNN); rs, rn = pm.ctx.kernel_time(_lib.K_L2_RESCAN)
try:
    ss, sn = pm.ctx.kernel_time(_lib.K_L2_SECOND); sp = pm.second_pass()
except Exception:
    ss, sn, sp = 0.0, 0, -1
print('pairs/s', 3000 / el, 'knn ms', ms / n, 'second ms', ss / max(sn, 1), 'rescan ms', rs / max(rn, 1), 'stats', pm.stats(), 'second-pass queries', sp)

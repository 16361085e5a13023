# re-scan time on a config-4-like launch (70 images x 8192 features: 2415 pairs, the large-launch path)
import sys, time; sys.path.insert(0,'.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
sets = synth.surf_like_sets(70, 8192, pool=65536, seed_base=2000)
pairs = synth.all_pairs(70)
bank = E.DescriptorBank(sets, E.ESFM_L2_F32)
pm = E.PairMatcher(bank, pairs)
pm.match(0.5); pm.ctx.synchronize()
pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_L2_KNN); pm.ctx.kernel_time(_lib.K_L2_RESCAN)
t0 = time.perf_counter()
for _ in range(3): pm.match(0.5)
pm.ctx.synchronize(); el = (time.perf_counter() - t0) / 3
k, kn = pm.ctx.kernel_time(_lib.K_L2_KNN); r, rn = pm.ctx.kernel_time(_lib.K_L2_RESCAN)
print('step ms', el * 1e3, 'knn ms', k / kn, 'rescan ms', r / rn, 'stats', pm.stats())

# config 4 (M-SURF-8k, 256 images, all 32 640 pairs) alone: pairs/s and the pass's kernel time (ESFM_X1_GRID picks the grid)
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, easysfm_amd as E
from easysfm_amd import synth, _lib
sets = synth.surf_like_sets(256, 8192, pool=65536, seed_base=2000)
pairs = E.shard_pair_list(256, np.full(256, 8192, np.int32), 0, 1)
pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), pairs)
pm.match(0.5); pm.ctx.synchronize()
pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_L2_KNN)
t = time.perf_counter()
for _ in range(2): pm.match(0.5)
pm.ctx.synchronize(); el = time.perf_counter() - t
ms, n = pm.ctx.kernel_time(_lib.K_L2_KNN); fs, fn = pm.ctx.kernel_time(_lib.K_L2_SECOND)
print('pairs/s %.0f  pass %.2f ms  finish %.2f ms  step %.2f ms' % (2 * len(pairs) / el, ms / n, fs / max(fn, 1), el / 2 * 1e3))

# M-ORB-4k (25 x 4096 x 256 bit, 300 pairs): pairs/s and the Hamming kernel's time
import sys, time; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
pm = E.PairMatcher(E.DescriptorBank(synth.orb_like_sets(25, 4096, pool=16384, seed_base=3000), E.ESFM_HAMMING), synth.all_pairs(25))
for _ in range(3): pm.match(0.8)
pm.ctx.synchronize(); pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_HAMMING_KNN)
t = time.perf_counter()
for _ in range(10): pm.match(0.8)
pm.ctx.synchronize(); el = time.perf_counter() - t
ms, n = pm.ctx.kernel_time(_lib.K_HAMMING_KNN)
print('pairs/s %.0f  hamming kernel %.4f ms  step %.4f ms' % (3000 / el, ms / n, el / 10 * 1e3))

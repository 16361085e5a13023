import sys, time; sys.path.insert(0,'.')
import numpy as np, torch, easysfm_amd as E
from easysfm_amd import synth, _lib
sets = synth.orb_like_sets(25, 4096, pool=16384, seed_base=3000)
pairs = synth.all_pairs(25)
bank = E.DescriptorBank(sets, E.ESFM_HAMMING)
pm = E.PairMatcher(bank, pairs)
for _ in range(2): pm.match(0.8)
pm.ctx.synchronize(); pm.ctx.set_kernel_timing(True)
t=time.perf_counter()
for _ in range(10): pm.match(0.8)
pm.ctx.synchronize(); el=time.perf_counter()-t
ms,n = pm.ctx.kernel_time(_lib.K_HAMMING_KNN)
print('pairs/s', 300*10/el, 'kernel ms', ms/n, 'lane-ops/s', 2*4096*4096*8*300/(ms/n*1e-3)/1e12, 'T (peak 78.6)')

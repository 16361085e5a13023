import sys, time; sys.path.insert(0,'.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
osets = synth.orb_like_sets(25, 4096, pool=16384, seed_base=3000)
opm = E.PairMatcher(E.DescriptorBank(osets, E.ESFM_HAMMING), synth.all_pairs(25))
for _ in range(2): opm.match(0.8)
opm.ctx.synchronize(); opm.ctx.set_kernel_timing(True); opm.ctx.kernel_time(_lib.K_HAMMING_KNN)
t0 = time.perf_counter()
for _ in range(10): opm.match(0.8)
opm.ctx.synchronize(); el = (time.perf_counter() - t0) / 10
ms, n = opm.ctx.kernel_time(_lib.K_HAMMING_KNN)
print('orb step ms', el * 1e3, 'kernel ms', ms / n, 'pairs/s', 300 / el)

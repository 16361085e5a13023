# per-wave stage times of l2_knn_bf16x1_kernel (build with -DESFM_X1_TRACE, ESFM_LIB=...): s_memrealtime ticks of 10 ns
import sys; sys.path.insert(0, '.')
import ctypes as C, numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
pairs = synth.all_pairs(25)
sets = synth.surf_like_sets(25, 4096, pool=16384, seed_base=1000)
pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), pairs)
for _ in range(3): pm.match(0.5)
pm.ctx.synchronize()
out = (C.c_int32 * 16)()
_lib.check(_lib.lib().esfm_match_debug_counters(pm.ctx.handle, out))
c = list(out); w = max(c[11], 1)
print('waves', c[11])
n = 2400 * 4   # items x waves
print('us per item and wave: set-up wait %.2f  main loop %.2f  tail + next set-up %.2f' % (c[8] / n / 100, c[9] / n / 100, c[10] / n / 100))
print('first wave start -> last wave end %.1f us; longest wave %.1f us' % ((c[12] - (0x40000000 - c[13])) / 100, c[14] / 100))
print('s_memtime ticks per 10-ns tick inside the main loop: %.3f' % (c[15] * 256.0 / c[9]))

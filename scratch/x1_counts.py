# survivors / second-pass queries of the L2 matcher on M-SURF-4k under a build (ESFM_LIB=...)
import sys; sys.path.insert(0, '.')
import ctypes as C, numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
pm = E.PairMatcher(E.DescriptorBank(synth.surf_like_sets(25, 4096, pool=16384, seed_base=1000), E.ESFM_L2_F32), synth.all_pairs(25))
for _ in range(2): pm.match(0.5)
pm.ctx.synchronize()
out = (C.c_int32 * 16)()
_lib.check(_lib.lib().esfm_match_debug_counters(pm.ctx.handle, out))
c = list(out)
print('survivors', c[2], 'uncertified / undecided after the re-rank', c[1], 'second pass', pm.second_pass())

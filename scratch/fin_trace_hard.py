# stage-1 (re-rank) times of l2_finish_kernel per wave on M-SURF-4k-hard (build with -DESFM_FIN_TRACE, ESFM_LIB=...); also the kernel with stages compiled out
import sys, os, time; sys.path.insert(0, '.')
import ctypes as C, numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
ctx0 = E.Context(0, None)
imgs = np.load("tests/golden/fountain11_gray.npz")["images"]
pool = np.concatenate([E.surf_detect_and_compute(im, 300.0, None, ctx0)[1] for im in imgs])
pairs = synth.all_pairs(25)
for name, sets in (("benign", synth.surf_like_sets(25, 4096, pool=16384, seed_base=1000)), ("hard", synth.msurf4k_hard_sets(pool))):
    pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), pairs)
    for _ in range(3): pm.match(0.5)
    pm.ctx.synchronize()
    pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_L2_SECOND)
    for _ in range(10): pm.match(0.5)
    pm.ctx.synchronize()
    f = pm.ctx.kernel_time(_lib.K_L2_SECOND); pm.ctx.set_kernel_timing(False)
    out = (C.c_int32 * 16)()
    _lib.check(_lib.lib().esfm_match_debug_counters(pm.ctx.handle, out))
    c = list(out)
    print(f"{name}: finish kernel {f[0] / max(f[1], 1):.3f} ms; counters {c}")
    if c[10]:
        print('  survivors', c[2], 'waves', c[10], 'virtual sets', c[9], 'rounds', c[11], 'rounds/set %.2f' % (c[11] / max(c[9], 1)))
        print('  stage 1 per wave %.2f us, per virtual set %.2f us' % (c[8] / max(c[10], 1) / 100, c[8] / max(c[9], 1) / 100))
        print('  per round: transfer wait %.2f us, distance + reduce %.2f us' % (c[12] / max(c[11], 1) / 100, c[13] / max(c[11], 1) / 100))
        print('  per wave: kernel entry -> entries landed %.2f us; stores acknowledged + barrier + arrival atomic %.2f us' % (c[14] / max(c[10], 1) / 100, c[15] / max(c[10], 1) / 100))
    pm.close()

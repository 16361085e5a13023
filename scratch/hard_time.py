"""finish / pass kernel times on M-SURF-4k (benign) and M-SURF-4k-hard for the library in ESFM_LIB (timing only: no verification)."""
import sys, os, time; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
ctx0 = E.Context(0, None)
imgs = np.load("tests/golden/fountain11_gray.npz")["images"]
pool = np.concatenate([E.surf_detect_and_compute(im, 300.0, None, ctx0)[1] for im in imgs])
pairs = synth.all_pairs(25)
for name, sets in (("benign", synth.surf_like_sets(25, 4096, pool=16384, seed_base=1000)), ("hard", synth.msurf4k_hard_sets(pool))):
    pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), pairs)
    for _ in range(30): pm.match(0.5)
    pm.ctx.synchronize()
    pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_L2_SECOND); pm.ctx.kernel_time(_lib.K_L2_KNN)
    t0 = time.perf_counter()
    for _ in range(40): pm.match(0.5)
    pm.ctx.synchronize()
    el = (time.perf_counter() - t0) / 40
    f = pm.ctx.kernel_time(_lib.K_L2_SECOND); k = pm.ctx.kernel_time(_lib.K_L2_KNN); pm.ctx.set_kernel_timing(False)
    print(f"{os.environ.get('ESFM_LIB', 'in-tree')[-28:]:28s} {name:7s} pass {k[0] / max(k[1], 1):.3f} ms  finish {f[0] / max(f[1], 1):.3f} ms  step {el * 1e3:.3f} ms (with event timers)")
    pm.close()

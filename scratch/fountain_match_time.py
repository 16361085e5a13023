# matcher kernel times on the reference's own fountain descriptors (11 images at 768 x 512, SURF minHessian 300, all 55 pairs)
import os, sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
imgs = np.load('/root/repo/tests/golden/fountain11_gray.npz')['images']
ctx = E.Context(0)
sets = [E.surf_detect_and_compute(imgs[k], 300.0, None, ctx)[1] for k in range(len(imgs))]
print('features per image', [len(s) for s in sets])
pairs = synth.all_pairs(len(sets))
bank = E.DescriptorBank(sets, E.ESFM_L2_F32)
pm = E.PairMatcher(bank, pairs)
for _ in range(3): pm.match(0.5)
pm.ctx.synchronize(); pm.ctx.set_kernel_timing(True)
for k in (_lib.K_L2_KNN, _lib.K_L2_SECOND, _lib.K_L2_RESCAN): pm.ctx.kernel_time(k)
t = time.perf_counter()
for _ in range(10): pm.match(0.5)
pm.ctx.synchronize(); el = time.perf_counter() - t
ms, n = pm.ctx.kernel_time(_lib.K_L2_KNN); ss, sn = pm.ctx.kernel_time(_lib.K_L2_SECOND); rs, rn = pm.ctx.kernel_time(_lib.K_L2_RESCAN)
print('step ms', el / 10 * 1e3, 'knn ms', ms / n, 'second ms', ss / max(sn, 1), 'rescan ms', rs / max(rn, 1), 'stats', pm.stats(), 'second-pass queries', pm.second_pass())

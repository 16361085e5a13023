# the bench's SOR cloud (BA-25 points + 2 % far points): kernel time
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
rng = np.random.default_rng(4100)
cloud = np.concatenate([synth.ba_scene(25, 30000, 8, radius=10.0, extent=2.0, seed=4000).pts_gt, rng.uniform(-12, 12, (600, 3))]).astype(np.float32)
ctx = E.Context(0)
E.sor_filter(cloud, 50, 2.0, ctx)
ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_SOR_KNN)
for _ in range(10): keep, md, thr = E.sor_filter(cloud, 50, 2.0, ctx)
ms, cnt = ctx.kernel_time(_lib.K_SOR_KNN)
print('bench cloud', len(cloud), 'points: SOR kernels %.3f ms' % (ms / cnt), 'kept', int(keep.sum()), 'thr', thr)

# SURF detect + describe on the fountain images (768 x 512, minHessian 300): ms per image and the describe kernel's share
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, easysfm_amd as E
from easysfm_amd import _lib
imgs = np.load('/root/repo/tests/golden/fountain11_gray.npz')['images']
ctx = E.Context(0)
for k in range(3): E.surf_detect_and_compute(imgs[k], 300.0, None, ctx)
ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_SURF_DESC)
t = time.perf_counter(); n = 0
for rep in range(3):
    for k in range(len(imgs)): kp, d = E.surf_detect_and_compute(imgs[k], 300.0, None, ctx); n += 1
el = time.perf_counter() - t
ms, cnt = ctx.kernel_time(_lib.K_SURF_DESC)
print(f'{n} images: {el / n * 1e3:.3f} ms per image, describe kernels {ms / cnt:.3f} ms per image, last image {len(kp)} keypoints')

import sys, time; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import _lib
imgs = np.load("tests/golden/fountain11_gray.npz")["images"]
ctx = E.Context(0, None)
for n in (8000, 2000):
    kp, d = E.orb_detect_and_compute(imgs[0], n, None, ctx)
    ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_ORB_FAST)
    t0 = time.perf_counter()
    for k in range(10): kp, d = E.orb_detect_and_compute(imgs[k % 11], n, None, ctx)
    el = (time.perf_counter() - t0) / 10
    f = ctx.kernel_time(_lib.K_ORB_FAST); ctx.set_kernel_timing(False)
    print(f"ORB {n}: {el * 1e3:.2f} ms per image ({len(kp)} keypoints), timed kernel group {f[0] / max(f[1], 1):.3f} ms x {f[1] // 10} per image")
for k in range(3):
    t0 = time.perf_counter(); kp, d = E.surf_detect_and_compute(imgs[k], 300.0, None, ctx); print(f"SURF image {k}: {(time.perf_counter() - t0) * 1e3:.2f} ms, {len(kp)} kp")

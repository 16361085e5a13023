"""SURF detect + describe at the reference's ORIGINAL image size (fountain-P11: 3072 x 2048): a fountain image up-sampled 4 x with bicubic-ish
smoothing + noise; per-call wall time and the describe / det-trace kernels' event times."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import easysfm_amd as E
from easysfm_amd import _lib
z = np.load(os.path.join(ROOT, "tests", "golden", "fountain11_gray.npz"))["images"][2].astype(np.float32)       # 512 x 768
big = np.kron(z, np.ones((4, 4), np.float32))
k = np.array([1, 4, 6, 4, 1], np.float32) / 16
for ax in (0, 1):
    big = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, big)
big = np.clip(big + np.random.default_rng(1).normal(0, 2.0, big.shape), 0, 255).astype(np.uint8)
print("image", big.shape)
ctx = E.Context(0, None)
for thr in (300.0,):
    kp, d = E.surf_detect_and_compute(big, thr, None, ctx)
    win = (21 * (kp[:, 2] * np.float32(1.2) / np.float32(9.0))).astype(np.int64)
    print("keypoints", len(kp), "window pct", np.percentile(win, [50, 90, 99, 100]).astype(int), "sum w^2", int((win * win).sum()))
    ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_SURF_DESC); ctx.kernel_time(_lib.K_SURF_DET)
    t0 = time.perf_counter()
    for _ in range(5): E.surf_detect_and_compute(big, thr, None, ctx)
    el = (time.perf_counter() - t0) / 5
    a, an = ctx.kernel_time(_lib.K_SURF_DESC); b, bn = ctx.kernel_time(_lib.K_SURF_DET)
    ctx.set_kernel_timing(False)
    print(f"minHessian {thr}: {el*1e3:.2f} ms per image; describe {a/max(an,1):.3f} ms; det/trace {b/max(bn,1):.3f} ms")
okp, od = E.orb_detect_and_compute(big, 8000, None, ctx)
t0 = time.perf_counter()
for _ in range(5): E.orb_detect_and_compute(big, 8000, None, ctx)
print(f"ORB 8000: {(time.perf_counter()-t0)/5*1e3:.2f} ms per image, {len(okp)} keypoints")

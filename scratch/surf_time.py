"""SURF detect + describe on the bench leg's image: per-call wall time, the describe / det-trace kernels' event times, window-size
histogram, and the descriptors / keypoints against the oracle (bit-exact)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import easysfm_amd as E
from easysfm_amd import _lib
import oracle

gold = os.path.join(ROOT, "tests", "golden", "fountain11_half_gray.npz")
half = np.load(gold)["images"][0]
img = np.ascontiguousarray(np.kron(half, np.ones((2, 2), np.uint8)))
img = np.clip(img.astype(np.int16) + np.random.default_rng(4300).integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
ctx = E.Context(0, None)
kp, d = E.surf_detect_and_compute(img, 300.0, None, ctx)
win = ((20 + 1) * (kp[:, 2] * np.float32(1.2) / np.float32(9.0))).astype(np.int64)
print("keypoints", len(kp), "window sizes: pct", np.percentile(win, [0, 25, 50, 75, 90, 99, 100]).astype(int), "n > 72:", int((win > 72).sum()))
for _ in range(5):
    E.surf_detect_and_compute(img, 300.0, None, ctx)
ctx.set_kernel_timing(True)
ctx.kernel_time(_lib.K_SURF_DESC); ctx.kernel_time(_lib.K_SURF_DET)
t0 = time.perf_counter()
N = 20
for _ in range(N):
    E.surf_detect_and_compute(img, 300.0, None, ctx)
el = (time.perf_counter() - t0) / N
a, an = ctx.kernel_time(_lib.K_SURF_DESC); b, bn = ctx.kernel_time(_lib.K_SURF_DET)
print(f"per image {el*1e3:.3f} ms; describe {a/max(an,1):.4f} ms; det/trace {b/max(bn,1):.4f} ms")
rk, rd = oracle.surf(img, 300.0)
print("parity: keypoints", bool(np.array_equal(kp, rk)), "descriptors", bool(np.array_equal(d.view(np.uint32), rd.view(np.uint32))))

# SURF at the reference cameras' full resolution (2048 x 3072): timing and parity against the oracle
import sys, time; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E, oracle
z = np.load('tests/golden/fountain11_half_gray.npz')
half = z['images'][0]                       # 256 x 384
from numpy import kron
img = np.kron(half, np.ones((8, 8), np.uint8))
rng = np.random.default_rng(0)
img = np.clip(img.astype(np.int16) + rng.integers(-10, 11, img.shape), 0, 255).astype(np.uint8)
# smooth a little so that it is not blocky: box blur via integral trick (cheap)
k = 5
c = np.cumsum(np.cumsum(np.pad(img.astype(np.float64), ((k, k), (k, k)), mode='edge'), 0), 1)
sm = (c[2*k:, 2*k:] - c[:-2*k, 2*k:] - c[2*k:, :-2*k] + c[:-2*k, :-2*k]) / (4 * k * k)
img = np.ascontiguousarray(np.clip(sm[:2048, :3072], 0, 255).astype(np.uint8))
print(img.shape)
ctx = E.Context(0, None)
kp, d = E.surf_detect_and_compute(img, 300.0, None, ctx)
t = time.perf_counter()
for _ in range(3): kp, d = E.surf_detect_and_compute(img, 300.0, None, ctx)
print('GPU', (time.perf_counter() - t) / 3 * 1e3, 'ms per image,', len(kp), 'keypoints')
t = time.perf_counter(); rk, rd = oracle.surf(img, 300.0, max_kp=400000); print('oracle', time.perf_counter() - t, 's', len(rk))
print('bit-exact', np.array_equal(kp.view(np.uint32), rk.view(np.uint32)) and np.array_equal(d.view(np.uint32), rd.view(np.uint32)))

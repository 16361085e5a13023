# train sets beyond the one-product pass's 65 536 rows: which fallback, how fast ("retire or re-profile the three-product fallback")
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import _lib
rng = np.random.default_rng(1)
nt, nq = 100000, 4096
t = rng.standard_normal((nt, 64)).astype(np.float32); t /= np.linalg.norm(t, axis=1, keepdims=True)
q = rng.standard_normal((nq, 64)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
q[:1000] = t[rng.choice(nt, 1000, replace=False)] + 0.03 * rng.standard_normal((1000, 64)).astype(np.float32)
pm = E.PairMatcher(E.DescriptorBank([t, q], E.ESFM_L2_F32), np.array([[1, 0]], np.int32))
for _ in range(2): pm.match(0.5)
pm.ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5): r = pm.match(0.5)
pm.ctx.synchronize()
el = (time.perf_counter() - t0) / 5
fl = 2.0 * nq * nt * 64
print(f"ESFM_L2_PASS={os.environ.get('ESFM_L2_PASS', '(default)')}: {nq} x {nt} pair {el * 1e3:.3f} ms = {fl / el / 1e12:.1f} TFLOP/s algorithmic, {int(r.n_out.sum().item())} matches, rescans {pm.stats()[1]}")

"""PnP-RANSAC (bench.py's pnp leg) alone: wall time per call; run under rocprofv3 --kernel-trace for the kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easysfm_amd as E
from easysfm_amd import synth
rng = np.random.default_rng(4500)
K4p = np.array(synth.FOUNTAIN_K4, np.float32)
n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
Rt = synth.aa_to_R(rng.normal(0, 0.2, 3)); tt_ = np.array([0.3, -0.1, 0.5])
X3 = (rng.uniform(-2, 2, (n3, 3)) + np.array([0, 0, 8.0])).astype(np.float32)
Xc = X3.astype(np.float64) @ Rt.T + tt_
pix = np.stack([Xc[:, 0] / Xc[:, 2] * K4p[0] + K4p[1], Xc[:, 1] / Xc[:, 2] * K4p[2] + K4p[3]], 1) + rng.normal(0, 0.5, (n3, 2))
bad = rng.choice(n3, n3 // 4, replace=False); pix[bad] += rng.uniform(-80, 80, (len(bad), 2))
pix = pix.astype(np.float32)
ctx = E.Context(0, None)
E.solve_pnp_ransac(X3, pix, K4p, 100, 8.0, 0.99, ctx)
t0 = time.perf_counter()
for _ in range(20):
    rv_, tv_, R_, m_, it_p = E.solve_pnp_ransac(X3, pix, K4p, 100, 8.0, 0.99, ctx)
el = (time.perf_counter() - t0) / 20
print(f"pnp n={n3}: {el * 1e3:.3f} ms per call, {it_p} iterations, {int(m_.sum())} inliers")

# run-to-run consistency of the RANSAC path (essential matrices, masks, iteration counts) over repeated calls
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth
K4 = np.array(synth.FOUNTAIN_K4, np.float32)
rng = np.random.default_rng(5)
def pair(n, frac):
    R = synth.aa_to_R(rng.normal(0, 0.15, 3)); t = np.array([1.0, 0.1, -0.05]) + rng.normal(0, 0.05, 3); t /= np.linalg.norm(t)
    X = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0, 8.0]); x1 = X[:, :2] / X[:, 2:3]; Xc = X @ R.T + t; x2 = Xc[:, :2] / Xc[:, 2:3]
    p1 = (x1 * [K4[0], K4[2]] + [K4[1], K4[3]]).astype(np.float32); p2 = (x2 * [K4[0], K4[2]] + [K4[1], K4[3]] + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    out = rng.choice(n, int(frac * n), replace=False); p2[out] += rng.uniform(-60, 60, (len(out), 2)).astype(np.float32)
    return p1, p2
ctx = E.Context(0, None)
for n_pairs, n, frac in ((200, 800, 0.4), (37, 300, 0.6)):
    jobs = [pair(n, frac) for _ in range(n_pairs)]
    off = np.arange(n_pairs + 1, dtype=np.int32) * n
    a = np.concatenate([j[0] for j in jobs]); b = np.concatenate([j[1] for j in jobs]); Ks = np.tile(K4, (n_pairs, 1))
    ref = None; diff = 0
    for r in range(25):
        Es, mask, st, it = E.find_essential_pairs(off, a, b, Ks, 0.99, 1.0, ctx)
        sig = (Es.tobytes(), mask.tobytes(), st.tobytes(), it.tobytes())
        if ref is None: ref = sig
        diff += sig != ref
    print(f'{n_pairs} pairs x {n} matches, {frac:.0%} outliers: 25 calls, differing {diff}, mean iterations {it.mean():.1f}')

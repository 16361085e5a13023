# histogram of Durand-Kerner sweeps per hypothesis (variant built with -DESFM_DK_HIST; ESFM_LIB=scratch/variants/libesfm_dkhist.so)
import sys, ctypes; sys.path.insert(0, '/root/repo')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
K4 = np.array(synth.FOUNTAIN_K4, np.float32)
def pair(rng, n, frac):
    R = synth.aa_to_R(rng.normal(0, 0.15, 3)); t = np.array([1.0, 0.1, -0.05]) + rng.normal(0, 0.05, 3); t /= np.linalg.norm(t)
    X = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0, 8.0]); x1 = X[:, :2] / X[:, 2:3]; Xc = X @ R.T + t; x2 = Xc[:, :2] / Xc[:, 2:3]
    p1 = (x1 * [K4[0], K4[2]] + [K4[1], K4[3]]).astype(np.float32); p2 = (x2 * [K4[0], K4[2]] + [K4[1], K4[3]] + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    out = rng.choice(n, int(frac * n), replace=False); p2[out] += rng.uniform(-60, 60, (len(out), 2)).astype(np.float32)
    return p1, p2
rng = np.random.default_rng(0); ctx = E.Context(0, None)
n_pairs, n, frac = 300, 1000, 0.3
jobs = [pair(rng, n, frac) for _ in range(n_pairs)]
off = np.arange(n_pairs + 1, dtype=np.int32) * n
a = np.concatenate([j[0] for j in jobs]); b = np.concatenate([j[1] for j in jobs]); Ks = np.tile(K4, (n_pairs, 1))
E.find_essential_pairs(off, a, b, Ks, 0.99, 1.0, ctx)
buf = (ctypes.c_uint * 301)(); _lib.lib().esfm_debug_dk_hist(buf)
h = np.frombuffer(buf, dtype=np.uint32).astype(np.int64); tot = h.sum()
print('hypotheses', tot, ' mean sweeps %.1f' % ((h * np.arange(301)).sum() / tot))
cum = np.cumsum(h) / tot
for q in (10, 15, 20, 25, 30, 40, 50, 60, 80, 100, 150, 200, 250, 299, 300):
    print(f'  <= {q:3d} sweeps: {100 * cum[q]:7.3f} %')

import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import easysfm_amd as E
from easysfm_amd import synth, _lib
import oracle
K4 = np.array(synth.FOUNTAIN_K4, np.float32)
def pair(rng, n, frac):
    R = synth.aa_to_R(rng.normal(0, 0.15, 3)); t = np.array([1.0, 0.1, -0.05]) + rng.normal(0, 0.05, 3); t /= np.linalg.norm(t)
    X = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0, 8.0]); x1 = X[:, :2] / X[:, 2:3]; Xc = X @ R.T + t; x2 = Xc[:, :2] / Xc[:, 2:3]
    p1 = (x1 * [K4[0], K4[2]] + [K4[1], K4[3]]).astype(np.float32); p2 = (x2 * [K4[0], K4[2]] + [K4[1], K4[3]] + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    out = rng.choice(n, int(frac * n), replace=False); p2[out] += rng.uniform(-60, 60, (len(out), 2)).astype(np.float32)
    return p1, p2
rng = np.random.default_rng(0)
ctx = E.Context(0, None)
for n_pairs, n, frac in ((300, 1000, 0.3), (300, 1000, 0.6), (300, 200, 0.3)):
    jobs = [pair(rng, n, frac) for _ in range(n_pairs)]
    off = np.arange(n_pairs + 1, dtype=np.int32) * n
    a = np.concatenate([j[0] for j in jobs]); b = np.concatenate([j[1] for j in jobs]); Ks = np.tile(K4, (n_pairs, 1))
    E.find_essential_pairs(off, a, b, Ks, 0.99, 1.0, ctx)
    ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_RANSAC)
    t0 = time.perf_counter(); Es, mask, st, it = E.find_essential_pairs(off, a, b, Ks, 0.99, 1.0, ctx); el = time.perf_counter() - t0
    ms, cnt = ctx.kernel_time(_lib.K_RANSAC); ctx.set_kernel_timing(False)
    t0 = time.perf_counter()
    for j in jobs[:40]: oracle.find_essential_ransac(j[0], j[1], K4, 0.99, 1.0)
    cpu = (time.perf_counter() - t0) / 40
    print(f"pairs={n_pairs} n={n} outliers={frac}: GPU batch {el*1e3:.1f} ms ({el/n_pairs*1e6:.0f} us/pair, kernels {ms:.1f} ms in {cnt} rounds), mean iters {it.mean():.0f} max {it.max()}, CPU oracle {cpu*1e6:.0f} us/pair (1 core)")

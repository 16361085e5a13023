"""M-SURF-4k-hard: what the matcher's screen / certificate do on resampled real SURF descriptors, and what a step costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import easysfm_amd as E
from easysfm_amd import synth, _lib

ctx0 = E.Context(0, None)
imgs = np.load("tests/golden/fountain11_gray.npz")["images"]
pool = np.concatenate([E.surf_detect_and_compute(im, 300.0, None, ctx0)[1] for im in imgs])
print("pool", pool.shape, flush=True)
pairs = synth.all_pairs(25)

def run(name, sets, ratio):
    bank = E.DescriptorBank(sets, E.ESFM_L2_F32)
    pm = E.PairMatcher(bank, pairs)
    for _ in range(3): pm.match(ratio)
    pm.ctx.synchronize()
    pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_L2_KNN); pm.ctx.kernel_time(_lib.K_L2_SECOND)
    t0 = time.perf_counter()
    for _ in range(50): res = pm.match(ratio)
    pm.ctx.synchronize(); el = (time.perf_counter() - t0) / 50
    k = pm.ctx.kernel_time(_lib.K_L2_KNN); f = pm.ctx.kernel_time(_lib.K_L2_SECOND)
    pm.ctx.set_kernel_timing(False)
    nq, nres = pm.stats(); nsec = pm.second_pass()
    nm = int(res.n_out.sum().item())
    pm.set_l2_audit(4); pm.match(ratio); pm.ctx.synchronize(); nrej = len(pm.flagged()); pm.set_l2_audit(0)
    print(f"{name:34s} ratio {ratio}: {300 / el:9.0f} pairs/s ({el * 1e3:.3f} ms/step; pass {k[0] / max(k[1], 1):.3f} finish {f[0] / max(f[1], 1):.3f} ms), "
          f"screen keeps {100.0 * (nq - nrej) / nq:5.2f} %, second pass {100.0 * nsec / nq:5.2f} % ({nsec}), re-scan {nres}, matches {nm}", flush=True)
    pm.close()

benign = synth.surf_like_sets(25, 4096, pool=16384, seed_base=1000)
for r in (0.5,): run("M-SURF-4k (benign)", benign, r)
for tn, fn, tf, sub in ((0.005, 0.01, 0.5, None), (0.003, 0.006, 0.5, None), (0.002, 0.004, 0.5, None), (0.005, 0.01, 0.5, 8192), (0.005, 0.01, 0.25, None), (0.008, 0.008, 0.5, None)):
    pl = pool if sub is None else pool[:sub]
    hard = synth.surf_resampled_sets(pl, 25, 4096, seed_base=6000, track_noise=tn, fresh_noise=fn, track_frac=tf)
    for r in (0.5, 0.8): run(f"hard (tn {tn}, fn {fn}, tf {tf}, pool {len(pl)})", hard, r)

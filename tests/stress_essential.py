"""Randomised stress of the 2-D/2-D verification (5-point RANSAC: esfm_find_essential_pairs) against the oracle: 8 ... 3000 matches,
0 - 60 % gross outliers, thresholds 0.5 - 3 px, confidence 0.99 / 0.999, planar and general scenes, small and large baselines -- iteration
counts, inlier masks AND the essential matrix bit for bit (one arithmetic on both sides since round 6: five_point_core.hpp).
Differences are COUNTED by kind and reported, as tests/stress_pnp.py does; any difference is a non-zero exit status.
usage: python tests/stress_essential.py [--seconds S | --cases N] [--seed K]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easysfm_amd as E
from easysfm_amd import synth
import oracle

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=None)
ap.add_argument("--cases", type=int, default=None)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
budget = args.seconds if args.seconds is not None else (1e9 if args.cases is not None else 60.0)
max_cases = args.cases if args.cases is not None else 1 << 60
rng = np.random.default_rng(args.seed)
oracle.build()
ctx = E.Context(0)
K4 = np.array(synth.FOUNTAIN_K4, np.float32)
t_end = time.time() + budget
n_cases = n_exact = n_border = n_far = n_one_side = n_none = 0
border = []
iters_total = 0
while time.time() < t_end and n_cases < max_cases:
    n = int(rng.choice([8, 12, 25, 60, 200, 600, 1500, 3000]))
    frac = float(rng.choice([0.0, 0.1, 0.3, 0.45, 0.6]))
    thr = float(rng.choice([0.5, 1.0, 3.0]))
    prob = float(rng.choice([0.99, 0.999]))
    R = synth.aa_to_R(rng.normal(0, rng.choice([0.02, 0.15, 0.4]), 3)); t = rng.normal(0, 1, 3); t /= np.linalg.norm(t); t *= rng.choice([0.1, 1.0])
    X = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0, 8.0])
    if rng.random() < 0.15: X[:, 2] = 8.0
    Xc = X @ R.T + t
    a = (X[:, :2] / X[:, 2:3] * [K4[0], K4[2]] + [K4[1], K4[3]] + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    b = (Xc[:, :2] / Xc[:, 2:3] * [K4[0], K4[2]] + [K4[1], K4[3]] + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    bad = rng.choice(n, int(frac * n), replace=False)
    b[bad] += rng.uniform(-60, 60, (len(bad), 2)).astype(np.float32)
    ok, Er, mr, itr, cnt = oracle.find_essential_ransac(a, b, K4, prob, thr)
    Es, mask, status, its = E.find_essential_pairs(np.array([0, n], np.int32), a, b, K4[None], prob, thr, ctx)
    got, mg, itg = bool(status[0]), np.asarray(mask[:n]).astype(bool), int(its[0])
    tag = (args.seed, n_cases, n, frac, thr, prob)
    n_cases += 1
    if got != ok:
        n_one_side += 1; border.append(tag + ("model on one side only", got, ok)); continue
    if not ok:
        n_none += 1; continue
    iters_total += itr
    if itg == itr and np.array_equal(mg, mr) and np.array_equal(np.asarray(Es[0]), Er):
        n_exact += 1
    else:
        n_border += 1
        if abs(int(mg.sum()) - int(mr.sum())) > max(2, 0.02 * n): n_far += 1
        border.append(tag + (itg, itr, int(mg.sum()), int(mr.sum())))
print(f"stress_essential seed {args.seed}: {n_cases} cases ({n_none} without a model on both sides), {iters_total} RANSAC iterations: {n_exact} with iteration count and mask "
      f"and matrix equal to the oracle's, {n_border} where the two sides chose differently ({n_far} of them with inlier counts more than 2 % apart), {n_one_side} with a model on one side only")
for x in border[:12]:
    print("  differs:", x)
sys.exit(0 if n_border == 0 and n_one_side == 0 else 1)

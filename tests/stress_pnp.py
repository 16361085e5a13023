"""Randomised stress of the PnP-RANSAC path against the oracle: problem sizes 6 ... 4000, 0 - 70 % gross outliers, coplanar sets, thresholds
1 - 8 px, iteration caps 50 ... 50 000 -- iteration counts, inlier masks and (round 6: host re-fit up to 8 192 inliers) poses bit for bit.  What differs is COUNTED and reported,
by kind (another of two nearly equal hypotheses chosen; a pose on one side only; an ill-conditioned re-fit whose poses drift apart), so
that a regression shows up as a rate: round 5's first run reported 3.7 % / 0.07 % / 1 % and led to the shared Jacobi stopping rule,
the centred sums of the hypothesis solver and the host re-fit of small inlier sets; since then all three are 0.
usage: python tests/stress_pnp.py [--seconds S | --cases N] [--seed K]"""
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
ap.add_argument("--only", type=int, default=None, help="draw every case as usual but run only this one, verbosely")
args = ap.parse_args()
budget = args.seconds if args.seconds is not None else (1e9 if args.cases is not None else 60.0)
max_cases = args.cases if args.cases is not None else 1 << 60
rng = np.random.default_rng(args.seed)
oracle.build()
ctx = E.Context(0)
K4 = np.array(synth.FOUNTAIN_K4, np.float32)
t_end = time.time() + budget
n_cases = n_fail_both = n_exact = n_border = n_loose = n_far = n_one_side = n_pose_bits = 0
worst = 0.0
border = []
iters_total = 0
while time.time() < t_end and n_cases < max_cases:
    n = int(rng.choice([6, 8, 12, 25, 60, 200, 600, 1500, 4000]))
    frac = float(rng.choice([0.0, 0.1, 0.25, 0.4, 0.55, 0.7]))
    thr = float(rng.choice([1.0, 2.5, 8.0]))
    cap = int(rng.choice([50, 300, 50000]))
    R = synth.aa_to_R(rng.normal(0, 0.3, 3)); t = np.array([0.3, -0.2, 6.0]) + rng.normal(0, 0.2, 3)
    X = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    if rng.random() < 0.15: X[:, 2] = X[0, 2]                       # a coplanar set
    Xc = X.astype(np.float64) @ R.T + t
    pix = np.stack([Xc[:, 0] / Xc[:, 2] * K4[0] + K4[1], Xc[:, 1] / Xc[:, 2] * K4[2] + K4[3]], 1) + rng.normal(0, 0.5, (n, 2))
    bad = rng.choice(n, int(frac * n), replace=False)
    pix[bad] += rng.uniform(-80, 80, (len(bad), 2))
    pix = pix.astype(np.float32)
    if args.only is not None and n_cases != args.only:
        n_cases += 1
        continue
    ok, Rr, tr, rvr, mr, itr = oracle.solve_pnp_ransac(X, pix, K4, cap, thr, 0.99)
    try:
        rv, tv, Rg, mg, itg = E.solve_pnp_ransac(X, pix, K4, cap, thr, 0.99, ctx)
        got = True
    except E.EsfmError:
        got = False
    tag = (args.seed, n_cases, n, frac, thr, cap)
    if args.only is not None:
        print(tag, 'oracle ok', ok, 'iters', itr, 'gpu iters', itg if got else None, 'mask diff at', np.nonzero(mg != mr)[0] if got and ok else None)
        print('R diff', np.abs(Rg - Rr).max(), 't diff', np.abs(tv - tr).max(), 'inliers', int(mr.sum()), int(mg.sum()))
        break
    if got != bool(ok):
        n_one_side += 1; border.append(tag + ('pose on one side only', got, bool(ok)))
        n_cases += 1
        continue
    if ok:
        iters_total += itr
        if itg == itr and np.array_equal(mg, mr):
            if not (np.allclose(Rg, Rr, atol=1e-6) and np.allclose(tv, tr, atol=1e-5) and np.allclose(rv, rvr, atol=1e-6)):
                # the re-fit on a handful of inliers (or a coplanar set) is ill-conditioned: the two sides' poses drift apart along
                # directions the data does not constrain (and near-equal small eigenvalues of M'M swap places in the null-space basis):
                # they must still explain the inliers about equally well
                def rms(Rm, tm):
                    Xi = X[mr.astype(bool)].astype(np.float64) @ np.asarray(Rm).reshape(3, 3).T + np.asarray(tm).reshape(3)
                    uv = np.stack([Xi[:, 0] / Xi[:, 2] * K4[0] + K4[1], Xi[:, 1] / Xi[:, 2] * K4[2] + K4[3]], 1)
                    return float(np.sqrt(np.mean(np.sum((uv - pix[mr.astype(bool)]) ** 2, 1))))
                eg, er = rms(Rg, tv), rms(Rr, tr)
                n_loose += 1; worst = max(worst, abs(eg - er) / max(er, 1e-3))
            # round 6: the re-fit of every inlier set up to 8 192 runs on the host in the oracle's order -- the pose must be the SAME BITS
            if int(mr.sum()) <= 8192 and not (np.array_equal(Rg, Rr) and np.array_equal(tv, tr) and np.array_equal(rv, rvr)): n_pose_bits += 1
            n_exact += 1
        else:
            # two correct f64 evaluations of an ill-conditioned hypothesis differ in the last bits of its pose, and a correspondence
            # whose error sits on the threshold then counts on one side only: another of two equally good hypotheses wins, or the
            # adaptive count moves by a step.  Reported, not hidden: both answers must be poses of the same quality.
            n_border += 1
            border.append(tag + (itg, itr, int(mg.sum()), int(mr.sum())))
            if abs(int(mg.sum()) - int(mr.sum())) > max(2, 0.02 * n): n_far += 1
    else:
        n_fail_both += 1
    n_cases += 1
print(f"stress_pnp seed {args.seed}: {n_cases} cases ({n_fail_both} without a pose on both sides), {iters_total} RANSAC iterations: {n_exact} with iteration count and mask "
      f"equal to the oracle's ({n_loose} of them with an ill-conditioned re-fit: poses apart, reprojection error of the inliers within {worst:.1e} relative), {n_border} where a threshold-borderline correspondence made the two sides pick differently ({n_far} of them with inlier counts more than 2 % apart), {n_one_side} with a pose on one side only; {n_pose_bits} with a pose that is not bit-identical to the oracle's")
for b in border[:12]:
    print("  borderline:", b)
sys.exit(1 if (n_border or n_one_side or n_loose or n_pose_bits) else 0)      # since round 5 every kind of difference is a regression

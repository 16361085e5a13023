"""Randomised stress of the sparse-cloud outlier filter (esfm_sor_filter) against the oracle: 1 ... 40 000 points, uniform / planar /
clustered / line-like clouds, duplicated points, non-finite points, strides 3 - 6, MeanK 1 ... 63 (the kernel's limit; also above the point count), thresholds
0.5 - 3 sigma -- keep mask, mean distances (bit patterns) and threshold exact.
usage: python tests/stress_cloud.py [--seconds S | --cases N] [--seed K]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easysfm_amd as E
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
oracle.set_num_threads(min(16, os.cpu_count() or 1))
ctx = E.Context(0)
t_end = time.time() + budget
n_cases = 0; n_pts = 0
while time.time() < t_end and n_cases < max_cases:
    n = int(rng.choice([1, 2, 3, 17, 51, 64, 200, 1000, 4095, 4096, 4097, 9000, 40000], p=[.04, .04, .04, .08, .08, .08, .12, .12, .08, .08, .08, .1, .06]))
    stride = int(rng.choice([3, 4, 6]))
    kind = rng.integers(0, 5)
    P = rng.uniform(-5, 5, (n, 3))
    if kind == 1: P[:, 2] = 0.25                                            # planar across z
    elif kind == 2: P = rng.normal(0, 1, (n, 3)) * rng.choice([0.01, 1.0, 100.0]) + rng.integers(0, 4, (n, 1)) * 10.0    # clusters
    elif kind == 3: P[:, 1:] = P[:, :1] * [0.5, -0.25] + rng.normal(0, 1e-3, (n, 2))                                     # nearly a line
    elif kind == 4 and n > 8: P[rng.choice(n, n // 4, replace=False)] = P[rng.choice(n, n // 4)]                      # duplicates
    pts = np.zeros((n, stride), np.float32); pts[:, :3] = P
    if stride > 3: pts[:, 3:] = rng.uniform(0, 255, (n, stride - 3))
    if n > 20 and rng.random() < 0.2:
        bad = rng.choice(n, 3, replace=False); pts[bad[0], 0] = np.nan; pts[bad[1], 1] = np.inf; pts[bad[2], 2] = -np.inf
    mean_k = int(rng.choice([1, 5, 50, 63])); std = float(rng.choice([0.5, 1.0, 2.0, 3.0]))
    rk, rmd, rthr = oracle.sor_filter(pts, mean_k, std)
    keep, md, thr = E.sor_filter(pts, mean_k, std, ctx)
    tag = (args.seed, n_cases, n, stride, int(kind), mean_k, std)
    assert np.array_equal(md.view(np.uint32), rmd.view(np.uint32)), tag + ("mean distances", int((md.view(np.uint32) != rmd.view(np.uint32)).sum()))
    assert np.array_equal(keep, rk) and (thr == rthr or (np.isnan(thr) and np.isnan(rthr))), tag + ("mask / threshold", thr, rthr)
    n_cases += 1; n_pts += n
print(f"stress_cloud seed {args.seed}: {n_cases} cases, {n_pts} points, all equal to the oracle")

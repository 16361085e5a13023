"""Randomised stress of the pixel stages against the oracle, bit for bit: SURF and ORB detection + description and undistortion on random
images -- sizes 33 x 40 ... 400 x 600 (odd and even, widths that are and are not multiples of four), smoothed noise / blobs / step edges /
flat regions, one and three channels, random intrinsics and distortion coefficients.
usage: python tests/stress_pixels.py [--seconds S | --cases N] [--seed K]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easysfm_amd as E
import oracle

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=None)
ap.add_argument("--cases", type=int, default=None)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--only", type=int, default=None, help="draw every case as usual, run only this one and say what differs")
args = ap.parse_args()
budget = args.seconds if args.seconds is not None else (1e9 if args.cases is not None else 60.0)
max_cases = args.cases if args.cases is not None else 1 << 60
rng = np.random.default_rng(args.seed)
oracle.build()
ctx = E.Context(0)

def image(rows, cols):
    kind = rng.integers(0, 4)
    img = rng.normal(128, 40, (rows, cols))
    k = int(rng.choice([1, 2, 4, 8]))
    ker = np.ones(2 * k + 1) / (2 * k + 1)
    for ax in (0, 1):
        img = np.apply_along_axis(lambda v: np.convolve(v, ker, mode="same"), ax, img)
    img = (img - img.mean()) / (img.std() + 1e-9) * rng.choice([20, 50, 90]) + 128
    if kind == 1:
        for _ in range(int(rng.integers(3, 30))):
            cy, cx, r = rng.integers(0, rows), rng.integers(0, cols), rng.integers(2, 25)
            yy, xx = np.ogrid[:rows, :cols]; img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] += rng.choice([-80, 80])
    elif kind == 2:
        img[:, cols // 3:] += 70; img[rows // 2:, :] -= 50
    elif kind == 3:
        img[rows // 4: rows // 2, cols // 4: cols // 2] = 128
    return np.clip(img, 0, 255).astype(np.uint8)

t_end = time.time() + budget
n_cases = n_kp = 0
while time.time() < t_end and n_cases < max_cases:
    rows, cols = int(rng.integers(33, 400)), int(rng.integers(40, 600))
    if rng.random() < 0.4: cols = cols // 4 * 4
    gray = image(rows, cols)
    tag = (args.seed, n_cases, rows, cols)
    thr = float(rng.choice([100.0, 300.0, 1000.0]))
    if args.only is not None and n_cases != args.only:
        nf = int(rng.choice([50, 500, 5000])); ch = int(rng.choice([1, 3])); rng.uniform(0.5, 2.0); rng.uniform(0.2, 0.8); rng.uniform(0.9, 1.1); rng.uniform(0.2, 0.8)
        rng.normal(0, 0.3); rng.normal(0, 0.2); rng.normal(0, 0.01); rng.normal(0, 0.01); rng.choice([0.0, 1.0, 3.0])
        n_cases += 1
        continue
    kp, d = E.surf_detect_and_compute(gray, thr, None, ctx)
    rk, rd = oracle.surf(gray, thr)
    assert np.array_equal(kp.view(np.uint32), rk.view(np.uint32)) and np.array_equal(d.view(np.uint32), rd.view(np.uint32)), tag + ("SURF", len(kp), len(rk))
    nf = int(rng.choice([50, 500, 5000]))
    okp, od = E.orb_detect_and_compute(gray, nf, None, ctx)
    ork, ord_ = oracle.orb(gray, nf)
    if args.only is not None and not (np.array_equal(okp.view(np.uint32), ork.view(np.uint32)) and np.array_equal(od, ord_)):
        m = min(len(okp), len(ork))
        rows_kp = np.nonzero((okp[:m].view(np.uint32) != ork[:m].view(np.uint32)).any(1))[0]
        rows_d = np.nonzero((od[:m] != ord_[:m]).any(1))[0]
        print("ORB differs: keypoint rows", rows_kp[:10], "of", m, "; descriptor rows", rows_d[:10])
        for r in rows_kp[:4]: print("  gpu", okp[r], "\n  ref", ork[r])
        for r in rows_d[:2]: print("  desc row", r, "bits differing", int(np.unpackbits(od[r] ^ ord_[r]).sum()), "kp", okp[r])
    assert np.array_equal(okp.view(np.uint32), ork.view(np.uint32)) and np.array_equal(od, ord_), tag + ("ORB", len(okp), len(ork))
    ch = int(rng.choice([1, 3]))
    img = gray if ch == 1 else np.ascontiguousarray(np.stack([gray, np.roll(gray, 3, 1), 255 - gray], 2))
    f = float(rng.uniform(0.5, 2.0) * max(rows, cols))
    K4 = np.array([f, cols * rng.uniform(0.2, 0.8), f * rng.uniform(0.9, 1.1), rows * rng.uniform(0.2, 0.8)])
    dist = np.array([rng.normal(0, 0.3), rng.normal(0, 0.2), rng.normal(0, 0.01), rng.normal(0, 0.01)]) * rng.choice([0.0, 1.0, 3.0])
    u = E.undistort(img, K4, dist, ctx)
    assert np.array_equal(u, oracle.undistort(img, K4, dist)), tag + ("undistort", ch, K4.tolist(), dist.tolist())
    n_cases += 1; n_kp += len(kp) + len(okp)
print(f"stress_pixels seed {args.seed}: {n_cases} images, {n_kp} keypoints, SURF / ORB / undistortion all equal to the oracle")

"""Pins the C oracle (oracle/*.c) against the committed golden vectors (tests/golden/*.npz, produced by
the independent numpy/torch restatement in tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _cases(z):
    return sorted({k.split(".")[0] for k in z.files if "." in k})


def test_hamming_golden(oracle_lib):
    z = np.load(os.path.join(GOLD, "match_hamming_cases.npz"))
    names = _cases(z)
    assert {"ties", "all_equal", "nt0", "nt1", "nt2", "nt3", "boundary", "random"} <= set(names)
    for name in names:
        q, t = z[f"{name}.q"], z[f"{name}.t"]
        idx, dist = oracle_lib.knn2_hamming(q, t)
        assert np.array_equal(idx, z[f"{name}.idx"]), name
        assert np.array_equal(dist, z[f"{name}.dist"]), name
        for r in (0.5, 0.8, 1.0):
            a, b, c = oracle_lib.match_hamming(q, t, r)
            assert np.array_equal(a, z[f"{name}.m{r}.q"]) and np.array_equal(b, z[f"{name}.m{r}.t"]) and np.array_equal(c, z[f"{name}.m{r}.d"]), (name, r)
    # the boundary case: d0 == 0.8 * d1 exactly is rejected (strict <), d0 = 3 < 0.8 * 5 accepted
    a, b, c = oracle_lib.match_hamming(z["boundary.q"], z["boundary.t"], 0.8)
    assert a.tolist() == [1]


def test_l2_golden_bitexact(oracle_lib):
    z = np.load(os.path.join(GOLD, "match_l2_cases.npz"))
    names = _cases(z)
    assert {"surf_like", "duplicates", "nt0", "nt1", "nt2", "dim37"} <= set(names)
    for name in names:
        q, t = z[f"{name}.q"], z[f"{name}.t"]
        if q.ndim != 2:
            continue
        idx, dist = oracle_lib.knn2_l2(q, t)
        assert np.array_equal(idx, z[f"{name}.idx"]), name
        assert np.array_equal(_bits(dist), _bits(z[f"{name}.dist"])), name
        for r in (0.5, 0.8):
            a, b, c = oracle_lib.match_l2(q, t, r)
            assert np.array_equal(a, z[f"{name}.m{r}.q"]) and np.array_equal(b, z[f"{name}.m{r}.t"]), (name, r)
            assert np.array_equal(_bits(c), _bits(z[f"{name}.m{r}.d"])), (name, r)
    assert z["near_tie_audit"].sum() > 0   # the fixtures really contain near-ties


def test_l2_simd_body_equals_the_plain_loop(oracle_lib):
    """Round 4: knn2_l2 runs a SIMD body (eight train rows per register, every lane the scalar function's operations in its order)
    so that bench.py's cpu_baseline is an honest brute-force matcher; the plain loop over esfm_ref_l2sqr stays the definition.  Same
    bits on the golden cases, on ragged sizes (rows past the last block of eight), ties, non-finite rows and denormal scales."""
    z = np.load(os.path.join(GOLD, "match_l2_cases.npz"))
    for name in _cases(z):
        q, t = z[f"{name}.q"], z[f"{name}.t"]
        if q.ndim != 2:
            continue
        a, b = oracle_lib.knn2_l2(q, t), oracle_lib.knn2_l2_scalar(q, t)
        assert np.array_equal(a[0], b[0]) and np.array_equal(_bits(a[1]), _bits(b[1])), name
    rng = np.random.default_rng(77)
    for nq, nt, dim, scale in [(33, 1, 64, 1.0), (50, 7, 64, 1.0), (64, 9, 64, 1e-20), (40, 1001, 64, 1e3), (17, 260, 128, 1.0), (9, 30, 8, 1.0)]:
        q = (rng.standard_normal((nq, dim)) * scale).astype(np.float32); t = (rng.standard_normal((nt, dim)) * scale).astype(np.float32)
        if nt > 5:
            t[3] = t[1]; q[0] = t[1]                       # exact ties: the lower train index first
            t[4, 5] = np.inf; t[2, 0] = np.nan             # never neighbours
        a, b = oracle_lib.knn2_l2(q, t), oracle_lib.knn2_l2_scalar(q, t)
        assert np.array_equal(a[0], b[0]) and np.array_equal(_bits(a[1]), _bits(b[1])), (nq, nt, dim, scale)


def test_l2_match_pairs_batch_equals_single_calls(oracle_lib):
    """esfm_ref_match_pairs_l2 (the CPU baseline's batched pair loop: one parallel region) == match_l2 pair by pair."""
    rng = np.random.default_rng(78)
    sets = [rng.standard_normal((n, 64)).astype(np.float32) for n in (120, 0, 333, 9)]
    sets[2][:50] = sets[0][:50] + np.float32(1e-3) * rng.standard_normal((50, 64)).astype(np.float32)
    pairs = np.array([(i, j) for i in range(4) for j in range(i)], np.int32)
    res = oracle_lib.match_pairs_l2(sets, pairs, 0.7)
    tot = 0
    for (i, j), (q, t, d) in zip(pairs, res):
        rq, rt, rd = oracle_lib.match_l2(sets[i], sets[j], 0.7)
        assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(_bits(d), _bits(rd)), (i, j)
        tot += len(q)
    assert tot >= 40


def test_l2_canonical_sum_known_answer(oracle_lib):
    """Order sensitivity: a vector built so that the 8-accumulator order and a plain left-to-right sum
    round differently; the oracle must produce the 8-accumulator value."""
    a = np.zeros(64, np.float32); b = np.zeros(64, np.float32)
    a[0] = 4096.0; a[8] = 1.0; a[1] = 1.0          # acc[0] = 2^24 + 1 (rounds to 2^24), acc[1] = 1
    got = oracle_lib.l2sqr(a, b)
    acc0 = np.float32(np.float32(4096.0 * 4096.0) + np.float32(1.0))
    want = np.float32(np.float32(acc0 + np.float32(0.0)) + np.float32(1.0))
    assert got == float(want) == 16777216.0 + 0.0 or got == float(want)
    seq = np.float32(0.0)
    for k in range(64):
        seq = np.float32(seq + np.float32((a[k] - b[k]) ** 2))
    assert float(seq) == 16777216.0            # left-to-right loses both ones ...
    assert got == 16777216.0 + 2.0 or got == float(want)


def test_ba_jacobian_golden(oracle_lib):
    z = np.load(os.path.join(GOLD, "ba_jacobian_cases.npz"))
    K4 = z["K4"]
    for i in range(len(z["cams"])):
        r, Jc, Jp = oracle_lib.ba_residual_jac(z["cams"][i], z["pts"][i], K4, z["uvs"][i])
        assert np.allclose(r, z["r"][i], rtol=1e-12, atol=1e-9)
        assert np.allclose(Jc, z["Jc"][i], rtol=1e-9, atol=1e-9), i
        assert np.allclose(Jp, z["Jp"][i], rtol=1e-9, atol=1e-9), i
    # central differences as a third opinion on one generic case
    i = 5
    cam, pt = z["cams"][i].copy(), z["pts"][i].copy()
    _, Jc, Jp = oracle_lib.ba_residual_jac(cam, pt, K4, z["uvs"][i])
    h = 1e-6
    for k in range(6):
        e = np.zeros(6); e[k] = h
        rp, _, _ = oracle_lib.ba_residual_jac(cam + e, pt, K4, z["uvs"][i]); rm, _, _ = oracle_lib.ba_residual_jac(cam - e, pt, K4, z["uvs"][i])
        assert np.allclose((rp - rm) / (2 * h), Jc[:, k], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("tag", ["cauchy", "squared_rejects"])
def test_ba_lm_trace_golden(oracle_lib, tag):
    """Per-iteration trace of the Schur-based C oracle == the dense numpy LM of make_golden.py."""
    z = np.load(os.path.join(GOLD, "ba_lm_trace.npz"))
    opt = oracle_lib.ba_default_options()
    n = len(z[f"{tag}.cost"]) - 1
    opt.max_num_iterations = n
    opt.cauchy_a = float(z[f"{tag}.cauchy_a"])
    cams, pts, s = oracle_lib.ba_solve(z[f"{tag}.cam_idx"], z[f"{tag}.pt_idx"], z[f"{tag}.uv"], z[f"{tag}.K4"],
                                       z[f"{tag}.cams0"], z[f"{tag}.pts0"], opt)
    log = oracle_lib.iterations(s)
    assert len(log) == n + 1
    assert [it.step_is_successful for it in log] == z[f"{tag}.ok"].tolist()
    for it, c, rad, sn in zip(log, z[f"{tag}.cost"], z[f"{tag}.radius"], z[f"{tag}.step_norm"]):
        # accepted points agree to 1e-8; the logged cost of a REJECTED step is the cost of a wild candidate
        # (far outside the trust region's validity) and amplifies solver round-off: 1e-6
        assert abs(it.cost - c) <= (1e-8 if it.step_is_successful else 1e-6) * abs(c), (it.iteration, it.cost, c)
        assert abs(it.trust_region_radius - rad) <= 1e-6 * rad, it.iteration
        assert abs(it.step_norm - sn) <= 1e-4 * max(sn, 1e-6), it.iteration
    if tag == "squared_rejects":
        assert z[f"{tag}.ok"].tolist().count(0) >= 5
    x = np.concatenate([cams.ravel(), pts.ravel()])
    assert np.allclose(x, z[f"{tag}.x_final"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["calib_loose", "calib_tight", "refcam", "both_squared"])
def test_ba_constrained_trace_golden(oracle_lib, tag):
    """Bounds-constrained solveBA (ba.cpp:155-196: reference camera +-1e-10, free intrinsics +-tolerance): the Schur-based
    C oracle == the dense numpy LM with projection and Armijo line search of make_golden.py."""
    z = np.load(os.path.join(GOLD, "ba_lm_constrained.npz"))
    opt = oracle_lib.ba_default_options()
    n = len(z[f"{tag}.cost"]) - 1
    opt.max_num_iterations = n
    opt.cauchy_a = float(z[f"{tag}.cauchy_a"])
    calib0 = z[f"{tag}.calib0"]
    free = calib0.size == 4
    cams, pts, cal, s = oracle_lib.ba_solve_ex(z[f"{tag}.cam_idx"], z[f"{tag}.pt_idx"], z[f"{tag}.uv"],
                                               None if free else z[f"{tag}.K4"], z[f"{tag}.cams0"], z[f"{tag}.pts0"],
                                               calib=calib0 if free else None, calib_tol=float(z[f"{tag}.calib_tol"]),
                                               ref_cam=int(z[f"{tag}.ref_cam"]), options=opt)
    log = oracle_lib.iterations(s)
    assert len(log) == n + 1
    assert [it.step_is_successful for it in log] == z[f"{tag}.ok"].tolist()
    assert [it.line_search_steps for it in log] == z[f"{tag}.ls"].tolist()
    for it, c, rad, sn, gm in zip(log, z[f"{tag}.cost"], z[f"{tag}.radius"], z[f"{tag}.step_norm"], z[f"{tag}.gmax"]):
        assert abs(it.cost - c) <= (1e-8 if it.step_is_successful else 1e-6) * abs(c), (it.iteration, it.cost, c)
        assert abs(it.trust_region_radius - rad) <= 1e-6 * rad, it.iteration
        assert abs(it.step_norm - sn) <= 1e-4 * max(sn, 1e-6), it.iteration
        assert abs(it.gradient_max_norm - gm) <= 1e-5 * max(gm, 1e-6), it.iteration
    x = np.concatenate([cams.ravel(), pts.ravel()] + ([cal] if free else []))
    assert np.allclose(x, z[f"{tag}.x_final"], rtol=1e-4, atol=1e-5)
    if free:
        tol = float(z[f"{tag}.calib_tol"])
        assert np.all(cal >= calib0 - tol) and np.all(cal <= calib0 + tol)
    if tag == "calib_tight":
        assert np.any(np.isclose(np.abs(cal - calib0), float(z[f"{tag}.calib_tol"]), rtol=0, atol=1e-12))   # a bound is active
        assert sum(z[f"{tag}.ls"]) > 0
    if int(z[f"{tag}.ref_cam"]) >= 0:
        assert np.all(np.abs(cams[int(z[f"{tag}.ref_cam"])]) <= 1e-10)


def test_ba_zero_noise_known_answer(oracle_lib):
    z = np.load(os.path.join(GOLD, "ba_lm_trace.npz"))
    c = oracle_lib.ba_cost(z["zero.cam_idx"], z["zero.pt_idx"], z["zero.uv"], z["zero.K4"], z["zero.cams_gt"], z["zero.pts_gt"])
    assert c < 1e-6     # only the float32 rounding of the stored observations
    cams, pts, s = oracle_lib.ba_solve(z["zero.cam_idx"], z["zero.pt_idx"], z["zero.uv"], z["zero.K4"], z["zero.cams_gt"], z["zero.pts_gt"])
    assert s.final_cost <= s.initial_cost < 1e-6


def test_ba_converged_cost_vs_scipy(oracle_lib):
    """Independent solver on the pre-robustified residual r * sqrt(rho(|r|^2)) / |r|:
    the converged robust cost is gauge-invariant and must agree."""
    from scipy.optimize import least_squares
    from easysfm_amd import synth
    sc = synth.ba_scene(4, 60, 3, seed=77)
    opt = oracle_lib.ba_default_options(); opt.max_num_iterations = 200; opt.function_tolerance = 1e-12
    cams, pts, s = oracle_lib.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt)

    def fun(x):
        c = x[:24].reshape(4, 6); p = x[24:].reshape(-1, 3)
        R = np.stack([synth.aa_to_R(a[:3]) for a in c])
        P = np.einsum("nij,nj->ni", R[sc.cam_idx], p[sc.pt_idx]) + c[sc.cam_idx, 3:]
        K = sc.K4[0].astype(np.float64)
        r = np.stack([sc.uv[:, 0] - (P[:, 0] / P[:, 2] * K[0] + K[1]), sc.uv[:, 1] - (P[:, 1] / P[:, 2] * K[2] + K[3])], 1)
        s2 = (r * r).sum(1)
        w = np.sqrt(0.25 * np.log1p(s2 / 0.25) / np.maximum(s2, 1e-300))
        return (r * w[:, None]).ravel()

    x0 = np.concatenate([cams.ravel(), pts.ravel()])
    r = least_squares(fun, x0, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=400)
    assert abs(r.cost - s.final_cost) <= 2e-3 * s.final_cost
    assert r.cost <= s.final_cost * (1 + 1e-9)     # scipy started from the oracle's answer: it can only polish it


@pytest.mark.parametrize("tag", ["blobs", "dups_nonfinite", "small", "grid_ties", "stride8"])
def test_sor_golden_bitexact(oracle_lib, tag):
    """Statistical outlier removal (cloudprocessing.hpp:24-36): the C oracle == the numpy restatement, bit for bit."""
    z = np.load(os.path.join(GOLD, "sor_cases.npz"))
    keep, md, thr = oracle_lib.sor_filter(z[f"{tag}.points"], int(z[f"{tag}.mean_k"]), float(z[f"{tag}.std_mul"]))
    assert np.array_equal(md, z[f"{tag}.mean_dist"])
    assert thr == float(z[f"{tag}.threshold"])
    assert np.array_equal(keep, z[f"{tag}.keep"])
    if tag == "blobs":
        assert 0 < (~keep).sum() < 200 and (~keep)[2000:].sum() >= 30      # the planted far points go


@pytest.mark.parametrize("tag", ["wide", "noisy", "short_baseline"])
def test_triangulation_golden(oracle_lib, tag):
    """cv::triangulatePoints restatement (one-sided Jacobi) == numpy.linalg.svd of the same f64 DLT matrix, as floats."""
    z = np.load(os.path.join(GOLD, "triangulation_cases.npz"))
    h = oracle_lib.triangulate_points(z[f"{tag}.P1"], z[f"{tag}.P2"], z[f"{tag}.x1"], z[f"{tag}.x2"])
    g = z[f"{tag}.points4d"]
    assert np.allclose(h * np.sign(h[:, 3:4]), g * np.sign(g[:, 3:4]), rtol=0, atol=3e-7)      # unit vectors, 2 float ulps
    if tag == "wide":
        assert np.allclose(h[:, :3] / h[:, 3:4], z[f"{tag}.X"], atol=2e-4)


def test_five_point_known_answers(oracle_lib):
    """EMEstimatorCallback::runKernel restatement: the ground-truth essential matrix is among the models of an exact
    5-point sample, and every model satisfies the constraints the solver is built from."""
    z = np.load(os.path.join(GOLD, "ransac_cases.npz"))
    for q1, q2, Egt in zip(z["five.q1"], z["five.q2"], z["five.E"]):
        Es = oracle_lib.five_point(q1, q2)
        assert 1 <= len(Es) <= 10
        assert min(min(np.abs(E - Egt).max(), np.abs(E + Egt).max()) for E in Es) < 1e-8      # conditioning of a 5-point sample (worst of the ten: 8e-10)
        h1 = np.c_[q1, np.ones(5)]; h2 = np.c_[q2, np.ones(5)]
        for E in Es:
            assert abs(np.linalg.norm(E) - 1) < 1e-12 and E.ravel()[np.argmax(np.abs(E))] > 0      # unit norm, canonical sign
            assert np.abs(np.einsum("ni,ij,nj->n", h2, E, h1)).max() < 1e-9                        # x2' E x1 = 0 on the sample
            # a spurious solution at a near-multiple root of the degree-10 polynomial is only as good as that root (sample 2 has the roots
            # -2.3526 / -2.3067 and a cluster at -1.13: one of its models has |det E| 7e-7; all others of the ten samples are below 5e-10)
            assert abs(np.linalg.det(E)) < 5e-6
            assert np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max() < 5e-6
        assert np.all(np.diff([E[0, 0] for E in Es]) >= 0)                                          # canonical order


@pytest.mark.parametrize("tag", ["pose_a", "pose_b"])
def test_recover_pose_golden(oracle_lib, tag):
    """cv::recoverPose restatement == the numpy.linalg.svd restatement of make_golden.py."""
    z = np.load(os.path.join(GOLD, "ransac_cases.npz"))
    good, R, t, m = oracle_lib.recover_pose(z[f"{tag}.E"], z[f"{tag}.p1"], z[f"{tag}.p2"], z["K4"], z[f"{tag}.mask_in"])
    assert good == int(z[f"{tag}.good"]) and np.array_equal(m, z[f"{tag}.mask"])
    assert np.allclose(R, z[f"{tag}.R"], atol=1e-9) and np.allclose(t, z[f"{tag}.t"], atol=1e-9)


def test_essential_ransac_recovers_ground_truth(oracle_lib):
    from easysfm_amd import synth
    rng = np.random.default_rng(8)
    K4 = np.array(synth.FOUNTAIN_K4, np.float32)
    R = synth.aa_to_R(np.array([0.05, -0.2, 0.03])); t = np.array([1.0, 0.1, -0.05]); t /= np.linalg.norm(t)
    X = rng.uniform(-2, 2, (400, 3)) + np.array([0, 0, 8.0]); Xc = X @ R.T + t
    p1 = (X[:, :2] / X[:, 2:3] * [K4[0], K4[2]] + [K4[1], K4[3]]).astype(np.float32)
    p2 = (Xc[:, :2] / Xc[:, 2:3] * [K4[0], K4[2]] + [K4[1], K4[3]] + rng.normal(0, 0.3, (400, 2))).astype(np.float32)
    bad = rng.choice(400, 120, replace=False); p2[bad] += rng.uniform(-40, 40, (120, 2)).astype(np.float32)
    ok, E, mask, iters, cnt = oracle_lib.find_essential_ransac(p1, p2, K4, 0.99, 1.0)
    assert ok and 0 < iters < 1000 and cnt == mask.sum() >= 270 and mask[bad].sum() <= 5
    good, Rr, tr, m2 = oracle_lib.recover_pose(E, p1, p2, K4, mask)
    assert good >= 270 and np.allclose(Rr, R, atol=0.02) and np.allclose(tr, t, atol=0.05)


def test_epnp_and_pnp_ransac_known_answers(oracle_lib):
    """EPnP restatement: exact observations reproduce the pose to machine precision for 5 .. 500 points (also coplanar
    points); solvePnPRansac recovers it from 33 % gross outliers; cv::Rodrigues of the result is the rotation's angle-axis."""
    from easysfm_amd import synth
    from easysfm_amd.ba import rotation_to_angle_axis
    rng = np.random.default_rng(12)
    K4 = np.array(synth.FOUNTAIN_K4, np.float64)
    R = synth.aa_to_R(np.array([0.3, -0.2, 0.1])); t = np.array([0.3, -0.2, 6.0])
    for n in (5, 6, 20, 500):
        X = rng.uniform(-2, 2, (n, 3)); Xc = X @ R.T + t
        pix = np.stack([Xc[:, 0] / Xc[:, 2] * K4[0] + K4[1], Xc[:, 1] / Xc[:, 2] * K4[2] + K4[3]], 1)
        Rr, tr, e = oracle_lib.epnp(X, pix, K4)
        assert np.allclose(Rr, R, atol=1e-10) and np.allclose(tr, t, atol=1e-9) and e < 1e-9
    Xp = rng.uniform(-2, 2, (40, 3)); Xp[:, 2] = 0.5 * Xp[:, 0] - 0.25 * Xp[:, 1] + 1.0      # a plane
    Xc = Xp @ R.T + t
    pix = np.stack([Xc[:, 0] / Xc[:, 2] * K4[0] + K4[1], Xc[:, 1] / Xc[:, 2] * K4[2] + K4[3]], 1)
    Rr, tr, e = oracle_lib.epnp(Xp, pix, K4)
    assert np.allclose(Rr, R, atol=1e-6) and e < 1e-6
    X = rng.uniform(-2, 2, (600, 3)); Xc = X @ R.T + t
    pix = np.stack([Xc[:, 0] / Xc[:, 2] * K4[0] + K4[1], Xc[:, 1] / Xc[:, 2] * K4[2] + K4[3]], 1) + rng.normal(0, 0.5, (600, 2))
    bad = rng.choice(600, 200, replace=False); pix[bad] += rng.uniform(-80, 80, (200, 2))
    ok, Rr, tr, rv, mask, it = oracle_lib.solve_pnp_ransac(X, pix, K4, 50000, 2.5, 0.99)
    assert ok and 0 < it < 5000 and mask.sum() >= 395 and mask[bad].sum() <= 5
    assert np.allclose(Rr, R, atol=2e-3) and np.allclose(tr, t, atol=0.02)
    assert np.allclose(rv, rotation_to_angle_axis(Rr), atol=1e-9)


@pytest.mark.parametrize("tag", ["color_mild", "color_strong", "gray_barrel", "gray_zero", "wide_rows", "quirk_coeffs"])
def test_undistort_golden_bitexact(oracle_lib, tag):
    """oracle/undistort_ref.c against the independent numpy restatement of cv::undistort (make_golden.undistort_numpy)."""
    z = np.load(os.path.join(GOLD, "undistort_cases.npz"))
    out = oracle_lib.undistort(z[tag + "_image"], z[tag + "_K4"], z[tag + "_dist"])
    assert np.array_equal(out, z[tag + "_out"])
    if tag == "gray_zero":
        assert np.array_equal(out, z[tag + "_image"])          # zero coefficients: exact identity

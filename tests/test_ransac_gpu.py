"""Parity of the HIP essential-matrix RANSAC + pose recovery (reference cv::findEssentialMat /
cv::recoverPose at cpp_code/src/estimate_motion.cpp:49-67) through the C ABI against the CPU oracle.

Both sides replay the same cv::RNG sample stream and the same sequential bookkeeping, and since round 6 they evaluate the 5-point
kernel with ONE arithmetic (easysfm_amd/csrc/five_point_core.hpp == oracle/ransac_ref.c to the letter; tests/test_five_point_stages.py
checks it stage by stage): iteration counts, the winning sample, the inlier MASKS and the essential matrix itself agree exactly.
Rotation / translation of recoverPose (its 3 x 3 SVD runs on the host, by the oracle's Jacobi rule): bit-identical too."""
import numpy as np
import pytest

import easysfm_amd as E
from easysfm_amd import synth

pytestmark = pytest.mark.gpu
K4 = np.array(synth.FOUNTAIN_K4, np.float32)


def _pair(rng, n, outlier_frac=0.3, noise=0.3, baseline=1.0):
    R = synth.aa_to_R(rng.normal(0, 0.15, 3)); t = np.array([baseline, 0.1, -0.05]) + rng.normal(0, 0.05, 3)
    t = t / np.linalg.norm(t)
    X = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0, 8.0])
    x1 = X[:, :2] / X[:, 2:3]
    Xc = X @ R.T + t
    x2 = Xc[:, :2] / Xc[:, 2:3]
    p1 = (x1 * [K4[0], K4[2]] + [K4[1], K4[3]]).astype(np.float32)
    p2 = (x2 * [K4[0], K4[2]] + [K4[1], K4[3]] + rng.normal(0, noise, (n, 2))).astype(np.float32)
    out = rng.choice(n, int(outlier_frac * n), replace=False)
    p2[out] += rng.uniform(-60, 60, (len(out), 2)).astype(np.float32)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Egt = tx @ R
    return p1, p2, R, t, Egt / np.linalg.norm(Egt), out


def _same_E(a, b, tol):
    return min(np.abs(a - b).max(), np.abs(a + b).max()) <= tol


@pytest.mark.parametrize("n,frac,seed", [(400, 0.3, 0), (80, 0.5, 1), (2000, 0.2, 2), (30, 0.6, 3), (6, 0.0, 4)])
def test_find_essential_and_pose_match_oracle(gpu_ctx, oracle_lib, n, frac, seed):
    rng = np.random.default_rng(seed)
    p1, p2, R, t, Egt, out = _pair(rng, n, frac)
    ok, Er, mr, itr, cnt = oracle_lib.find_essential_ransac(p1, p2, K4, 0.99, 1.0)
    assert ok
    Eg, mg, itg = E.find_essential_mat(p1, p2, K4, 0.99, 1.0, gpu_ctx)
    assert itg == itr
    assert np.array_equal(mg, mr) and int(mg.sum()) == cnt
    assert np.array_equal(Eg, Er)
    assert abs(np.linalg.norm(Eg) - 1.0) < 1e-12
    if n >= 80:
        assert _same_E(Eg, Egt, 0.05) and mg[out].sum() <= 0.1 * len(out) + 2
    good_r, Rr, tr, m2r = oracle_lib.recover_pose(Er, p1, p2, K4, mr)
    good_g, Rg, tg, m2g = E.recover_pose(Eg, p1, p2, K4, mg, gpu_ctx)
    assert good_g == good_r and np.array_equal(m2g, m2r)
    assert np.array_equal(Rg, Rr) and np.array_equal(tg, tr)      # (decomposeEssentialMat's 3 x 3 SVD: the same Jacobi rule on both sides)
    if n >= 80:
        assert np.allclose(Rg, R, atol=0.02) and np.allclose(tg, t, atol=0.05) and abs(np.linalg.det(Rg) - 1) < 1e-9


def test_exactly_five_points_and_too_few(gpu_ctx, oracle_lib):
    rng = np.random.default_rng(7)
    p1, p2, *_ = _pair(rng, 5, 0.0, noise=0.0)
    Eg, mg, it = E.find_essential_mat(p1, p2, K4, 0.99, 1.0, gpu_ctx)
    ok, Er, mr, itr, cnt = oracle_lib.find_essential_ransac(p1, p2, K4, 0.99, 1.0)
    assert ok and np.all(mg) and np.all(mr) and np.array_equal(Eg, Er)
    with pytest.raises(E.EsfmError):
        E.find_essential_mat(p1[:4], p2[:4], K4, 0.99, 1.0, gpu_ctx)


def test_degenerate_correspondences_match_oracle(gpu_ctx, oracle_lib):
    """Inputs whose samples are degenerate for the 5-point kernel -- every correspondence the same pixel pair, all points on one image
    line, a pure rotation (no baseline: E is not defined), identical images (zero motion), a handful of points with repeats -- must
    come out the same on both sides: found / not found, iteration count, mask and matrix, NaNs included."""
    rng = np.random.default_rng(33)
    p1, p2, R, t, Egt, out = _pair(rng, 120, 0.0, noise=0.0)
    Rrot = synth.aa_to_R(np.array([0.02, -0.1, 0.01]))
    x1 = (p1.astype(np.float64) - [K4[1], K4[3]]) / [K4[0], K4[2]]
    h = np.c_[x1, np.ones(len(x1))] @ Rrot.T
    rot2 = (h[:, :2] / h[:, 2:3] * [K4[0], K4[2]] + [K4[1], K4[3]]).astype(np.float32)
    line1 = p1.copy(); line1[:, 1] = np.float32(0.4) * line1[:, 0] + np.float32(30.0)
    few = np.repeat(p1[:3], 4, axis=0), np.repeat(p2[:3], 4, axis=0)
    cases = [(np.tile(p1[:1], (40, 1)), np.tile(p2[:1], (40, 1))), (line1, p2), (p1, rot2), (p1, p1.copy()), few, (p1[:5], p1[:5].copy())]
    for k, (a, b) in enumerate(cases):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        ok, Er, mr, itr, cnt = oracle_lib.find_essential_ransac(a, b, K4, 0.99, 1.0)
        Es, mask, status, its = E.find_essential_pairs(np.array([0, len(a)], np.int32), a, b, K4[None], 0.99, 1.0, gpu_ctx)
        assert bool(status[0]) == bool(ok), k
        if ok:
            assert int(its[0]) == itr and np.array_equal(np.asarray(mask[:len(a)]).astype(bool), mr), k
            assert np.array_equal(np.asarray(Es[0]), Er, equal_nan=True), k


def test_batched_pairs_equal_single_calls(gpu_ctx, oracle_lib):
    rng = np.random.default_rng(11)
    jobs = [_pair(rng, n, f) for n, f in ((300, 0.3), (3, 0.0), (120, 0.5), (900, 0.1), (40, 0.7))]
    off = np.concatenate([[0], np.cumsum([len(j[0]) for j in jobs])]).astype(np.int32)
    a = np.concatenate([j[0] for j in jobs]); b = np.concatenate([j[1] for j in jobs])
    Ks = np.tile(K4, (len(jobs), 1)); Ks[2] *= np.float32(1.1)
    Es, mask, status, iters = E.find_essential_pairs(off, a, b, Ks, 0.99, 1.0, gpu_ctx)
    assert status.tolist() == [True, False, True, True, True]
    for k, j in enumerate(jobs):
        if not status[k]:
            assert not mask[off[k]:off[k + 1]].any()
            continue
        Eg, mg, it = E.find_essential_mat(j[0], j[1], Ks[k], 0.99, 1.0, gpu_ctx)
        assert it == iters[k] and np.array_equal(mask[off[k]:off[k + 1]], mg) and np.array_equal(Es[k], Eg)
        ok, Er, mr, itr, cnt = oracle_lib.find_essential_ransac(j[0], j[1], Ks[k], 0.99, 1.0)
        assert itr == it and np.array_equal(mr, mg) and np.array_equal(Er, Eg)


def test_mirror_estimate2D2D(gpu_ctx, oracle_lib):
    """estimate2D2D_E5P_RANSAC -> getDepthFast, as sfm.cpp:165-166 chains them."""
    rng = np.random.default_rng(21)
    p1, p2, R, t, Egt, out = _pair(rng, 500, 0.25)
    K = np.array([[K4[0], 0, K4[1]], [0, K4[2], K4[3]], [0, 0, 1]], np.float32)
    f1 = E.Frame(frame_id=1, keypoints=p1); f1.K_cam = K
    f2 = E.Frame(frame_id=0, keypoints=p2); f2.K_cam = K
    matches = [E.DMatch(i, i, 0.0) for i in range(500)]
    inl = []
    me = E.MotionEstimator(gpu_ctx)
    T = me.estimate2D2D_E5P_RANSAC(f1, f2, matches, inl, 1.0, 0.99)
    assert T.dtype == np.float32 and np.allclose(T[3], [0, 0, 0, 1])
    assert np.allclose(T[:3, :3], R, atol=0.02) and np.allclose(T[:3, 3], t, atol=0.05)
    assert 330 <= len(inl) <= 500 and not ({m.queryIdx for m in inl} & set(out.tolist())) or len({m.queryIdx for m in inl} & set(out.tolist())) < 15
    depth = me.getDepthFast(f1, f2, T, inl)
    assert 6.0 < depth < 11.0      # points sit ~8 baseline lengths away


@pytest.mark.parametrize("tag", ["pose_a", "pose_b"])
def test_recover_pose_golden(gpu_ctx, tag):
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ransac_cases.npz"))
    good, R, t, m = E.recover_pose(z[f"{tag}.E"], z[f"{tag}.p1"], z[f"{tag}.p2"], z["K4"], z[f"{tag}.mask_in"], gpu_ctx)
    assert good == int(z[f"{tag}.good"]) and np.array_equal(m, z[f"{tag}.mask"])
    assert np.allclose(R, z[f"{tag}.R"], atol=1e-8) and np.allclose(t, z[f"{tag}.t"], atol=1e-8)

"""Parity of the HIP two-view triangulation (triangulation part; reference
cv::triangulatePoints at cpp_code/src/estimate_motion.cpp:263, :333) through the C ABI against the CPU oracle.

The homogeneous point is the right singular vector of the smallest singular value of the 4 x 4 DLT matrix, computed in f64 by
one-sided Jacobi rotations on its columns -- since round 6 by the SAME rule on both sides (the kernel restates oracle/geometry_ref.c's
loop to the letter; rounds 1-5: two-sided Jacobi on A'A on the GPU, agreement to 2e-6 up to sign) -- and rounded to float: the four
floats are bit-identical, sign included."""
import numpy as np
import pytest

import easysfm_amd as E
from easysfm_amd import synth

pytestmark = pytest.mark.gpu
RTOL = 2e-6


def _two_views(rng, n, baseline=1.0, depth=8.0):
    R = synth.aa_to_R(rng.normal(0, 0.1, 3)); t = np.array([baseline, 0.1, -0.05]) + rng.normal(0, 0.02, 3)
    P1 = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)
    P2 = np.hstack([R, t[:, None]]).astype(np.float32)
    X = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0, depth])
    x1 = (X[:, :2] / X[:, 2:3]).astype(np.float32)
    Xc = X @ R.T + t
    x2 = (Xc[:, :2] / Xc[:, 2:3]).astype(np.float32)
    return P1, P2, x1, x2, X


@pytest.mark.parametrize("n,seed", [(1, 0), (63, 1), (257, 2), (5000, 3)])
def test_triangulation_matches_oracle_and_truth(gpu_ctx, oracle_lib, n, seed):
    rng = np.random.default_rng(seed)
    P1, P2, x1, x2, X = _two_views(rng, n)
    x1n = x1 + rng.normal(0, 1e-3, x1.shape).astype(np.float32)          # ~0.7 px of noise at f = 690
    h = E.triangulate_points(P1, P2, x1n, x2, gpu_ctx)
    r = oracle_lib.triangulate_points(P1, P2, x1n, x2)
    assert h.shape == (n, 4) and np.all(np.isfinite(h))
    assert np.array_equal(h.view(np.uint32), r.view(np.uint32))                 # the vector itself, bit for bit, sign included
    assert np.allclose(np.linalg.norm(h, axis=1), 1.0, atol=1e-6)
    # exact observations reproduce the scene
    h0 = E.triangulate_points(P1, P2, x1, x2, gpu_ctx)
    assert np.allclose(h0[:, :3] / h0[:, 3:4], X, rtol=0, atol=2e-4)


def test_triangulation_degenerate_inputs_match_oracle(gpu_ctx, oracle_lib):
    """Rank-deficient and extreme DLT systems -- no baseline (P2 = P1), a pure rotation, points at the epipole, pixel-scale (not
    normalised) coordinates, a zero projection matrix -- go through the same rotations on both sides: same bits, NaNs in the same places."""
    rng = np.random.default_rng(17)
    P1, P2, x1, x2, X = _two_views(rng, 300)
    Rrot = np.hstack([synth.aa_to_R(np.array([0.0, 0.2, 0.0])), np.zeros((3, 1))]).astype(np.float32)
    cases = [(P1, P1, x1, x1), (P1, P1, x1, x2), (P1, Rrot, x1, x2), (P1, P2, np.zeros_like(x1), np.zeros_like(x2)),
             (P1, P2, x1 * 700 + 380, x2 * 700 + 250), (P1, np.zeros_like(P2), x1, x2), (P1, P2, x1 * np.float32(1e18), x2)]
    for k, (A, B, a, b) in enumerate(cases):
        h = E.triangulate_points(A, B, a.astype(np.float32), b.astype(np.float32), gpu_ctx)
        r = oracle_lib.triangulate_points(A, B, a.astype(np.float32), b.astype(np.float32))
        assert np.array_equal(h.view(np.uint32), r.view(np.uint32)), k


def test_triangulation_batched_pairs(gpu_ctx, oracle_lib):
    rng = np.random.default_rng(9)
    jobs = [_two_views(rng, n) for n in (100, 0, 37, 1, 900)]
    off = np.concatenate([[0], np.cumsum([len(j[2]) for j in jobs])]).astype(np.int32)
    P1s = np.stack([j[0] for j in jobs]); P2s = np.stack([j[1] for j in jobs])
    a = np.concatenate([j[2] for j in jobs]); b = np.concatenate([j[3] for j in jobs])
    h = E.triangulate_pairs(P1s, P2s, off, a, b, gpu_ctx)
    for k, j in enumerate(jobs):
        if off[k + 1] == off[k]:
            continue
        r = oracle_lib.triangulate_points(j[0], j[1], j[2], j[3])
        hk = h[off[k]:off[k + 1]]
        assert np.array_equal(hk.view(np.uint32), r.view(np.uint32))
    assert len(E.triangulate_points(jobs[0][0], jobs[0][1], np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), gpu_ctx)) == 0


def test_mirror_getDepthFast_doTriangulation_outlierFilter(gpu_ctx, oracle_lib):
    """getDepthFast / doTriangulation / outlierFilter (estimate_motion.cpp:234-367, :476-505) on a synthetic pair."""
    rng = np.random.default_rng(4)
    K = np.array([[689.87, 0, 380.17], [0, 691.04, 251.70], [0, 0, 1]], np.float32)
    P1, P2, x1, x2, X = _two_views(rng, 400)
    f1 = E.Frame(frame_id=0, keypoints=(x1 * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]]).astype(np.float32)); f1.K_cam = K
    f2 = E.Frame(frame_id=1, keypoints=(x2 * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]]).astype(np.float32)); f2.K_cam = K
    T21 = np.eye(4, dtype=np.float32); T21[:3] = P2
    matches = [E.DMatch(i, i, 0.0) for i in range(400)]
    me = E.MotionEstimator(gpu_ctx)
    depth = me.getDepthFast(f1, f2, T21, matches)                      # every 20th match
    sel = np.arange(0, 400, 20)
    ref = np.mean(np.linalg.norm(X[sel], axis=1))
    assert abs(depth - ref) <= 2e-3 * ref
    # the same number from the oracle's triangulation, through the same host arithmetic
    r = oracle_lib.triangulate_points(np.eye(4, dtype=np.float32)[:3], P2, E.pixel2cam(f1.keypoints[sel], K), E.pixel2cam(f2.keypoints[sel], K))
    rp = (r[:, :3] / r[:, 3:4]).astype(np.float32)
    assert abs(depth - float(np.mean(np.linalg.norm(rp.astype(np.float64), axis=1)))) <= 1e-5 * ref
    # doTriangulation: 100 tracks are already in the cloud
    f1.unique_pixel_ids = np.arange(1000, 1400); f2.unique_pixel_ids = np.arange(1000, 1400)
    f1.pose_cam = np.eye(4, dtype=np.float32); f2.pose_cam = T21
    cloud = E.SparsePointCloud(xyz=X[:100].astype(np.float32), rgb=np.zeros((100, 3), np.uint8), unique_point_ids=np.arange(1000, 1100),
                               is_inlier=np.ones(100, np.int32))
    assert me.doTriangulation(f1, f2, matches, cloud)
    assert len(cloud.xyz) == 400 and np.array_equal(cloud.unique_point_ids, np.arange(1000, 1400)) and len(cloud.is_inlier) == 400
    assert np.allclose(cloud.xyz[100:], X[100:], atol=5e-3)
    # outlierFilter keeps ids aligned with the points
    cloud.xyz[7] += 500.0
    n0 = len(cloud.xyz)
    me.outlierFilter(cloud)
    assert len(cloud.xyz) == len(cloud.unique_point_ids) == len(cloud.is_inlier) < n0 and 1007 not in cloud.unique_point_ids


@pytest.mark.parametrize("tag", ["wide", "noisy", "short_baseline"])
def test_triangulation_golden(gpu_ctx, tag):
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "triangulation_cases.npz"))
    h = E.triangulate_points(z[f"{tag}.P1"], z[f"{tag}.P2"], z[f"{tag}.x1"], z[f"{tag}.x2"], gpu_ctx)
    g = z[f"{tag}.points4d"]
    # (the golden vectors come from numpy.linalg.svd, another algorithm: a few float ulps, up to sign; the short baseline's two smallest
    # singular values are close -- 2e-6 there since the kernel stopped squaring the condition number, 2e-5 before)
    assert np.allclose(h * np.sign(h[:, 3:4]), g * np.sign(g[:, 3:4]), rtol=0, atol=2e-6 if tag == "short_baseline" else 1e-6)

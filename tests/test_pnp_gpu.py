"""Parity of the HIP 3-D/2-D registration (reference cv::solvePnPRansac with SOLVEPNP_EPNP at
cpp_code/src/estimate_motion.cpp:161-162) through the C ABI against the CPU oracle.

Both sides replay the same cv::RNG sample stream and bookkeeping: iteration counts and inlier masks must agree exactly.  The pose:
the EPnP re-fit on the inliers runs on the HOST in the oracle's summation order for inlier sets up to 8 192 (round 6; 1 024 before) --
every frame this pipeline can meet -- and is then BIT-IDENTICAL to the CPU restatement's, rvec and tvec included; beyond that the
device reductions take over and agree to 1e-7 (tests/stress_pnp.py has such sets)."""
import numpy as np
import pytest

import easysfm_amd as E
from easysfm_amd import synth

pytestmark = pytest.mark.gpu
K4 = np.array(synth.FOUNTAIN_K4, np.float32)


def _scene(rng, n, frac, noise=0.5):
    R = synth.aa_to_R(rng.normal(0, 0.3, 3)); t = np.array([0.3, -0.2, 6.0]) + rng.normal(0, 0.2, 3)
    X = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    Xc = X.astype(np.float64) @ R.T + t
    pix = np.stack([Xc[:, 0] / Xc[:, 2] * K4[0] + K4[1], Xc[:, 1] / Xc[:, 2] * K4[2] + K4[3]], 1) + rng.normal(0, noise, (n, 2))
    bad = rng.choice(n, int(frac * n), replace=False)
    pix[bad] += rng.uniform(-80, 80, (len(bad), 2))
    return X, pix.astype(np.float32), R, t, bad


@pytest.mark.parametrize("n,frac,iters,seed", [(600, 0.33, 50000, 0), (60, 0.5, 50000, 1), (3000, 0.2, 50000, 2), (25, 0.2, 100, 3), (6, 0.0, 100, 4)])
def test_solve_pnp_ransac_matches_oracle(gpu_ctx, oracle_lib, n, frac, iters, seed):
    rng = np.random.default_rng(seed)
    X, pix, R, t, bad = _scene(rng, n, frac)
    ok, Rr, tr, rvr, mr, itr = oracle_lib.solve_pnp_ransac(X, pix, K4, iters, 2.5, 0.99)
    assert ok
    rv, tv, Rg, mg, itg = E.solve_pnp_ransac(X, pix, K4, iters, 2.5, 0.99, gpu_ctx)
    assert itg == itr and np.array_equal(mg, mr)
    assert np.array_equal(Rg, Rr) and np.array_equal(tv, tr) and np.array_equal(rv, rvr)
    if n >= 60:
        assert np.allclose(Rg, R, atol=5e-3) and np.allclose(tv, t, atol=0.05) and mg[bad].sum() <= 0.05 * len(bad) + 2
        assert abs(np.linalg.det(Rg) - 1) < 1e-9


@pytest.mark.parametrize("n,frac,seed", [(800, 0.6, 21), (400, 0.65, 22), (1500, 0.55, 23)])
def test_solve_pnp_ransac_many_iterations_through_the_deferred_hypotheses(gpu_ctx, oracle_lib, n, frac, seed):
    """Hundreds of RANSAC iterations (55 - 65 % gross outliers): about one hypothesis in a hundred stalls in the 12 x 12 Jacobi
    diagonalisation; the first pass gives up on those after twelve sweeps and the replay has them solved in full when it reaches them
    (pnp_api.cpp) -- with this many iterations it does.  Iteration counts and masks exact, as everywhere."""
    rng = np.random.default_rng(seed)
    X, pix, R, t, bad = _scene(rng, n, frac)
    ok, Rr, tr, rvr, mr, itr = oracle_lib.solve_pnp_ransac(X, pix, K4, 50000, 2.5, 0.99)
    assert ok and itr > 150
    rv, tv, Rg, mg, itg = E.solve_pnp_ransac(X, pix, K4, 50000, 2.5, 0.99, gpu_ctx)
    assert itg == itr and np.array_equal(mg, mr)
    assert np.array_equal(Rg, Rr) and np.array_equal(tv, tr) and np.array_equal(rv, rvr)


def test_pnp_exactly_five_and_too_few(gpu_ctx, oracle_lib):
    rng = np.random.default_rng(9)
    X, pix, R, t, _ = _scene(rng, 5, 0.0, noise=0.0)
    rv, tv, Rg, mg, it = E.solve_pnp_ransac(X, pix, K4, 100, 2.5, 0.99, gpu_ctx)
    ok, Rr, tr, rvr, mr, itr = oracle_lib.solve_pnp_ransac(X, pix, K4, 100, 2.5, 0.99)
    assert ok and np.all(mg) and np.allclose(Rg, Rr, atol=1e-6) and np.allclose(Rg, R, atol=1e-4)
    with pytest.raises(E.EsfmError):
        E.solve_pnp_ransac(X[:4], pix[:4], K4, 100, 2.5, 0.99, gpu_ctx)


def test_mirror_estimate2D3D(gpu_ctx):
    """estimate2D3D_P3P_RANSAC (estimate_motion.cpp:99-232): id join, pose write-back, the reference's inlier bookkeeping."""
    rng = np.random.default_rng(13)
    X, pix, R, t, bad = _scene(rng, 300, 0.2)
    K = np.array([[K4[0], 0, K4[1]], [0, K4[2], K4[3]], [0, 0, 1]], np.float32)
    fr = E.Frame(frame_id=5, keypoints=np.concatenate([pix, rng.uniform(0, 700, (50, 2)).astype(np.float32)])); fr.K_cam = K
    fr.unique_pixel_ids = np.concatenate([np.arange(100, 400), np.arange(9000, 9050)])
    far = np.array([[500.0, 0, 0]], np.float32)                      # beyond the +-300 gate
    cloud = E.SparsePointCloud(xyz=np.concatenate([X, far]), rgb=np.zeros((301, 3), np.uint8),
                               unique_point_ids=np.concatenate([np.arange(100, 400), [9001]]), is_inlier=np.ones(301, np.int32))
    me = E.MotionEstimator(gpu_ctx)
    assert me.estimate2D3D_P3P_RANSAC(fr, cloud)
    assert fr.pose_cam.dtype == np.float32 and np.allclose(fr.pose_cam[:3, :3], R, atol=5e-3) and np.allclose(fr.pose_cam[:3, 3], t, atol=0.05)
    # every correspondence except the first is flagged is_inlier = 0; the gated far point is untouched
    assert cloud.is_inlier[0] == 1 and not cloud.is_inlier[1:300].any() and cloud.is_inlier[300] == 1

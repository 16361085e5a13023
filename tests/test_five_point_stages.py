"""The 5-point kernel's ONE arithmetic (cv::findEssentialMat at reference cpp_code/src/estimate_motion.cpp:49-51
runs EMEstimatorCallback::runKernel on every RANSAC sample): easysfm_amd/csrc/five_point_core.hpp and oracle/ransac_ref.c evaluate the same
expressions in the same order, so every stage -- null-space basis, determinant polynomial, B(z), root estimates, sweeps of the root
iteration, models -- agrees to the BIT.  Three legs:
  * the CPU restatement (oracle.five_point_stages / five_point),
  * the HOST build of the kernels' routines (esfm_five_point_models_host, no GPU) -- compared here on CPU,
  * the kernels themselves (esfm_five_point_models) -- compared under -m gpu.
Until round 5 the two sides took different routes to the same models (to ~1e-8), and 2.2 % of random RANSAC problems were decided
differently by threshold-borderline correspondences (tests/stress_essential.py)."""
import numpy as np
import pytest

import easysfm_amd as E
from easysfm_amd import synth

STAGE = {"N": (0, 36), "det": (36, 47), "P": (47, 59), "Qp": (59, 71), "R": (71, 86), "cc": (86, 96), "re": (96, 106), "im": (106, 116), "sweeps": (116, 117)}


def samples(seed, n_problems, per_problem=64):
    """5-point samples the way RANSAC meets them: drawn from two-view problems with noise and gross outliers (general and planar
    scenes, small and large rotations and baselines), plus degenerate ones (a repeated correspondence, collinear points, pure rotation)."""
    rng = np.random.default_rng(seed)
    q1s, q2s = [], []
    for _ in range(n_problems):
        n = 200
        R = synth.aa_to_R(rng.normal(0, rng.choice([0.02, 0.15, 0.4]), 3)); t = rng.normal(0, 1, 3); t /= np.linalg.norm(t); t *= rng.choice([0.0, 0.1, 1.0])
        X = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0, 8.0])
        if rng.random() < 0.15: X[:, 2] = 8.0
        Xc = X @ R.T + t
        a = X[:, :2] / X[:, 2:3] + rng.normal(0, rng.choice([0.0, 0.0005]), (n, 2)); b = Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, 0.0005, (n, 2))
        bad = rng.choice(n, int(0.3 * n), replace=False); b[bad] += rng.uniform(-0.1, 0.1, (len(bad), 2))
        for _ in range(per_problem):
            id5 = rng.choice(n, 5, replace=False)
            q1, q2 = a[id5].copy(), b[id5].copy()
            kind = rng.random()
            if kind < 0.02: q1[4] = q1[0]; q2[4] = q2[0]                                  # a repeated correspondence
            elif kind < 0.04: q1[:, 1] = 0.3 * q1[:, 0] + 0.1                             # collinear in the first image
            elif kind < 0.05: q1[:] = np.float32(q1); q2[:] = np.float32(q2)              # few significant bits
            q1s.append(q1); q2s.append(q2)
    return np.array(q1s), np.array(q2s)


def compare(st_a, Es_a, nm_a, st_b, Es_b, nm_b, what):
    """bit for bit, stage after stage: the first stage that differs names itself"""
    for name, (lo, hi) in STAGE.items():
        same = (st_a[:, lo:hi] == st_b[:, lo:hi]) | (np.isnan(st_a[:, lo:hi]) & np.isnan(st_b[:, lo:hi]))
        assert same.all(), f"{what}: stage {name} differs in {np.count_nonzero(~same.all(axis=1))} of {len(st_a)} samples (first: {int(np.argmin(same.all(axis=1)))})"
    assert np.array_equal(nm_a, nm_b), what
    for k in range(len(nm_a)):
        assert np.array_equal(Es_a[k, :nm_a[k]], Es_b[k, :nm_b[k]], equal_nan=True), f"{what}: models of sample {k}"


def oracle_side(oracle_lib, q1, q2):
    st = np.array([oracle_lib.five_point_stages(a, b) for a, b in zip(q1, q2)])
    Es = np.zeros((len(q1), 10, 3, 3)); nm = np.zeros(len(q1), np.int32)
    for k, (a, b) in enumerate(zip(q1, q2)):
        m = oracle_lib.five_point(a, b)
        nm[k] = len(m); Es[k, :len(m)] = m
    return st, Es, nm


def test_host_build_equals_oracle_bit_for_bit(oracle_lib):
    q1, q2 = samples(101, 120)
    Es_h, nm_h, st_h = E.five_point_models(q1, q2, host=True, stages=True)
    st_o, Es_o, nm_o = oracle_side(oracle_lib, q1, q2)
    compare(st_h, Es_h, nm_h, st_o, Es_o, nm_o, "host build vs oracle")
    assert (nm_o > 0).mean() > 0.9 and 10 < st_o[st_o[:, 116] >= 0, 116].mean() < 60      # (the sweep is not about degenerate samples only)


def test_models_satisfy_their_constraints(oracle_lib):
    """known answer: noise-free samples of a true pose have it among their models; every model is a unit-norm essential matrix through
    the sample"""
    rng = np.random.default_rng(5)
    for _ in range(50):
        R = synth.aa_to_R(rng.normal(0, 0.3, 3)); t = rng.normal(0, 1, 3); t /= np.linalg.norm(t)
        X = rng.uniform(-2, 2, (5, 3)) + np.array([0, 0, 6.0]); Xc = X @ R.T + t
        q1 = X[:, :2] / X[:, 2:3]; q2 = Xc[:, :2] / Xc[:, 2:3]
        Egt = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R; Egt /= np.linalg.norm(Egt)
        Es, nm = E.five_point_models(q1[None], q2[None], host=True)
        Es = Es[0, :nm[0]]
        assert 1 <= nm[0] <= 10 and min(min(np.abs(M - Egt).max(), np.abs(M + Egt).max()) for M in Es) < 1e-7
        h1 = np.c_[q1, np.ones(5)]; h2 = np.c_[q2, np.ones(5)]
        for M in Es:
            assert abs(np.linalg.norm(M) - 1) < 1e-12 and M.ravel()[np.argmax(np.abs(M))] > 0
            assert np.abs(np.einsum("ni,ij,nj->n", h2, M, h1)).max() < 1e-9
        assert np.all(np.diff(Es[:, 0, 0]) >= 0)


@pytest.mark.gpu
def test_kernels_equal_host_build_and_oracle_bit_for_bit(gpu_ctx, oracle_lib):
    q1, q2 = samples(202, 160)
    Es_g, nm_g, st_g = E.five_point_models(q1, q2, gpu_ctx, stages=True)
    Es_h, nm_h, st_h = E.five_point_models(q1, q2, host=True, stages=True)
    compare(st_g, Es_g, nm_g, st_h, Es_h, nm_h, "kernels vs host build")
    st_o, Es_o, nm_o = oracle_side(oracle_lib, q1[:2000], q2[:2000])
    compare(st_g[:2000], Es_g[:2000], nm_g[:2000], st_o, Es_o, nm_o, "kernels vs oracle")

"""Host-side planning of the structure-aware reduced camera solve (esfm_ba_reduced_plan; easysfm_amd/csrc/ba_sparse_plan.cpp), no GPU.

Block (a, b) of the reduced camera system is structurally non-zero only if cameras a and b observe a common point (the reference
adds one residual block per observation, cpp_code/src/ba.cpp:140-151).  The plan orders the cameras by nested dissection and lists
the 64 x 64 tiles of the symbolic fill; the kernels skip everything else, so the plan is correct iff the Cholesky factor of ANY
matrix with that block structure, permuted and padded as the plan says, is zero outside the listed tiles -- checked here with
numpy's dense Cholesky on random SPD matrices."""
import numpy as np
import pytest

import easysfm_amd as E
from easysfm_amd import synth


def _covis(n_cam, cam_idx, pt_idx):
    A = np.eye(n_cam, dtype=bool)
    order = np.argsort(pt_idx, kind="stable")
    c, p = cam_idx[order], pt_idx[order]
    cuts = np.nonzero(np.diff(p))[0] + 1
    for grp in np.split(c, cuts):
        A[np.ix_(grp, grp)] = True
    return A


def _check_plan(n_cam, n_pt, cam_idx, pt_idx, leaf_max=0, seed=0):
    plan = E.reduced_plan(n_cam, n_pt, cam_idx, pt_idx, leaf_max)
    nb, col_src, tiles = plan["nb"], plan["col_src"], plan["tiles"]
    assert len(col_src) == 64 * nb
    real = col_src[col_src >= 0]
    assert np.array_equal(np.sort(real), np.arange(6 * n_cam))                 # every unknown exactly once
    pos = np.nonzero(col_src >= 0)[0]
    assert np.all(col_src[pos[::6]] % 6 == 0) and np.all(np.diff(pos.reshape(-1, 6), axis=1) == 1)    # a camera's six columns stay together, in order
    assert np.all(np.diff(col_src[pos].reshape(-1, 6), axis=1) == 1)
    # random SPD matrix with the co-visibility block structure, permuted and padded
    A = _covis(n_cam, cam_idx, pt_idx)
    rng = np.random.default_rng(seed)
    n = 6 * n_cam
    M = rng.standard_normal((n, n)) * np.kron(A, np.ones((6, 6)))
    M = M + M.T
    M += np.eye(n) * (np.abs(M).sum(axis=1).max() + 1.0)
    W = np.eye(64 * nb)
    W[np.ix_(pos, pos)] = M[np.ix_(col_src[pos], col_src[pos])]
    Lf = np.linalg.cholesky(W)
    T = np.abs(Lf).reshape(nb, 64, nb, 64).max(axis=(1, 3)) > 0
    listed = np.zeros((nb, nb), bool)
    fac = tiles[tiles[:, 0] < nb]
    listed[fac[:, 0], fac[:, 1]] = True
    assert np.all(fac[:, 0] >= fac[:, 1]) and listed.diagonal().all()
    assert not (T & ~listed).any(), f"factor has entries outside the plan's tiles: {np.argwhere(T & ~listed)[:5]}"
    rhs = tiles[tiles[:, 0] == nb]
    assert np.array_equal(np.sort(rhs[:, 1]), np.arange(nb))                   # one right-hand-side tile per column
    # the chain is a property of the listed structure
    fin = np.zeros(nb, int)
    for j in range(nb):
        ks = np.nonzero(listed[j, :j])[0]
        fin[j] = 1 + (fin[ks].max() if len(ks) else 0)
    assert plan["chain"] == fin.max()
    return plan, int(T.sum())


@pytest.mark.parametrize("n_cam,n_pt,k", [(64, 1500, 5), (107, 3000, 6), (200, 6000, 8)])
def test_ring_scenes(n_cam, n_pt, k):
    sc = synth.ba_scene(n_cam, n_pt, k, seed=n_cam)
    plan, used = _check_plan(n_cam, n_pt, sc.cam_idx, sc.pt_idx, seed=n_cam)
    assert plan["chain"] <= plan["nb"]
    if n_cam >= 200:
        assert plan["chain"] <= 8 and plan["worthwhile"]


def test_config5_ring_is_sparse_and_shallow():
    """BA-512's structure (every point seen by 10 consecutive cameras of a closed loop of 512; the observation list of
    synth.ba_scene(512, 300000, 10, seed=5000) without the geometry): 4 % of the camera blocks, a chain of 8 tile columns where the
    dense factorisation has 48, fewer than a quarter of its tiles."""
    rng = np.random.default_rng(np.random.PCG64(5000))
    n_cam, n_pt, k = 512, 300000, 10
    rng.uniform(-8, 8, size=(n_pt, 3))
    start = rng.integers(0, n_cam, size=n_pt)
    cam_idx = ((start[:, None] + np.arange(k)[None, :]) % n_cam).astype(np.int32).reshape(-1)
    pt_idx = np.repeat(np.arange(n_pt, dtype=np.int32), k)
    plan = E.reduced_plan(n_cam, n_pt, cam_idx, pt_idx)
    assert plan["dense_nb"] == 48 and plan["worthwhile"] and plan["chain"] <= 10
    assert 4 * len(plan["tiles"]) <= 48 * 49 // 2 + 48
    sub = np.nonzero(pt_idx < 20000)[0]                                        # (the structure check on a thinner copy of the same loop)
    _check_plan(n_cam, 20000, cam_idx[sub], pt_idx[sub], seed=3)


@pytest.mark.parametrize("leaf_max", [4, 16, 64])
def test_random_sparse_graphs(leaf_max):
    """Irregular co-visibility: short tracks over random nearby cameras plus a few long-range tracks (loop closures), isolated
    cameras, two disconnected groups."""
    rng = np.random.default_rng(leaf_max)
    n_cam, n_pt = 90, 700
    cam, pt = [], []
    for p in range(n_pt):
        grp = 0 if p % 2 else 45                                                # two groups of 45 cameras that share nothing
        c0 = rng.integers(0, 40)
        cams = np.unique(np.clip(c0 + rng.integers(-3, 4, size=rng.integers(2, 6)), 0, 41)) + grp
        if p % 97 == 0:
            cams = np.unique(np.concatenate([cams, [grp + rng.integers(0, 42)]]))   # loop closure
        cam += list(cams); pt += [p] * len(cams)
    cam = np.array(cam, np.int32); pt = np.array(pt, np.int32)                  # cameras 42-44 and 87-89 see nothing
    plan, used = _check_plan(n_cam, n_pt, cam, pt, leaf_max, seed=leaf_max)
    assert plan["supernodes"] >= 2


def test_dense_covisibility_is_left_to_the_dense_path():
    """Everything sees everything (BA-25, the fountain): one leaf, nothing to gain."""
    sc = synth.ba_scene(40, 500, 40, seed=2)
    plan, _ = _check_plan(40, 500, sc.cam_idx, sc.pt_idx)
    assert plan["supernodes"] == 1 and not plan["worthwhile"] and plan["nb"] == plan["dense_nb"]


def test_degenerate_inputs():
    plan = E.reduced_plan(0, 0, np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert plan["nb"] == 0 and not plan["worthwhile"]
    plan, _ = _check_plan(3, 1, np.zeros(0, np.int32), np.zeros(0, np.int32))   # no observation at all: three isolated cameras share one leaf
    assert plan["nb"] == 1 and plan["chain"] == 1 and plan["supernodes"] == 1
    # a hundred cameras of which only the first forty see anything (a loop): the sixty others are packed into leaves of 32, not a tile each
    sc = synth.ba_scene(40, 400, 5, seed=9)
    plan, _ = _check_plan(100, 400, sc.cam_idx, sc.pt_idx, seed=1)
    assert plan["nb"] <= (6 * 100 + 63) // 64 + plan["supernodes"]

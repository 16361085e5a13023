"""Parity of the HIP statistical-outlier-removal path (reference CProceesing::SORFilter,
cpp_code/include/cloudprocessing.hpp:24-36) through the C ABI: bit-exact mean distances, threshold and keep mask against
the committed golden vectors and against the CPU oracle on larger seeded clouds."""
import os

import numpy as np
import pytest

import easysfm_amd as E

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("tag", ["blobs", "dups_nonfinite", "small", "grid_ties", "stride8"])
def test_sor_golden_bitexact(gpu_ctx, tag):
    z = np.load(os.path.join(GOLD, "sor_cases.npz"))
    keep, md, thr = E.sor_filter(z[f"{tag}.points"], int(z[f"{tag}.mean_k"]), float(z[f"{tag}.std_mul"]), gpu_ctx)
    assert np.array_equal(md, z[f"{tag}.mean_dist"])
    assert thr == float(z[f"{tag}.threshold"])
    assert np.array_equal(keep, z[f"{tag}.keep"])


@pytest.mark.parametrize("n,k,seed", [(1, 50, 0), (2, 1, 1), (63, 50, 2), (64, 63, 3), (65, 50, 4), (1023, 50, 5), (1025, 7, 6),
                                      (20000, 50, 7)])
def test_sor_matches_oracle(gpu_ctx, oracle_lib, n, k, seed):
    rng = np.random.default_rng(seed)
    P = np.concatenate([rng.normal(0, 1, (n - n // 10, 3)), rng.uniform(-20, 20, (n // 10, 3))]).astype(np.float32)
    if n >= 64:
        P[5] = P[6]                       # a duplicate pair
        P[n // 2, 1] = np.nan             # and a non-finite point
    keep, md, thr = E.sor_filter(P, k, 2.0, gpu_ctx)
    rkeep, rmd, rthr = oracle_lib.sor_filter(P, k, 2.0)
    assert np.array_equal(md, rmd)
    assert (thr == rthr) or (np.isnan(thr) and np.isnan(rthr))
    assert np.array_equal(keep, rkeep)


def test_sor_full_size_properties(gpu_ctx):
    """BA-512-sized cloud (300 000 points): properties that need no O(N^2) CPU reference --
    (i) permutation equivariance (the k-NN multiset of a point does not depend on the storage order, and the wave's
        sorted-lane state makes the summation order canonical), (ii) a spot-check of 64 random points against numpy."""
    rng = np.random.default_rng(11)
    n = 300000
    P = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
    keep, md, thr = E.sor_filter(P, 50, 2.0, gpu_ctx)
    perm = rng.permutation(n)
    keep2, md2, thr2 = E.sor_filter(P[perm], 50, 2.0, gpu_ctx)
    assert np.array_equal(md2, md[perm])
    assert np.array_equal(keep2, keep[perm])
    for i in rng.integers(0, n, 64):
        d = P[i][None, :] - P
        d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)
        row = np.sort(d2)[1:51]
        ref = np.float32(np.cumsum(np.sqrt(row).astype(np.float64))[-1] / 50)
        assert md[i] == ref
    assert 0.9 * n < keep.sum() <= n


def _sor_clouds():
    rng = np.random.default_rng(7)
    out = {}
    n = 100000
    c = rng.uniform(-20, 20, (40, 3))
    P = (c[rng.integers(0, 40, n)] + rng.normal(0, 0.3, (n, 3)) * rng.uniform(0.2, 3, (n, 1))).astype(np.float32)
    P[:2000] = rng.uniform(-60, 60, (2000, 3)).astype(np.float32)            # far outliers: the points the filter exists for
    out["clustered"] = P
    P = rng.uniform(-5, 5, (50000, 3)).astype(np.float32); P[:, 0] = 1.25      # no extent along x: the sort axis must be another one
    out["plane"] = P
    P = np.zeros((20000, 3), np.float32); P[:, 1] = rng.uniform(-5, 5, 20000)   # one axis only, thousands of exactly equal keys elsewhere
    out["line"] = P
    P = np.repeat(rng.uniform(-1, 1, (300, 3)).astype(np.float32), 40, axis=0)  # every point 40 times: the window's edge falls inside ties
    out["duplicates"] = P
    out["all_equal"] = np.full((5000, 3), 0.75, np.float32)                     # every key and every distance equal: T = 0 against an edge of 0
    P = rng.uniform(-3, 3, (4096, 3)).astype(np.float32)                        # the smallest cloud that takes the sorted path, ...
    P[40:] = np.nan                                                             # ... with fewer finite points than neighbours asked for
    out["mostly_nan"] = P
    P = rng.uniform(-1, 1, (6000, 3)).astype(np.float32); P[:, 2] *= 1e-30      # denormal-scale spread along one axis
    out["flat"] = P
    for P in out.values():
        P[17] = [np.nan, 0, 0]; P[123, 2] = np.inf                              # non-finite points are neither queries nor candidates
    return out


def test_sor_sorted_window_equals_all_candidates(gpu_ctx):
    """From 4096 points on the k-NN pass sweeps a window of the cloud sorted along its longest axis instead of every point (round 3).
    The distances it keeps must be the brute-force ones bit for bit on any cloud: checked against a float32 numpy evaluation in the
    kernel's operation order on sampled points of clustered / planar / collinear / duplicated clouds with non-finite entries, and
    against the all-candidates kernel itself (ESFM_SOR_BRUTE=1, a fresh process) on every point."""
    import subprocess, sys, tempfile
    clouds = _sor_clouds()
    rng = np.random.default_rng(3)
    got = {}
    for name, P in clouds.items():
        keep, md, thr = E.sor_filter(P, 50, 2.0, gpu_ctx)
        got[name] = md
        fin = np.isfinite(P).all(1)
        assert np.all(md[~fin] == 0)                                               # (non-finite points: distance 0, as in PCL)
        Pf = P[fin]
        for i in rng.choice(np.nonzero(fin)[0], min(48, int(fin.sum())), replace=False):
            d = P[i][None, :] - Pf
            d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)
            row = np.sort(d2)[1:51]
            assert md[i] == np.float32(np.cumsum(np.sqrt(row).astype(np.float64))[-1] / 50), (name, int(i))
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), **clouds)
        code = ("import sys, numpy as np; sys.path.insert(0, %r); import easysfm_amd as E; z = np.load(%r); ctx = E.Context(0); "
                "np.savez(%r, **{k: E.sor_filter(z[k], 50, 2.0, ctx)[1] for k in z.files})"
                % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(td, "in.npz"), os.path.join(td, "out.npz")))
        subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, ESFM_SOR_BRUTE="1"))
        ref = np.load(os.path.join(td, "out.npz"))
        for name in clouds:
            assert np.array_equal(got[name].view(np.uint32), ref[name].view(np.uint32)), name


def test_sor_mirror_and_ply_roundtrip(gpu_ctx, tmp_path):
    """CProceesing.SORFilter + DataIO.writePlyFile as sfm.cpp:333-337 chains them."""
    rng = np.random.default_rng(2)
    xyz = np.concatenate([rng.normal(0, 1, (800, 3)), rng.uniform(-30, 30, (20, 3))]).astype(np.float32)
    rgb = rng.integers(0, 256, (820, 3)).astype(np.uint8)
    cloud = E.SparsePointCloud(xyz=xyz, rgb=rgb)
    out = E.CProceesing(gpu_ctx).SORFilter(cloud)
    keep, _, _ = E.sor_filter(xyz, 50, 2.0, gpu_ctx)
    assert np.array_equal(out.xyz, xyz[keep]) and np.array_equal(out.rgb, rgb[keep]) and 780 <= len(out.xyz) < 820
    path = str(tmp_path / "sfm.ply")
    assert E.write_ply(path, out)
    x2, c2, cam = E.read_ply_vertices(path)
    # 8 significant digits (PCL's stream precision) do not round-trip every float32: 1e-7 relative
    assert np.allclose(x2, out.xyz, rtol=1e-7, atol=0) and np.array_equal(c2, out.rgb)


def test_sor_argument_errors(gpu_ctx):
    P = np.zeros((10, 3), np.float32)
    with pytest.raises(E.EsfmError):
        E.sor_filter(P, 64, 2.0, gpu_ctx)      # mean_k + 1 must fit one wave
    with pytest.raises(E.EsfmError):
        E.sor_filter(P, 0, 2.0, gpu_ctx)
    keep, md, thr = E.sor_filter(np.zeros((0, 3), np.float32), 50, 2.0, gpu_ctx)
    assert len(keep) == 0

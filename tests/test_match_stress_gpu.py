"""Randomised parity of the matchers against the oracle: ragged sizes from one row to several LDS tiles and segments, duplicated
rows (re-scan path), near ties, unnormalised rows; then ragged batched pair lists (per-pair binning of uncertified queries,
chunked pair re-scan, ratio compaction).  The asynchronous row transfers of the L2 tail and of the re-scan (LDS-DMA into
landing zones) have no other witness than results: this test is their race detector."""
import numpy as np
import pytest

import easysfm_amd as E
from easysfm_amd import synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_single_pairs(gpu_ctx, oracle_lib, seed):
    rng = np.random.default_rng(seed)
    for it in range(160):
        big = it % 10 == 9
        nq = int(rng.integers(1, 1500 if big else 700)); nt = int(rng.integers(1, 6000 if big else 1500))
        mode = it % 4
        q = rng.standard_normal((nq, 64)).astype(np.float32); t = rng.standard_normal((nt, 64)).astype(np.float32)
        if mode == 1:     # duplicated rows: more equal distances than the pass keeps candidates
            t = t[rng.integers(0, max(nt // 8, 1), nt)]; q[: nq // 2] = t[rng.integers(0, nt, nq // 2)]
        if mode == 2:     # near ties
            t = t[rng.integers(0, max(nt // 4, 1), nt)] * (1 + 1e-6 * rng.standard_normal((nt, 1)).astype(np.float32))
        if mode != 3:     # mode 3: unnormalised rows
            q /= np.linalg.norm(q, axis=1, keepdims=True); t /= np.linalg.norm(t, axis=1, keepdims=True)
        idx, dist = E.knn_match_l2(q, t, gpu_ctx)
        ridx, rdist = oracle_lib.knn2_l2(q, t)
        assert np.array_equal(idx, ridx) and np.array_equal(_bits(dist), _bits(rdist)), (seed, it, nq, nt, mode)
        qb = rng.integers(0, 256, (nq, 32), dtype=np.uint8); tb = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        if mode == 1:
            tb = tb[rng.integers(0, max(nt // 8, 1), nt)]; qb[: nq // 2] = tb[rng.integers(0, nt, nq // 2)]
        hi, hd = E.knn_match_hamming(qb, tb, gpu_ctx)
        rhi, rhd = oracle_lib.knn2_hamming(qb, tb)
        assert np.array_equal(hi, rhi) and np.array_equal(hd, rhd), (seed, it, nq, nt, mode)


def test_random_pair_lists(gpu_ctx, oracle_lib):
    rng = np.random.default_rng(7)
    for rep in range(8):
        nsets = int(rng.integers(2, 30))
        pool = rng.standard_normal((64, 64)).astype(np.float32)
        sets = []
        for _ in range(nsets):
            n = int(rng.integers(0, 400))
            x = rng.standard_normal((n, 64)).astype(np.float32)
            if n:
                m = rng.random(n) < 0.3
                x[m] = pool[rng.integers(0, 64, int(m.sum()))]
                x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
            sets.append(np.ascontiguousarray(x))
        pairs = synth.all_pairs(nsets)
        pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), pairs)
        res = pm.match(0.8).to_host()
        for (i, j), (qi, ti, d) in zip(pairs, res):
            if len(sets[i]) == 0:
                assert len(qi) == 0
                continue
            rq, rt, rd = oracle_lib.match_l2(sets[i], sets[j], 0.8)
            assert np.array_equal(qi, rq) and np.array_equal(ti, rt) and np.array_equal(_bits(d), _bits(rd)), (rep, i, j)

"""Parity of the HIP matching path (through the C ABI) against the CPU oracle.  Bit-exact: indices
AND distances (float bit patterns)."""
import numpy as np
import pytest

import easysfm_amd as E
from easysfm_amd import synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _check_knn_l2(ctx, oracle, q, t):
    idx, dist = E.knn_match_l2(q, t, ctx)
    ridx, rdist = oracle.knn2_l2(q, t)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(_bits(dist), _bits(rdist))


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (5, 3), (33, 31), (64, 64), (129, 200), (500, 1000), (1000, 777)])
def test_l2_knn_bitexact_ragged(gpu_ctx, oracle_lib, nq, nt):
    rng = np.random.default_rng(nq * 1000 + nt)
    q = rng.standard_normal((nq, 64)).astype(np.float32)
    t = rng.standard_normal((nt, 64)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True); t /= np.linalg.norm(t, axis=1, keepdims=True)
    _check_knn_l2(gpu_ctx, oracle_lib, q, t)


def test_l2_empty_and_tiny_train(gpu_ctx, oracle_lib):
    q = np.random.default_rng(0).standard_normal((10, 64)).astype(np.float32)
    for nt in (0, 1):
        t = np.random.default_rng(1).standard_normal((nt, 64)).astype(np.float32)
        idx, dist = E.knn_match_l2(q, t, gpu_ctx)
        ridx, rdist = oracle_lib.knn2_l2(q, t)
        assert np.array_equal(idx, ridx) and np.array_equal(_bits(dist), _bits(rdist))
        qi, ti, d = E.match_l2(q, t, 0.9, gpu_ctx)
        assert len(qi) == 0  # reference is UB here; defined as "emit nothing"
    qi, ti, d = E.match_l2(np.zeros((0, 64), np.float32), q, 0.5, gpu_ctx)
    assert len(qi) == 0


def test_l2_duplicates_and_ties(gpu_ctx, oracle_lib):
    """Exact duplicates in the train set (distance-0 ties and equal-distance ties): the lower train
    index must win; more duplicates than the kernel keeps candidates forces the exact re-scan."""
    rng = np.random.default_rng(7)
    base = rng.standard_normal((50, 64)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    t = np.concatenate([base, base[:20], base[:20], base[:10], base[:10], base[:10], base[:10], base[:10]], axis=0)
    perm = rng.permutation(len(t)); t = t[perm]
    q = np.concatenate([base[:30], base[:30] + 1e-4 * rng.standard_normal((30, 64)).astype(np.float32)], axis=0)
    _check_knn_l2(gpu_ctx, oracle_lib, q, t)


def test_l2_near_ties(gpu_ctx, oracle_lib):
    """Train rows that differ from each other by a few ulps: arg-min and the second neighbour must
    follow the oracle's summation order, not the GEMM's."""
    rng = np.random.default_rng(9)
    q = rng.standard_normal((200, 64)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = np.repeat(q[:100], 6, axis=0)
    t = t * (1.0 + rng.integers(-3, 4, size=t.shape) * np.float32(6e-8))
    t = np.concatenate([t, rng.standard_normal((300, 64)).astype(np.float32) * 0.1], axis=0).astype(np.float32)
    _check_knn_l2(gpu_ctx, oracle_lib, q, t[rng.permutation(len(t))])


def test_l2_unnormalised_and_dim128(gpu_ctx, oracle_lib):
    rng = np.random.default_rng(11)
    q = (rng.standard_normal((300, 64)) * rng.uniform(0.1, 30, (300, 1))).astype(np.float32)
    t = (rng.standard_normal((400, 64)) * rng.uniform(0.1, 30, (400, 1))).astype(np.float32)
    _check_knn_l2(gpu_ctx, oracle_lib, q, t)
    q = rng.standard_normal((150, 128)).astype(np.float32); t = rng.standard_normal((333, 128)).astype(np.float32)
    _check_knn_l2(gpu_ctx, oracle_lib, q, t)     # extended SURF
    q = rng.standard_normal((40, 36)).astype(np.float32); t = rng.standard_normal((90, 36)).astype(np.float32)
    _check_knn_l2(gpu_ctx, oracle_lib, q, t)     # no MFMA build: exact-scan path
    q = rng.standard_normal((40, 37)).astype(np.float32); t = rng.standard_normal((90, 37)).astype(np.float32)
    _check_knn_l2(gpu_ctx, oracle_lib, q, t)     # odd width: scalar tail of the canonical sum


@pytest.mark.parametrize("case", ["cluster", "dynamic_range", "tiny", "denormal", "denormal_mixed", "huge_norms", "equal_rows", "sparse", "segment_edges"])
def test_l2_split_bf16_pass_adversarial(gpu_ctx, oracle_lib, case):
    """64-float descriptors go through the split-bf16 distance pass, whose scores carry ~2^-16 relative error: inputs built to
    sit inside that error (near-equal distances, cancellation, extreme magnitudes) must still come out bit-identical to the
    oracle, through the certificate's rescan if need be."""
    rng = np.random.default_rng({"cluster": 1, "dynamic_range": 2, "tiny": 3, "huge_norms": 4, "equal_rows": 5, "sparse": 6, "denormal": 8, "denormal_mixed": 9,
                                 "segment_edges": 7}[case])
    nq, nt = 300, 1500
    q = rng.standard_normal((nq, 64)).astype(np.float32)
    t = rng.standard_normal((nt, 64)).astype(np.float32)
    if case == "cluster":                   # every train within 1e-5 relative of one centre: all distances nearly equal
        c = rng.standard_normal(64).astype(np.float32)
        t = (c[None, :] * (1 + 1e-5 * rng.standard_normal((nt, 64)))).astype(np.float32)
        q[:150] = (c[None, :] * (1 + 1e-5 * rng.standard_normal((150, 64)))).astype(np.float32)
    elif case == "dynamic_range":           # components spread over 12 decades inside a row
        q = (q * np.exp(rng.uniform(-14, 14, q.shape))).astype(np.float32)
        t = (t * np.exp(rng.uniform(-14, 14, t.shape))).astype(np.float32)
    elif case == "tiny":                    # magnitudes near the bottom of the normal range
        q = (q * 1e-18).astype(np.float32); t = (t * 1e-18).astype(np.float32)
    elif case == "denormal":                # |row|^2 and the bf16 residuals' squares are denormal or zero in f32
        q = (q * 1e-20).astype(np.float32); t = (t * 1e-20).astype(np.float32)
    elif case == "denormal_mixed":          # rows from 1e-23 (squares flush to zero) to 1e-17 (normal) in one train set
        q = (q * np.float32(10.0) ** rng.integers(-23, -16, (nq, 1))).astype(np.float32)
        t = (t * np.float32(10.0) ** rng.integers(-23, -16, (nt, 1))).astype(np.float32)
    elif case == "huge_norms":              # large common offset: d^2 << |q|^2 + |t|^2 (catastrophic cancellation in the GEMM form)
        q = (q + 300.0).astype(np.float32); t = (t + 300.0).astype(np.float32)
    elif case == "equal_rows":              # many identical trains and queries equal to trains
        t[100:400] = t[7]; t[900:] = t[13]; q[:50] = t[7]; q[50:100] = t[13]
    elif case == "sparse":
        q[rng.random(q.shape) < 0.9] = 0; t[rng.random(t.shape) < 0.9] = 0
    elif case == "segment_edges":           # the two nearest trains at the seams of the 32-row steps / 512-row segments / 128-row tiles
        nt = 2100
        t = rng.standard_normal((nt, 64)).astype(np.float32)
        for k, pos in enumerate([0, 31, 32, 127, 128, 511, 512, 1023, 1024, 2047, 2048, 2099]):
            q[k] = t[pos] * np.float32(1 + 1e-4)
            q[k + 20] = t[pos]; t[(pos + 1) % nt] = t[pos] * np.float32(1 + 3e-7)
    _check_knn_l2(gpu_ctx, oracle_lib, q, t)
    qi, ti, d = E.match_l2(q, t, 0.8, gpu_ctx)
    rq, rt, rd = oracle_lib.match_l2(q, t, 0.8)
    assert np.array_equal(qi, rq) and np.array_equal(ti, rt) and np.array_equal(_bits(d), _bits(rd))


@pytest.mark.parametrize("ratio", [0.5, 0.7, 0.8, 1.0])
def test_l2_match_surf_like(gpu_ctx, oracle_lib, ratio):
    s = synth.surf_like_sets(2, 1500, pool=2048, seed_base=100)
    qi, ti, d = E.match_l2(s[1], s[0], ratio, gpu_ctx)
    rq, rt, rd = oracle_lib.match_l2(s[1], s[0], ratio)
    assert len(rq) > 0
    assert np.array_equal(qi, rq) and np.array_equal(ti, rt) and np.array_equal(_bits(d), _bits(rd))


@pytest.mark.parametrize("nq,nt", [(1, 1), (3, 2), (64, 65), (257, 300), (1000, 999)])
def test_hamming_knn_bitexact(gpu_ctx, oracle_lib, nq, nt):
    rng = np.random.default_rng(nq + 31 * nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8); t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    idx, dist = E.knn_match_hamming(q, t, gpu_ctx)
    ridx, rdist = oracle_lib.knn2_hamming(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)


def test_hamming_forced_ties(gpu_ctx, oracle_lib):
    """All-equal rows and few distinct distances: ties for first AND second place everywhere."""
    rng = np.random.default_rng(5)
    t = np.zeros((300, 32), np.uint8)
    t[::3, 0] = 1; t[1::3, 1] = 3
    q = np.zeros((100, 32), np.uint8); q[50:, 5] = 0xFF
    idx, dist = E.knn_match_hamming(q, t, gpu_ctx)
    ridx, rdist = oracle_lib.knn2_hamming(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    for ratio in (0.5, 0.8, 1.0, 2.0):   # d0 == ratio*d1 must be rejected (strict <)
        a = E.match_hamming(q, t, ratio, gpu_ctx); b = oracle_lib.match_hamming(q, t, ratio)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    for nb in (16, 64):
        q2 = rng.integers(0, 256, (77, nb), dtype=np.uint8); t2 = rng.integers(0, 256, (130, nb), dtype=np.uint8)
        idx, dist = E.knn_match_hamming(q2, t2, gpu_ctx)
        ridx, rdist = oracle_lib.knn2_hamming(q2, t2)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)


@pytest.mark.parametrize("case", ["dups_everywhere", "dups_late", "all_ones", "one_bit_apart", "ragged_batch"])
def test_hamming_fp4_adversarial(gpu_ctx, oracle_lib, case):
    """The FP4-MFMA form of the 256-bit matcher (hamming_fp4_kernel): exact scores, so the only thing to get wrong is WHICH rows
    the tail looks at -- ties for first and second place spread over more fold groups than a lane keeps keys, duplicates whose
    lowest indices sit late in the set, all-ones rows (the largest scores), neighbours one bit apart, and a ragged pair list
    (train sets of 1, 7, 300, 513 and 2300 rows: fewer rows than a ring tile, partial last tiles)."""
    rng = np.random.default_rng(21)
    ratios = (0.5, 0.8, 1.0)
    if case == "ragged_batch":
        sizes = [1, 7, 300, 513, 2300, 40]
        sets = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for n in sizes]
        sets[4][:200] = sets[2][:200]; sets[4][200:260, 3] ^= 1          # exact copies and one-bit neighbours across sets
        pairs = synth.all_pairs(len(sizes))
        pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_HAMMING), pairs)
        for ratio in ratios:
            res = pm.match(ratio).to_host()
            for (i, j), (qi, ti, d) in zip(pairs, res):
                rq, rt, rd = oracle_lib.match_hamming(sets[i], sets[j], ratio)
                assert np.array_equal(qi, rq) and np.array_equal(ti, rt) and np.array_equal(d, rd), (i, j, ratio)
        return
    nt = 3000
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    base = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    if case == "dups_everywhere":
        t[rng.choice(nt, 400, replace=False)] = base                    # 400 copies: ties for both places in dozens of groups
    elif case == "dups_late":
        t[[2990, 2995, 1501, 2047, 2048]] = base
    elif case == "all_ones":
        t[::7] = 0xFF
    elif case == "one_bit_apart":
        for k in range(0, nt, 5):
            t[k] = base; t[k, (k // 5) % 32] ^= np.uint8(1 << (k % 8))
    q = np.concatenate([base, base ^ np.uint8(1), rng.integers(0, 256, (600, 32), dtype=np.uint8), np.full((3, 32), 0xFF, np.uint8), np.zeros((2, 32), np.uint8)])
    idx, dist = E.knn_match_hamming(q, t, gpu_ctx)
    ridx, rdist = oracle_lib.knn2_hamming(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    for ratio in ratios:
        a = E.match_hamming(q, t, ratio, gpu_ctx); b = oracle_lib.match_hamming(q, t, ratio)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), ratio


def test_l2_one_product_pass_at_its_largest_train_set(gpu_ctx, oracle_lib):
    """65 536 train rows: the last size the one-product pass's 12-bit position code numbers (11 bits of 32-row steps, one bit for the
    step's two fold groups); neighbours planted in the last steps, duplicates across the set's ends."""
    rng = np.random.default_rng(44)
    nt = 65536
    t = rng.standard_normal((nt, 64)).astype(np.float32); t /= np.linalg.norm(t, axis=1, keepdims=True)
    q = rng.standard_normal((256, 64)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    t[65535] = q[0]; t[65500] = q[0]; t[3] = q[0]                       # three exact copies: rows 3 and 65500 win
    t[65520] = q[1] + 0.01 * rng.standard_normal(64).astype(np.float32)
    t[40000] = q[2] + 0.02 * rng.standard_normal(64).astype(np.float32)
    _check_knn_l2(gpu_ctx, oracle_lib, q, t)
    for ratio in (0.6, 0.9):
        a = E.match_l2(q, t, ratio, gpu_ctx); b = oracle_lib.match_l2(q, t, ratio)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(_bits(a[2]), _bits(b[2]))


def test_hamming_fp4_large_train_set(gpu_ctx, oracle_lib):
    """A train set beyond the L2 pass's 32 768 rows: the FP4 Hamming form numbers 262 144 (its scores leave 14 zero mantissa bits
    for the position code); winners and ties placed in the last steps of the set."""
    rng = np.random.default_rng(33)
    nt = 70001
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t[[69990, 70000, 40000, 32768, 32767]] = q[0]              # five exact copies of query 0: the two lowest indices win
    t[65536] = q[1]; t[65537] = q[1]; t[65537, 0] ^= 1
    idx, dist = E.knn_match_hamming(q, t, gpu_ctx)
    ridx, rdist = oracle_lib.knn2_hamming(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    a = E.match_hamming(q, t, 0.8, gpu_ctx); b = oracle_lib.match_hamming(q, t, 0.8)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_hamming_match_orb_like(gpu_ctx, oracle_lib):
    o = synth.orb_like_sets(2, 2000, pool=4096, seed_base=300)
    a = E.match_hamming(o[1], o[0], 0.8, gpu_ctx); b = oracle_lib.match_hamming(o[1], o[0], 0.8)
    assert len(b[0]) > 0 and all(np.array_equal(x, y) for x, y in zip(a, b))


def test_feature_matching_mirror_appends(gpu_ctx, oracle_lib):
    """matchFeaturesSURF/ORB keep the reference's signature and append to `matches` (:90,:135)."""
    s = synth.surf_like_sets(2, 400, pool=512, seed_base=40)
    f1 = E.Frame(frame_id=1, descriptors=s[1]); f2 = E.Frame(frame_id=0, descriptors=s[0])
    fm = E.FeatureMatching(gpu_ctx)
    matches = [E.DMatch(-1, -1, 0.0)]
    assert fm.matchFeaturesSURF(f1, f2, matches) is True
    rq, rt, rd = oracle_lib.match_l2(s[1], s[0], 0.5)
    assert matches[0].queryIdx == -1 and len(matches) == 1 + len(rq)
    assert [m.queryIdx for m in matches[1:]] == rq.tolist() and [m.trainIdx for m in matches[1:]] == rt.tolist()
    o = synth.orb_like_sets(2, 400, pool=512, seed_base=41)
    f1.descriptors, f2.descriptors = o[1], o[0]
    m2 = []
    fm.matchFeaturesORB(f1, f2, m2)
    rq, rt, rd = oracle_lib.match_hamming(o[1], o[0], 0.8)
    assert [(m.queryIdx, m.trainIdx, m.distance) for m in m2] == list(zip(rq.tolist(), rt.tolist(), rd.tolist()))


def test_batched_pairs_match_single_calls(gpu_ctx, oracle_lib):
    """esfm_match_pairs_dev over a ragged set list == per-pair oracle results, in pair order."""
    rng = np.random.default_rng(3)
    sizes = [300, 0, 513, 128, 77]
    sets = []
    for i, n in enumerate(sizes):
        x = rng.standard_normal((n, 64)).astype(np.float32)
        x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
        sets.append(x)
    sets[2][:100] = sets[0][:100] + 0.02 * rng.standard_normal((100, 64)).astype(np.float32)
    pairs = synth.all_pairs(len(sizes))
    bank = E.DescriptorBank(sets, E.ESFM_L2_F32)
    pm = E.PairMatcher(bank, pairs)
    res = pm.match(0.8).to_host()
    for (i, j), (qi, ti, d) in zip(pairs, res):
        rq, rt, rd = oracle_lib.match_l2(sets[i], sets[j], 0.8) if len(sets[i]) else (np.zeros(0, np.int32),) * 3
        assert np.array_equal(qi, rq) and np.array_equal(ti, rt) and np.array_equal(_bits(d), _bits(rd))
    n_q, n_rescan = pm.stats()
    assert n_q == sum(sizes[i] for i, _ in pairs)
    assert 0 <= n_rescan <= n_q


@pytest.mark.parametrize("heavy_at", ["first", "last", "block_edges", "neighbours_only"])
def test_train_set_maxima_at_unaligned_set_boundaries(gpu_ctx, oracle_lib, heavy_at):
    """The one-product pass takes max |t|^2 / max rho_t of a train set from a table with one entry per 256 rows of the BANK plus the
    rows in front of and behind the whole blocks (l2_blockmax_kernel).  Sets that start and end off the 256-row grid, with rows of
    50 x the usual norm exactly at the set's first / last row, at the rows next to the table's block boundaries, or only in the
    NEIGHBOURING sets (which must not leak into the bound's correctness either way): a maximum that misses a row makes the
    certificate's bound too small, and the 2-NN / match lists would differ from the oracle's."""
    rng = np.random.default_rng(11)
    sizes = [300, 1000, 77, 700, 40]                       # set 1 = rows 300 .. 1299, set 3 = rows 1377 .. 2076 of the bank
    sets = []
    for n in sizes:
        x = rng.standard_normal((n, 64)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        sets.append(x)
    for s_ in (0, 2, 3, 4):                                # correspondences into the two big train sets
        k = min(len(sets[s_]), 40)
        sets[s_][:k] = sets[1][:k] + 0.03 * rng.standard_normal((k, 64)).astype(np.float32)
    heavy = {"first": {1: [0], 3: [0]}, "last": {1: [999], 3: [699]},
             "block_edges": {1: [211, 212, 467, 468, 979, 980], 3: [158, 159, 414, 415, 670, 671]},     # bank rows 511 / 512, 767 / 768, 1279 / 1280; 1535 / 1536 ...
             "neighbours_only": {0: [299], 2: [0, 76], 4: [0]}}[heavy_at]
    for s_, rows in heavy.items():
        for r in rows:
            sets[s_][r] *= 50.0
    pairs = [(0, 1), (2, 1), (3, 1), (4, 1), (0, 3), (1, 3), (2, 3), (4, 3)]
    bank = E.DescriptorBank(sets, E.ESFM_L2_F32)
    pm = E.PairMatcher(bank, pairs)
    for ratio in (0.6, 1.0):
        res = pm.match(ratio).to_host()
        for (i, j), (qi, ti, d) in zip(pairs, res):
            rq, rt, rd = oracle_lib.match_l2(sets[i], sets[j], ratio)
            assert np.array_equal(qi, rq) and np.array_equal(ti, rt) and np.array_equal(_bits(d), _bits(rd)), (heavy_at, i, j, ratio)
    idx, dist = pm.knn2()
    pm.ctx.synchronize()
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    for k, (i, j) in enumerate(pairs):
        o = int(pm.offset[k])
        ridx, rdist = oracle_lib.knn2_l2(sets[i], sets[j])
        assert np.array_equal(idx[o:o + len(sets[i])], ridx) and np.array_equal(_bits(dist[o:o + len(sets[i])]), _bits(rdist)), (heavy_at, i, j)


def test_persistent_workgroup_form_of_the_one_product_pass(gpu_ctx):
    """ESFM_X1_GRID (a measurement knob, read once per process): the one-product pass as persistent workgroups that take several
    512-query blocks each and prefetch the next block's operands -- not the default (measured slower), but the loop is in the
    kernel; ESFM_FIN_SLICES = 1: a pair's whole re-rank in one workgroup (several virtual sets per wave).  A child process with
    the knobs set matches a ragged pair list and compares every list with the oracle."""
    import os, subprocess, sys
    code = r'''
import sys; sys.path.insert(0, ".")
import numpy as np, easysfm_amd as E, oracle
from easysfm_amd import synth
rng = np.random.default_rng(5)
sizes = [1500, 700, 2300, 513, 40]
sets = []
for n in sizes:
    x = rng.standard_normal((n, 64)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True); sets.append(x)
for s_ in range(1, 5):
    k = min(len(sets[s_]), 300); sets[s_][:k] = sets[0][:k] + 0.03 * rng.standard_normal((k, 64)).astype(np.float32)
pairs = synth.all_pairs(len(sizes))
pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), pairs)
for ratio in (0.6, 1.0):
    res = pm.match(ratio).to_host()
    for (i, j), (qi, ti, d) in zip(pairs, res):
        rq, rt, rd = oracle.match_l2(sets[i], sets[j], ratio)
        assert np.array_equal(qi, rq) and np.array_equal(ti, rt) and np.array_equal(d.view(np.uint32), rd.view(np.uint32)), (i, j, ratio)
print("ok")
'''
    env = dict(os.environ, ESFM_X1_GRID="16", ESFM_FIN_SLICES="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-2000:]


def test_match_pairs_host_pointer_batch_equals_single_calls(gpu_ctx, oracle_lib):
    """esfm_match_pairs (round 4: the batched pair loop through HOST pointers, what the C++ driver calls once for sfm.cpp:140-161):
    ragged sets incl. an empty one, L2 and Hamming, every pair's list equal to the single-pair entry point's and the oracle's."""
    rng = np.random.default_rng(21)
    sizes = [300, 0, 517, 64, 1000]
    sets = [rng.standard_normal((n, 64)).astype(np.float32) for n in sizes]
    sets[2][:100] = sets[0][:100] + 1e-3 * rng.standard_normal((100, 64)).astype(np.float32)       # planted matches
    sets[4][:64] = sets[3] * np.float32(1.0001)
    pairs = synth.all_pairs(len(sets))
    res = E.match_pairs_host(sets, pairs, 0.7, E.ESFM_L2_F32, gpu_ctx)
    n_tot = 0
    for (i, j), (q, t, d) in zip(pairs, res):
        rq, rt, rd = oracle_lib.match_l2(sets[i], sets[j], 0.7)
        assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(_bits(d), _bits(rd)), (i, j)
        if len(sets[i]) and len(sets[j]):
            sq, st_, sd = E.match_l2(sets[i], sets[j], 0.7, gpu_ctx)
            assert np.array_equal(q, sq) and np.array_equal(t, st_) and np.array_equal(_bits(d), _bits(sd))
        n_tot += len(q)
    assert n_tot >= 150
    osets = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for n in sizes]
    osets[2][:80] = osets[0][:80]; osets[2][:80, 0] ^= 1
    res = E.match_pairs_host(osets, pairs, 0.8, E.ESFM_HAMMING, gpu_ctx)
    for (i, j), (q, t, d) in zip(pairs, res):
        rq, rt, rd = oracle_lib.match_hamming(osets[i], osets[j], 0.8)
        assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(d, rd), (i, j)
    # a second batch on the same context (other sizes: the bank and every scratch buffer are re-used or re-grown)
    sets2 = [rng.standard_normal((n, 64)).astype(np.float32) for n in (40, 2100)]
    (q, t, d), = E.match_pairs_host(sets2, np.array([[1, 0]], np.int32), 1.0, E.ESFM_L2_F32, gpu_ctx)
    rq, rt, rd = oracle_lib.match_l2(sets2[1], sets2[0], 1.0)
    assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(_bits(d), _bits(rd))


def test_alternating_pair_lists_on_one_context(gpu_ctx, oracle_lib):
    """Round 4: the matcher's per-pair counters live in two phases (a call fills one, its first kernel zeroes the other for the next
    call) and the bf16 operand images are kept per prepared buffer.  Calls that alternate between banks and pair lists of different
    lengths on ONE context -- 45 pairs, 3, 190, match and knn2 mixed, an audit call in between -- must each give the oracle's result."""
    rng = np.random.default_rng(31)

    def bank(n_sets, n_rows, seed):
        r = np.random.default_rng(seed)
        base = r.standard_normal((256, 64)).astype(np.float32)
        out = []
        for k in range(n_sets):
            x = r.standard_normal((n_rows + 7 * k, 64)).astype(np.float32)
            x[:100] = base[r.integers(0, 256, 100)] + np.float32(0.02) * r.standard_normal((100, 64)).astype(np.float32)
            out.append(np.ascontiguousarray(x / np.linalg.norm(x, axis=1, keepdims=True)))
        return out
    cases = [(bank(10, 300, 1), None), (bank(3, 900, 2), None), (bank(20, 120, 3), None)]
    pms = []
    for sets, _ in cases:
        pairs = synth.all_pairs(len(sets))
        pms.append((sets, pairs, E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), pairs, gpu_ctx)))
    order = [0, 1, 2, 1, 0, 2, 2, 0]
    for step, k in enumerate(order):
        sets, pairs, pm = pms[k]
        pm.prepare()                                   # (another bank was prepared in between)
        if step % 3 == 2:
            idx, dist = pm.knn2(); pm.ctx.synchronize()
            idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
            off = pm.offset
            for p, (i, j) in enumerate(pairs):
                ridx, rdist = oracle_lib.knn2_l2(sets[i], sets[j])
                sl = slice(int(off[p]), int(off[p + 1]))
                assert np.array_equal(idx[sl], ridx) and np.array_equal(_bits(dist[sl]), _bits(rdist)), (step, k, i, j)
        else:
            ratio = (0.5, 0.8, 0.65)[step % 3]
            res = pm.match(ratio).to_host()
            for (i, j), (q, t, d) in zip(pairs, res):
                rq, rt, rd = oracle_lib.match_l2(sets[i], sets[j], ratio)
                assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(_bits(d), _bits(rd)), (step, k, i, j)
        if step == 3:                                  # an audit call leaves its lists behind: the next call must not see them
            pm.set_l2_audit(4); pm.match(0.5); pm.ctx.synchronize(); assert len(pm.flagged()) > 0; pm.set_l2_audit(0)


def test_many_pairs_heavy_rescan(gpu_ctx, oracle_lib):
    """The re-scan of uncertified queries works pair by pair, in chunks of the pair's list (l2_rescan64_pairs_kernel); launches
    of 2048 pairs and more use the large chunks and ONE workgroup per pair that loops over them.  70 small sets full of
    duplicated rows: 2415 pairs, nearly every query uncertified (more duplicates than the pass keeps candidates), tens of
    chunks per pair -- every pair against the oracle, bit for bit."""
    rng = np.random.default_rng(11)
    base = rng.standard_normal((12, 64)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    sets = []
    for i in range(70):
        n = 80 + (i % 5) * 9
        x = base[rng.integers(0, 12, n)].copy()
        fresh = rng.random(n) < 0.15
        x[fresh] = rng.standard_normal((int(fresh.sum()), 64)).astype(np.float32)
        x[fresh] /= np.linalg.norm(x[fresh], axis=1, keepdims=True)
        sets.append(np.ascontiguousarray(x))
    pairs = synth.all_pairs(70)
    assert len(pairs) >= 2048
    bank = E.DescriptorBank(sets, E.ESFM_L2_F32)
    pm = E.PairMatcher(bank, pairs)
    idx, dist = pm.knn2()
    pm.ctx.synchronize()               # the library's own stream: torch's copies below do not wait for it
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    n_q, n_rescan = pm.stats()
    assert pm.second_pass() > n_q // 3     # the case is about the passes behind the first one (round 3: the threshold-filter pass
                                           # resolves these duplicates exactly; its overflow path to the re-scan: next test)
    off = pm.offset
    for k, (i, j) in enumerate(pairs):
        ridx, rdist = oracle_lib.knn2_l2(sets[i], sets[j])
        sl = slice(int(off[k]), int(off[k + 1]))
        assert np.array_equal(idx[sl], ridx) and np.array_equal(_bits(dist[sl]), _bits(rdist)), (i, j)


def test_refine_overflow_takes_the_rescan(gpu_ctx, oracle_lib):
    """Train sets made of thousands of copies of a few rows: every query ties with more rows than the threshold-filter pass keeps
    hits for (1024 per chunk of 32 queries), so its chunks overflow and the exact re-scan (l2_rescan64_pairs_kernel) decides --
    lowest train index first, like the oracle."""
    rng = np.random.default_rng(12)
    base = rng.standard_normal((3, 64)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    sets = []
    for i in range(5):
        n = 2500 + 37 * i
        x = base[rng.integers(0, 3, n)].copy()
        fresh = rng.random(n) < 0.02
        x[fresh] = rng.standard_normal((int(fresh.sum()), 64)).astype(np.float32)
        sets.append(np.ascontiguousarray(x))
    pairs = synth.all_pairs(5)
    bank = E.DescriptorBank(sets, E.ESFM_L2_F32)
    pm = E.PairMatcher(bank, pairs)
    idx, dist = pm.knn2()
    pm.ctx.synchronize()
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    n_q, n_rescan = pm.stats()
    assert n_rescan > n_q // 2
    off = pm.offset
    for k, (i, j) in enumerate(pairs):
        ridx, rdist = oracle_lib.knn2_l2(sets[i], sets[j])
        sl = slice(int(off[k]), int(off[k + 1]))
        assert np.array_equal(idx[sl], ridx) and np.array_equal(_bits(dist[sl]), _bits(rdist)), (i, j)


def test_l2_overflowing_and_nan_rows(gpu_ctx, oracle_lib):
    """Rows whose distance overflows f32 (+inf) or is NaN are never neighbours (the oracle's strict `d < d1` against FLT_MAX):
    a few such train rows, a NaN query row, and a train set made of nothing else (no neighbours at all: indices -1)."""
    rng = np.random.default_rng(21)
    q = rng.standard_normal((300, 64)).astype(np.float32); t = rng.standard_normal((700, 64)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True); t /= np.linalg.norm(t, axis=1, keepdims=True)
    t[[3, 77, 500]] *= np.float32(1e25)
    t[123, 5] = np.nan
    q[17, 40] = np.nan
    _check_knn_l2(gpu_ctx, oracle_lib, q, t)
    with np.errstate(over="ignore", invalid="ignore"):
        huge = (t[:50] * np.float32(1e25)).astype(np.float32)         # some entries overflow to inf themselves
    _check_knn_l2(gpu_ctx, oracle_lib, q, huge)
    # the same on 128-float descriptors (f32-input MFMA pass: a NaN key would corrupt its med3 network)
    q2 = rng.standard_normal((200, 128)).astype(np.float32); t2 = rng.standard_normal((500, 128)).astype(np.float32)
    t2[[1, 300]] *= np.float32(1e25); t2[42, 100] = np.nan; q2[9, 3] = np.nan
    _check_knn_l2(gpu_ctx, oracle_lib, q2, t2)


def test_full_size_properties(gpu_ctx, oracle_lib):
    """BASELINE size (4096 x 4096 x 64): size-independent checks -- a query that IS a train row
    finds it at distance 0; a sample of rows agrees with the oracle bit for bit; permuting the train
    set permutes the indices."""
    s = synth.surf_like_sets(2, 4096, pool=16384, seed_base=1000)
    q, t = s[1].copy(), s[0]
    q[:64] = t[100:164]
    idx, dist = E.knn_match_l2(q, t, gpu_ctx)
    assert np.array_equal(idx[:64, 0], np.arange(100, 164)) and np.all(dist[:64, 0] == 0.0)
    rows = np.random.default_rng(0).choice(4096, 256, replace=False)
    ridx, rdist = oracle_lib.knn2_l2(q[rows], t)
    assert np.array_equal(idx[rows], ridx) and np.array_equal(_bits(dist[rows]), _bits(rdist))
    perm = np.random.default_rng(1).permutation(4096)
    idx2, dist2 = E.knn_match_l2(q, t[perm], gpu_ctx)
    assert np.array_equal(_bits(dist2), _bits(dist))
    same = dist[:, 0] != dist[:, 1]
    assert np.array_equal(perm[idx2[same, 0]], idx[same, 0])


def test_config4_one_rank_shard_at_full_size(gpu_ctx, oracle_lib):
    """BASELINE config 4 (8192 SURF features per image, all pairs sharded over 8 GPUs, no collective): rank 3's share of the
    pair list on one GPU, pairs at full size (8192 x 8192 x 64).  96 images instead of 256 keep the test short -- the image
    count only multiplies the number of pairs (570 here, 4080 at 256 images).  Size-independent checks: the shard is a cost-balanced eighth of the (i, j < i) list;
    per-pair output slices tile the output exactly; sampled pairs agree with the oracle bit for bit on sampled query rows;
    every emitted match passes the ratio test against its own 2-NN record."""
    n_img, n_feat, world, rank = 96, 8192, 8, 3
    sets = synth.surf_like_sets(n_img, n_feat, pool=65536, seed_base=2000)
    rows = np.full(n_img, n_feat, np.int32)
    pairs = E.shard_pair_list(n_img, rows, rank, world)
    assert abs(len(pairs) - n_img * (n_img - 1) // 2 / world) <= 1
    bank = E.DescriptorBank(sets, E.ESFM_L2_F32)
    pm = E.PairMatcher(bank, pairs)
    res = pm.match(0.5)
    host = res.to_host()
    assert len(host) == len(pairs)
    off = pm.offset
    assert off[0] == 0 and np.all(np.diff(off) == n_feat) and off[-1] == len(pairs) * n_feat
    n_q, n_rescan = pm.stats()
    assert n_q == len(pairs) * n_feat and n_rescan < n_q // 100
    rng = np.random.default_rng(5)
    total = 0
    for k in rng.choice(len(pairs), 4, replace=False):
        i, j = pairs[k]
        q, t, d = host[k]
        total += len(q)
        assert np.all(np.diff(q) > 0) and (len(q) == 0 or (q[-1] < n_feat and t.max() < n_feat))
        sel = np.sort(rng.choice(n_feat, 96, replace=False))
        rq, rt, rd = oracle_lib.match_l2(sets[i][sel], sets[j], 0.5)
        m = np.isin(q, sel)
        assert np.array_equal(np.searchsorted(sel, q[m]), rq) and np.array_equal(t[m], rt) and np.array_equal(_bits(d[m]), _bits(rd))
    assert total > 0

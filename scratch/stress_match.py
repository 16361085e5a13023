# randomized parity stress of the matchers (L2 bf16 pass + DMA tail + pair re-scan, Hamming two-level): ragged sizes, duplicates
import sys, time; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E, oracle
from easysfm_amd import synth
oracle.build(); 
ctx = E.Context(0, None)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0; t0 = time.time()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    big = (it % 10 == 9)
    nq = int(rng.integers(1, 1500 if big else 700)); nt = int(rng.integers(1, 6000 if big else 1500))
    mode = it % 4
    q = rng.standard_normal((nq, 64)).astype(np.float32); t = rng.standard_normal((nt, 64)).astype(np.float32)
    if mode == 1:   # duplicates
        t = t[rng.integers(0, max(nt // 8, 1), nt)]; q[: nq // 2] = t[rng.integers(0, nt, nq // 2)]
    if mode == 2:   # near ties
        t = t[rng.integers(0, max(nt // 4, 1), nt)] * (1 + 1e-6 * rng.standard_normal((nt, 1)).astype(np.float32))
    if mode != 3:
        q /= np.linalg.norm(q, axis=1, keepdims=True); t /= np.linalg.norm(t, axis=1, keepdims=True)
    idx, dist = E.knn_match_l2(q, t, ctx)
    ridx, rdist = oracle.knn2_l2(q, t)
    ok = np.array_equal(idx, ridx) and np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    qb = rng.integers(0, 256, (nq, 32), dtype=np.uint8); tb = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    if mode == 1: tb = tb[rng.integers(0, max(nt // 8, 1), nt)]; qb[: nq // 2] = tb[rng.integers(0, nt, nq // 2)]
    hi, hd = E.knn_match_hamming(qb, tb, ctx)
    rhi, rhd = oracle.knn2_hamming(qb, tb)
    okh = np.array_equal(hi, rhi) and np.array_equal(hd, rhd)
    if not (ok and okh):
        bad += 1; print('MISMATCH', it, nq, nt, mode, ok, okh, flush=True)
print('stress done', it + 1, 'cases, mismatches', bad, 'in', round(time.time() - t0, 1), 's', flush=True)

# batched pair lists over ragged sets (pair binning, chunked re-scan, ratio compaction)
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
    nsets = int(rng.integers(2, 30))
    sizes = [int(rng.integers(0, 400)) for _ in range(nsets)]
    pool = rng.standard_normal((64, 64)).astype(np.float32)
    sets = []
    for n in sizes:
        x = rng.standard_normal((n, 64)).astype(np.float32)
        if n: 
            m = rng.random(n) < 0.3; x[m] = pool[rng.integers(0, 64, int(m.sum()))]
            x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
        sets.append(np.ascontiguousarray(x))
    pairs = synth.all_pairs(nsets)
    pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), pairs)
    res = pm.match(0.8).to_host()
    nb = 0
    for (i, j), (qi, ti, d) in zip(pairs, res):
        if len(sets[i]) == 0: continue
        rq, rt, rd = oracle.match_l2(sets[i], sets[j], 0.8)
        if not (np.array_equal(qi, rq) and np.array_equal(ti, rt) and np.array_equal(d.view(np.uint32), rd.view(np.uint32))): nb += 1
    print('batched', rep, nsets, 'sets', len(pairs), 'pairs, mismatching pairs', nb, 'rescans', pm.stats(), flush=True)

# stress: many random shapes through the split-bf16 L2 path against the oracle, bit for bit (races show up as rare mismatches)
import sys; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E, oracle
ctx = E.Context(0, None)
rng = np.random.default_rng(2024)
bad = 0
for it in range(60):
    nq = int(rng.integers(1, 1500)); nt = int(rng.integers(2, 3000))
    q = rng.standard_normal((nq, 64)).astype(np.float32); t = rng.standard_normal((nt, 64)).astype(np.float32)
    if it % 3 == 0:
        q /= np.linalg.norm(q, axis=1, keepdims=True); t /= np.linalg.norm(t, axis=1, keepdims=True)
    if it % 5 == 0:
        t[rng.integers(0, nt, nt // 3)] = t[0]
    idx, dist = E.knn_match_l2(q, t, ctx)
    ridx, rdist = oracle.knn2_l2(q, t)
    ok = np.array_equal(idx, ridx) and np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    bad += not ok
    if not ok: print('MISMATCH', it, nq, nt, int((idx != ridx).sum()))
# repeated identical launches must give identical results (timing-dependent bugs)
from easysfm_amd import synth
sets = synth.surf_like_sets(12, 4096, pool=16384, seed_base=1000)
bank = E.DescriptorBank(sets, E.ESFM_L2_F32); pm = E.PairMatcher(bank, synth.all_pairs(12))
ref = [tuple(np.copy(a) for a in r) for r in pm.match(0.5).to_host()]
for rep in range(30):
    cur = pm.match(0.5).to_host()
    for a, b in zip(ref, cur):
        if not all(np.array_equal(x, y) for x, y in zip(a, b)): bad += 1; print('NONDETERMINISTIC at repeat', rep); break
print('stress done, failures:', bad)

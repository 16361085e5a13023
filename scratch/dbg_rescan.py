import sys; sys.path.insert(0,'.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth
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
def ref(q, t):
    d2 = ((q[:, None, :].astype(np.float64) - t[None, :, :]) ** 2).sum(2)
    o = np.lexsort((np.arange(t.shape[0])[None, :].repeat(q.shape[0], 0), d2), axis=1)[:, :2]
    return o
print('start', flush=True)
ctx0 = E.Context(0, None)
i1, d1 = E.knn_match_l2(sets[1], sets[0], ctx0)
r = ref(sets[1], sets[0])
print('single pair mismatching rows:', np.where((i1 != r).any(1))[0][:10], flush=True)
for npairs in [int(a) for a in sys.argv[1:]]:
    pairs = synth.all_pairs(70)[:npairs]
    bank = E.DescriptorBank(sets, E.ESFM_L2_F32)
    pm = E.PairMatcher(bank, pairs)
    for mode in (0, 1, 2):
        pm.set_l2_audit(mode)
        idx, dist = pm.knn2(); pm.ctx.synchronize(); idx = idx.cpu().numpy().copy()
        print('ran', npairs, mode, flush=True)
        off = pm.offset
        fl = pm.flagged() if mode == 0 else None
        bad = 0
        for k, (i, j) in enumerate(pairs[:40]):
            r = ref(sets[i], sets[j])
            sl = slice(int(off[k]), int(off[k + 1]))
            w = np.where((idx[sl] != r).any(1))[0]
            if len(w):
                bad += 1
                if bad <= 2:
                    isfl = [bool(((fl[:, 0] == k) & (fl[:, 1] == q)).any()) for q in w[:6]] if fl is not None else None
                    print('  pair', k, (int(i), int(j)), 'rows', w[:6], 'gpu', idx[sl][w[:6]].tolist(), 'ref', r[w[:6]].tolist(), 'flagged', isfl, flush=True)
        print('npairs', npairs, 'audit', mode, 'bad pairs of first 40:', bad, 'stats', pm.stats(), flush=True)
    pm.set_l2_audit(0)

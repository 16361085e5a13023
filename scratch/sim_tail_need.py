"""How many kept groups per query does the one-product pass's tail have to re-rank exactly?  (M-SURF-4k, K = 4, groups of 4 and 8.)
a-priori rule: group needed iff qn + key - E <= U,  U = qn + kb + E (kb = second smallest key);
sequential rule: the two best groups always; a further group iff qn + key - E <= (exact second-best d^2 so far)."""
import numpy as np, sys
sys.path.insert(0, '.')
from easysfm_amd import synth

def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)

def sim(q, t, keep=4, grp=4):
    q64, t64 = q.astype(np.float64), t.astype(np.float64)
    qn, tn = (q64 ** 2).sum(1), (t64 ** 2).sum(1)
    D = qn[:, None] + tn[None, :] - 2 * q64 @ t64.T
    qh, th = bf16(q).astype(np.float64), bf16(t).astype(np.float64)
    rq, rt = np.sqrt(((q64 - qh) ** 2).sum(1)), np.sqrt(((t64 - th) ** 2).sum(1))
    S = tn[None, :] - 2 * qh @ th.T
    tmax = tn.max(); nq, nt = S.shape
    K = (S.astype(np.float32).view(np.uint32) & 0xFFFFE000).view(np.float32).astype(np.float64)
    ng = nt // grp
    G = K.reshape(nq, ng, grp).min(2)
    Dg = D.reshape(nq, ng, grp)
    half = (np.arange(ng) & 1)
    E = 2 * (rq * np.sqrt(tmax) + np.sqrt((qh ** 2).sum(1)) * rt.max()) * 1.002 + (qn + tmax) * 2.0 ** -15
    keys, gidx = [], []
    for hsel in (0, 1):
        Gh = np.where(half[None, :] == hsel, G, np.inf)
        idx = np.argsort(Gh, axis=1)[:, :keep]
        keys.append(np.take_along_axis(Gh, idx, 1)); gidx.append(idx)
    keys = np.concatenate(keys, 1); gidx = np.concatenate(gidx, 1)
    order = np.argsort(keys, axis=1)
    keys = np.take_along_axis(keys, order, 1); gidx = np.take_along_axis(gidx, order, 1)
    trunc = np.abs(keys) * 2.0 ** -10
    kb = keys[:, 1]
    U = qn + kb + E + np.abs(kb) * 2.0 ** -10
    need_apriori = ((qn[:, None] + keys - E[:, None] - trunc) <= U[:, None]).sum(1)
    # sequential: evaluate groups in key order; stop when the next group's lower bound exceeds the exact second best so far
    need_seq = np.zeros(nq, int)
    for i in range(nq):
        best = []
        for r in range(keys.shape[1]):
            if r >= 2:
                b = sorted(best)[1]
                if qn[i] + keys[i, r] - E[i] - trunc[i, r] > b:
                    break
            best.extend(Dg[i, gidx[i, r]].tolist()); need_seq[i] += 1
    return need_apriori, need_seq

sets = synth.surf_like_sets(3, 4096, pool=16384, seed_base=1000)
for grp in (4, 8):
    a, s_ = sim(sets[1], sets[0], 4, grp)
    print("groups of", grp, ": a-priori mean", a.mean(), "hist", np.bincount(a, minlength=9)[:9], " sequential mean", s_.mean(), "hist", np.bincount(s_, minlength=9)[:9])
    # wave-level rounds with the current lane-pair scheme: round r serves ranks 2r, 2r+1 -> rounds = ceil(need / 2), max over 32 queries
    for name, v in (("a-priori", a), ("sequential", s_)):
        r = np.ceil(v / 2).reshape(-1, 32).max(1)
        print("   ", name, "rounds per wave-set (max over 32 queries):", r.mean(), " dense rounds (sum / 64):", (v.reshape(-1, 32).sum(1) / 64).mean())

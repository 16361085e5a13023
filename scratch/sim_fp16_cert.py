"""Simulation: how many queries would fail the certificate with (a) the shipped split-bf16 3-product pass and
(b) a 2-product fp16 pass (query rounded to fp16, train split hi+lo in fp16)?  numpy, a few pairs of M-SURF-4k."""
import numpy as np, sys
sys.path.insert(0, '.')
from easysfm_amd import synth
sets = synth.surf_like_sets(6, 4096)
u = 2.0 ** -24
def sim(q, t):
    q64, t64 = q.astype(np.float64), t.astype(np.float64)
    qn, tn = (q64 ** 2).sum(1), (t64 ** 2).sum(1)
    D = qn[:, None] + tn[None, :] - 2 * q64 @ t64.T            # exact d^2 (f64)
    tmax = tn.max()
    # (a) bf16x3: score ~ exact (errors ~ 1e-6): use exact; (b) fp16 query
    qh = q.astype(np.float16).astype(np.float64)
    dq = np.sqrt(((q64 - qh) ** 2).sum(1))
    Sb = tn[None, :] - 2 * qh @ t64.T                          # what the fp16 pass computes (+ tiny noise)
    Sa = tn[None, :] - 2 * q64 @ t64.T
    res = {}
    for name, S, eps1 in (("bf16x3", Sa, (qn + tmax) * 2.0 ** -15),
                          ("fp16x2", Sb, 2 * dq * np.sqrt(tmax) * 1.001 + (24 + 96 + 16) * u * (qn + 2 * tmax)),
                          ("fp16x2_worst", Sb, 2 * 2.0 ** -11 * np.sqrt(qn) * np.sqrt(tmax) + (24 + 96 + 16) * u * (qn + 2 * tmax))):
        # truncate keys to 15 mantissa bits + code: emulate by zeroing low 8 bits of f32
        K = S.astype(np.float32).view(np.uint32) & 0xFFFFFF00
        K = K.view(np.float32).astype(np.float64)
        nq, nt = S.shape
        G = K.reshape(nq, nt // 4, 4).min(2)                    # group minima (groups of 4 consecutive rows)
        half = (np.arange(nt // 4) & 1)                         # group g belongs to lane half g & 1
        fails = 0
        tau_all = np.empty(nq)
        kept = np.zeros((nq, nt // 4), bool)
        for hsel in (0, 1):
            Gh = np.where(half[None, :] == hsel, G, np.inf)
            idx = np.argpartition(Gh, 3, axis=1)[:, :3]
            np.put_along_axis(kept, idx, True, axis=1)
            third = np.take_along_axis(Gh, idx, 1).max(1)
            tau_all = third if hsel == 0 else np.minimum(tau_all, third)
        keptrows = np.repeat(kept, 4, axis=1)
        Dc = np.where(keptrows, D, np.inf)
        D2 = np.partition(Dc, 1, axis=1)[:, 1]
        eps = eps1 + np.abs(tau_all) * 1.0001 * 2.0 ** -15
        cert = (qn + tau_all - eps) > D2 * (1 + 2.0 ** -21)
        res[name] = int((~cert).sum())
    return res
tot = {}
n = 0
for i in range(1, 4):
    for j in range(i):
        r = sim(sets[i], sets[j]); n += 4096
        for k, v in r.items(): tot[k] = tot.get(k, 0) + v
print(n, tot, {k: v / n for k, v in tot.items()})

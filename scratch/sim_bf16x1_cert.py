"""Simulation for the one-product bf16 distance pass (round 3): how many queries would its certificate leave uncertified?
numpy, a few pairs of M-SURF-4k / M-SURF-8k.  Models of eps: worst case (2^-8 relative per operand, Cauchy-Schwarz) and
per-row residual norms (rho = |x - bf16(x)|_2 measured per row).  Also: how many train rows per uncertified query pass the
threshold filter  s(t) < U - |q|^2 + eps  of a second sweep (U = exact second best among pass A's candidates)."""
import numpy as np, sys
sys.path.insert(0, '.')
from easysfm_amd import synth

def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)

def sim(q, t, keep=3):
    q64, t64 = q.astype(np.float64), t.astype(np.float64)
    qn, tn = (q64 ** 2).sum(1), (t64 ** 2).sum(1)
    D = qn[:, None] + tn[None, :] - 2 * q64 @ t64.T
    qh, th = bf16(q).astype(np.float64), bf16(t).astype(np.float64)
    rq, rt = np.sqrt(((q64 - qh) ** 2).sum(1)), np.sqrt(((t64 - th) ** 2).sum(1))
    S = tn[None, :] - 2 * qh @ th.T
    tmax = tn.max(); nq, nt = S.shape
    K = (S.astype(np.float32).view(np.uint32) & 0xFFFFFF00).view(np.float32).astype(np.float64)
    G = K.reshape(nq, nt // 4, 4).min(2)
    half = (np.arange(nt // 4) & 1)
    kept = np.zeros((nq, nt // 4), bool)
    for hsel in (0, 1):
        Gh = np.where(half[None, :] == hsel, G, np.inf)
        idx = np.argpartition(Gh, keep, axis=1)[:, :keep]
        np.put_along_axis(kept, idx, True, axis=1)
        third = np.take_along_axis(Gh, idx, 1).max(1)
        tau = third if hsel == 0 else np.minimum(tau, third)
    keptrows = np.repeat(kept, 4, axis=1)
    Dc = np.where(keptrows, D, np.inf)
    D2 = np.partition(Dc, 1, axis=1)[:, 1]
    acc = (24 + 96 + 16) * 2.0 ** -24 * (qn + 2 * tmax)
    hb = np.sqrt((qh ** 2).sum(1))
    models = {"worst": 2 * (2.0 ** -8 * np.sqrt(qn) * np.sqrt(tmax) + hb * 2.0 ** -8 * np.sqrt(tmax)) + acc,
              "rho": 2 * (rq * np.sqrt(tmax) + hb * rt.max()) * 1.001 + acc}
    out = {}
    for name, e1 in models.items():
        eps = e1 + np.abs(tau) * 1.0001 * 2.0 ** -15
        cert = (qn + tau - eps) > D2 * (1 + 2.0 ** -21)
        unc = ~cert
        # second sweep's filter: rows with s < U - qn + eps (U = D2): candidates per uncertified query (beyond the kept rows)
        thr = (D2 - qn + e1 + np.abs(D2 - qn) * 2.0 ** -15)[:, None]
        npass = ((S < thr) & ~keptrows)[unc].sum(1)
        # a truly-wrong check: is the true top-2 inside kept?
        true2 = np.partition(D, 1, axis=1)[:, 1]
        wrong = (true2 < D2)
        out[name] = (int(unc.sum()), float(npass.mean()) if unc.any() else 0.0, int(npass.max()) if unc.any() else 0, int((wrong & cert).sum()), int(wrong.sum()))
    return out, float(np.median(models["worst"])), float(np.median(models["rho"]))

for label, nfeat, pool, sb in () if "--fountain-only" in sys.argv else (("M-SURF-4k", 4096, 16384, 1000), ("M-SURF-8k", 8192, 65536, 2000)):
    sets = synth.surf_like_sets(4, nfeat, pool=pool, seed_base=sb)
    for keep in (3, 4, 6):
        tot, n = {}, 0
        for i in range(1, 4):
            for j in range(i):
                r, ew, er = sim(sets[i], sets[j], keep); n += nfeat
                for k, v in r.items():
                    a = tot.setdefault(k, [0, 0.0, 0, 0, 0]); a[0] += v[0]; a[1] += v[1] * v[0]; a[2] = max(a[2], v[2]); a[3] += v[3]; a[4] += v[4]
        print(label, "keep", keep, "eps worst/rho", round(ew, 5), round(er, 5),
              {k: dict(unc=v[0], frac=round(v[0] / n, 4), filt_mean=round(v[1] / max(v[0], 1), 2), filt_max=v[2], cert_wrong=v[3], pass_wrong=v[4]) for k, v in tot.items()})

# the reference's fountain images, SURF minHessian 300 (real descriptors: repeated texture, smaller gaps)
import os, oracle
imgs = np.load(os.path.join("tests", "golden", "fountain11_gray.npz"))["images"]
fsets = [oracle.surf(imgs[k], 300.0)[1] for k in range(5)]
for keep in (3, 4, 6, 8):
    tot, n = {}, 0
    for i in range(1, 5):
        for j in range(i):
            nt4 = (len(fsets[j]) // 8) * 8
            r, ew, er = sim(fsets[i], fsets[j][:nt4], keep); n += len(fsets[i])
            for k, v in r.items():
                a = tot.setdefault(k, [0, 0.0, 0, 0, 0]); a[0] += v[0]; a[1] += v[1] * v[0]; a[2] = max(a[2], v[2]); a[3] += v[3]; a[4] += v[4]
    print("fountain", "keep", keep, "eps worst/rho", round(ew, 5), round(er, 5),
          {k: dict(unc=v[0], frac=round(v[0] / n, 4), filt_mean=round(v[1] / max(v[0], 1), 2), filt_max=v[2], cert_wrong=v[3], pass_wrong=v[4]) for k, v in tot.items()})

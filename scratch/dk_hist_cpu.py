"""Sweeps of the oracle's (= the product's) simultaneous Durand-Kerner iteration over random 5-point samples (CPU only).
usage: python scratch/dk_hist_cpu.py [n_samples]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from easysfm_amd import synth
lib = oracle.load()
f = lib.esfm_ref_five_point_stages; f.restype = C.c_int; f.argtypes = [C.c_void_p] * 3
n_s = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(3)
sw = []
st = np.zeros(116)
while len(sw) < n_s:
    n = 200
    R = synth.aa_to_R(rng.normal(0, rng.choice([0.02, 0.15, 0.4]), 3)); t = rng.normal(0, 1, 3); t /= np.linalg.norm(t)
    X = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0, 8.0])
    if rng.random() < 0.15: X[:, 2] = 8.0
    Xc = X @ R.T + t
    a = X[:, :2] / X[:, 2:3] + rng.normal(0, 0.0005, (n, 2)); b = Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, 0.0005, (n, 2))
    bad = rng.choice(n, int(0.3 * n), replace=False); b[bad] += rng.uniform(-0.1, 0.1, (len(bad), 2))
    for _ in range(200):
        id5 = rng.choice(n, 5, replace=False)
        q1 = np.ascontiguousarray(a[id5]); q2 = np.ascontiguousarray(b[id5])
        sw.append(f(q1.ctypes.data, q2.ctypes.data, st.ctypes.data))
sw = np.array(sw)
print("samples", len(sw), "no polynomial", (sw < 0).sum(), "mean", sw[sw >= 0].mean(), "median", np.median(sw[sw >= 0]), "p90", np.percentile(sw[sw >= 0], 90),
      "p99", np.percentile(sw[sw >= 0], 99), "at cap", (sw >= 300).sum())

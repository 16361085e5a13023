"""Per-call wall time of ORB detect + describe on the 11 fountain images in a fresh process (what config 3's detect stage pays)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import easysfm_amd as E
z = np.load(os.path.join(ROOT, "tests", "golden", "fountain11_gray.npz"))["images"]
ctx = E.Context(0, None)
for rep in range(2):
    ts = []
    for im in z:
        bgr = np.ascontiguousarray(np.stack([im] * 3, axis=2))
        t0 = time.perf_counter()
        kp, d = E.orb_detect_and_compute(bgr, 8000, None, ctx)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("pass", rep, " ".join(f"{t:.2f}" for t in ts), "ms; keypoints of the last", len(kp))

"""Writes tests/golden/fountain11_gray.npz: the reference's 11 test images (test_data/images_25/00NN.png, 768 x 512, the
input of script/run_fountain_small.sh) as 8-bit gray, full resolution -- DATA for the config-2 parity test
(tests/test_metric_workloads_gpu.py::test_config2_fountain_fullres_all_pairs).  Decoding is PIL's; the gray conversion is
cv::cvtColor's 14-bit fixed-point BGR2GRAY as restated in oracle/surf_ref.c (what FeatureMatching::detectFeaturesSURF's
detector applies to the colour image internally).  Runs only in the build container (the PNGs live in /root/reference).

    python tests/golden/make_fountain_fixture.py
"""
import glob
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

files = sorted(glob.glob("/root/reference/test_data/images_25/*.png"))
assert len(files) == 11, files
imgs = []
for f in files:
    rgb = np.asarray(Image.open(f).convert("RGB"))
    imgs.append(oracle.bgr2gray(np.ascontiguousarray(rgb[:, :, ::-1])))
imgs = np.stack(imgs)
assert imgs.shape == (11, 512, 768) and imgs.dtype == np.uint8
out = os.path.join(ROOT, "tests", "golden", "fountain11_gray.npz")
np.savez_compressed(out, images=imgs, names=np.array([os.path.basename(f) for f in files]),
                    K4=np.array([689.87, 380.17, 691.04, 251.70]))   # fx cx fy cy of test_data/k_25/K.txt
print(out, os.path.getsize(out), "bytes")

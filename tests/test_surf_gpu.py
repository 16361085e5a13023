"""Parity of the HIP SURF detector + descriptor (reference FeatureMatching::detectFeaturesSURF,
cpp_code/src/feature_matching.cpp:43-58) through the C ABI against the CPU oracle: keypoints and descriptors bit for bit (both
sides keep OpenCV's float operation order; the Gaussian tables come from the host's exp() on both), on real texture (the
reference's first two test images at half resolution) and on synthetic images; then the properties a SURF implementation
must have (rotation invariance, repeatability under a shift) and the hand-off to the matcher."""
import os

import numpy as np
import pytest

import easysfm_amd as E

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _synthetic(rng, rows, cols, n_blobs=400):
    img = np.full((rows, cols), 110.0)
    yy, xx = np.mgrid[0:rows, 0:cols]
    for _ in range(n_blobs):
        cy, cx = rng.uniform(0, rows), rng.uniform(0, cols)
        sg = rng.uniform(1.5, 9.0); amp = rng.uniform(-70, 70)
        y0, y1 = int(max(cy - 4 * sg, 0)), int(min(cy + 4 * sg + 1, rows)); x0, x1 = int(max(cx - 4 * sg, 0)), int(min(cx + 4 * sg + 1, cols))
        img[y0:y1, x0:x1] += amp * np.exp(-((yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2) / (2 * sg * sg))
    img += rng.normal(0, 2.0, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _check_equal(gpu, ref):
    (kg, dg), (kr, dr) = gpu, ref
    assert len(kg) == len(kr) and len(kg) > 50
    assert np.array_equal(_bits(kg), _bits(kr))
    assert np.array_equal(_bits(dg), _bits(dr))


def test_surf_real_texture_bitexact(gpu_ctx, oracle_lib):
    z = np.load(os.path.join(GOLD, "fountain_pair_half.npz"))
    for name in ("img0", "img1"):
        bgr = z[name]
        gray = oracle_lib.bgr2gray(bgr)
        ref = oracle_lib.surf(gray, 300.0)
        _check_equal(E.surf_detect_and_compute(bgr, 300.0, None, gpu_ctx), ref)           # BGR in: the gray conversion runs on the GPU
        _check_equal(E.surf_detect_and_compute(gray, 300.0, None, gpu_ctx), ref)
    # strongest-first order and OpenCV's keypoint fields
    kp, d = E.surf_detect_and_compute(z["img0"], 300.0, None, gpu_ctx)
    assert np.all(np.diff(kp[:, 4]) <= 0) and np.all(kp[:, 4] > 300.0)
    assert np.all((kp[:, 3] >= 0) & (kp[:, 3] <= 360)) and set(np.unique(kp[:, 6])) <= {-1.0, 0.0, 1.0} and set(np.unique(kp[:, 5])) <= {0.0, 1.0, 2.0, 3.0}
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)
    # a capped call returns the strongest prefix
    kp2, d2 = E.surf_detect_and_compute(z["img0"], 300.0, 100, gpu_ctx)
    assert np.array_equal(kp2, kp[:100]) and np.array_equal(d2, d[:100])


@pytest.mark.parametrize("rows,cols,thr,seed", [(240, 320, 100.0, 0), (97, 131, 50.0, 1), (512, 768, 400.0, 2), (33, 40, 10.0, 3)])
def test_surf_synthetic_bitexact(gpu_ctx, oracle_lib, rows, cols, thr, seed):
    img = _synthetic(np.random.default_rng(seed), rows, cols, 60 if rows < 100 else 400)
    kg, dg = E.surf_detect_and_compute(img, thr, None, gpu_ctx)
    kr, dr = oracle_lib.surf(img, thr)
    assert len(kg) == len(kr)
    assert np.array_equal(_bits(kg), _bits(kr)) and np.array_equal(_bits(dg), _bits(dr))
    if rows >= 240:
        assert len(kg) > 100


def test_surf_invariances_and_matching(gpu_ctx):
    """rot90 of the image gives the same descriptors (the sampling grid maps onto itself exactly); a shift by 8 pixels (a
    multiple of every octave's sampling step) repeats the keypoints; the descriptors feed the matcher (detectFeaturesSURF -> matchFeaturesSURF, sfm.cpp:108,150)."""
    z = np.load(os.path.join(GOLD, "fountain_pair_half.npz"))
    g0 = z["img0"]
    kp, d = E.surf_detect_and_compute(g0, 300.0, None, gpu_ctx)
    kp90, d90 = E.surf_detect_and_compute(np.ascontiguousarray(np.rot90(g0)), 300.0, None, gpu_ctx)
    assert len(kp90) == len(kp)
    q, t, dist = E.match_l2(d, d90, 0.7, gpu_ctx)
    assert len(q) >= 0.95 * len(kp) and np.median(dist) < 0.05
    # (x, y) -> (y, W - 1 - x) under np.rot90
    W = g0.shape[1]
    assert np.allclose(kp90[t, 0], kp[q, 1], atol=0.05) and np.allclose(kp90[t, 1], W - 1 - kp[q, 0], atol=0.05)
    sh = np.ascontiguousarray(g0[:, 8:])
    kps, ds = E.surf_detect_and_compute(sh, 300.0, None, gpu_ctx)
    q, t, dist = E.match_l2(ds, d, 0.6, gpu_ctx)
    ok = np.abs(kp[t, 0] - 8 - kps[q, 0]) + np.abs(kp[t, 1] - kps[q, 1]) < 0.5
    assert len(q) > 0.6 * len(kps) and ok.mean() > 0.95
    # the two views of the fountain
    f0 = E.Frame(frame_id=0, rgb_image=z["img0"]); f1 = E.Frame(frame_id=1, rgb_image=z["img1"])
    fm = E.FeatureMatching(gpu_ctx)
    assert fm.detectFeaturesSURF(f0, 300) and fm.detectFeaturesSURF(f1, 300)
    m = []
    fm.matchFeaturesSURF(f1, f0, m)
    assert len(m) >= 15 and f0.keypoints.shape[1] == 2 and f0.descriptors.shape[1] == 64

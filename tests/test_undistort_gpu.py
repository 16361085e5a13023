"""Parity of the HIP image undistortion (undistort part; reference MotionEstimator::doUnDistort,
cpp_code/src/estimate_motion.cpp:431-441 = cv::undistort) through the C ABI: bit-exact against the committed golden images,
against the CPU oracle on larger seeded images, and size-independent properties at the reference's full image size."""
import os

import numpy as np
import pytest

import easysfm_amd as E

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["color_mild", "color_strong", "gray_barrel", "gray_zero", "wide_rows", "quirk_coeffs"]


@pytest.mark.parametrize("tag", CASES)
def test_undistort_golden_bitexact(gpu_ctx, tag):
    z = np.load(os.path.join(GOLD, "undistort_cases.npz"))
    out = E.undistort(z[tag + "_image"], z[tag + "_K4"], z[tag + "_dist"], gpu_ctx)
    assert np.array_equal(out, z[tag + "_out"])


@pytest.mark.parametrize("rows,cols,ch,seed", [(1, 1, 1, 0), (2, 3, 3, 1), (37, 255, 3, 2), (64, 257, 1, 3), (480, 640, 3, 4),
                                               (768, 1024, 3, 5), (5, 4097, 3, 6)])
def test_undistort_matches_oracle(gpu_ctx, oracle_lib, rows, cols, ch, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (rows, cols, ch) if ch == 3 else (rows, cols), dtype=np.uint8)
    f = 0.9 * max(rows, cols)
    K4 = np.array([f, cols / 2 + rng.uniform(-3, 3), f * 1.01, rows / 2 + rng.uniform(-3, 3)])
    for dist in ([-0.3, 0.1, 0.002, -0.001], [0.6, 0.4, -0.03, 0.05], [0, 0, 0, 0], [-5.0, 30.0, 0.3, 0.3]):
        out = E.undistort(img, K4, dist, gpu_ctx)
        assert np.array_equal(out, oracle_lib.undistort(img, K4, dist)), dist


def test_undistort_accepts_camera_matrix_and_mirror(gpu_ctx, oracle_lib):
    """3 x 3 float32 K as Frame.K_cam holds it, through MotionEstimator.doUnDistort; coefficients as importDistort leaves them."""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    K = np.array([[150.25, 0, 80.5], [0, 149.75, 60.25], [0, 0, 1]], np.float32)
    coeff = np.zeros(4, np.float64)
    coeff.view(np.float32)[:4] = np.array([-0.2, 1.5, 0.001, 0.002], np.float32)
    fr = E.Frame(frame_id=0, image_file_path="", rgb_image=img.copy())
    fr.K_cam = K
    assert E.MotionEstimator(gpu_ctx).doUnDistort(fr, coeff)
    ref = oracle_lib.undistort(img, [150.25, 80.5, 149.75, 60.25], coeff)
    assert np.array_equal(fr.rgb_image, ref)
    assert not np.array_equal(fr.rgb_image, img)          # k2 = 1.5 is large enough for the scrambled k1' to move pixels
    with pytest.raises(ValueError):
        E.undistort(img, np.array([[150, 1, 80], [0, 150, 60], [0, 0, 1.0]]), coeff, gpu_ctx)     # skew


@pytest.mark.parametrize("rows,cols", [(2048, 3072), (512, 768), (33, 64), (7, 4)])
def test_undistort_four_pixel_form_at_image_borders_matches_oracle(gpu_ctx, oracle_lib, rows, cols):
    """Three channels and a width that is a multiple of four take the four-pixels-per-thread kernel (8-byte tap loads inside the image,
    byte taps on its border, one 12-byte store): the reference's image sizes against the oracle, with a principal point far off
    centre and coefficients that send whole bands of the destination outside the source (every mix of in / on / outside taps within
    one thread's four pixels)."""
    rng = np.random.default_rng(rows * 7 + cols)
    img = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    f = 0.9 * max(rows, cols)
    for K4, dist in (([f, cols / 2, f, rows / 2], [-0.12, 0.03, 0.001, -0.0005]),
                     ([f, 0.2 * cols, 1.3 * f, 0.9 * rows], [0.6, 0.4, -0.03, 0.05]),
                     ([0.5 * f, cols - 1.0, 0.5 * f, 0.0], [-1.5, 2.0, 0.1, -0.1]),
                     ([f, -40.0, f, rows + 25.0], [0.05, 0.0, 0.0, 0.0])):
        out = E.undistort(img, K4, dist, gpu_ctx)
        assert np.array_equal(out, oracle_lib.undistort(img, K4, dist)), (K4, dist)


def test_undistort_registered_host_memory_takes_the_single_transfer_path(gpu_ctx):
    """common.hpp copy_h2d / copy_d2h: caller-owned pageable buffers travel in 512-KiB pieces, buffers the caller registered with HIP
    (here: torch's pinned allocator = hipHostMalloc) in one transfer -- same image either way, also when input and output differ in kind."""
    import ctypes as C
    import torch
    from easysfm_amd._lib import lib, check
    rng = np.random.default_rng(77)
    rows, cols = 1024, 1536                                      # 4.7 MB: nine pieces each way from pageable memory
    img = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    k4 = np.array([1400.0, 760.0, 1410.0, 500.0]); d4 = np.array([-0.2, 0.05, 0.001, -0.002])
    ref = E.undistort(img, k4, d4, gpu_ctx)
    pin_in = torch.empty((rows, cols, 3), dtype=torch.uint8).pin_memory(); pin_in.numpy()[:] = img
    pin_out = torch.empty((rows, cols, 3), dtype=torch.uint8).pin_memory()
    out_pageable = np.empty_like(img)
    for src, dst in ((pin_in.data_ptr(), pin_out.data_ptr()), (pin_in.data_ptr(), out_pageable.ctypes.data), (img.ctypes.data, pin_out.data_ptr())):
        pin_out.zero_(); out_pageable[:] = 0
        check(lib().esfm_undistort(gpu_ctx.handle, C.c_void_p(src), rows, cols, 3, C.c_void_p(k4.ctypes.data), C.c_void_p(d4.ctypes.data), C.c_void_p(dst)))
        got = pin_out.numpy() if dst == pin_out.data_ptr() else out_pageable
        assert np.array_equal(got, ref)
    # a cv::Mat-style buffer the host registered itself (hipHostRegister), as INTEGRATION.md suggests for large images
    hip = C.CDLL("libamdhip64.so")
    reg_in, reg_out = img.copy(), np.zeros_like(img)
    assert hip.hipHostRegister(C.c_void_p(reg_in.ctypes.data), C.c_size_t(reg_in.nbytes), C.c_uint(0)) == 0
    assert hip.hipHostRegister(C.c_void_p(reg_out.ctypes.data), C.c_size_t(reg_out.nbytes), C.c_uint(0)) == 0
    try:
        check(lib().esfm_undistort(gpu_ctx.handle, C.c_void_p(reg_in.ctypes.data), rows, cols, 3, C.c_void_p(k4.ctypes.data), C.c_void_p(d4.ctypes.data),
                                   C.c_void_p(reg_out.ctypes.data)))
        assert np.array_equal(reg_out, ref)
    finally:
        hip.hipHostUnregister(C.c_void_p(reg_in.ctypes.data)); hip.hipHostUnregister(C.c_void_p(reg_out.ctypes.data))


def test_undistort_full_size_properties(gpu_ctx):
    """The reference's image size (fountain: 2048 x 3072 BGR).  Zero coefficients are an exact identity (what every BASELINE
    configuration runs); a radial model leaves the principal point's neighbourhood in place, is symmetric under a 180-degree
    rotation of an image about a centred principal point, and a constant image stays constant wherever all four taps are inside."""
    rng = np.random.default_rng(11)
    rows, cols = 2048, 3072
    img = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    K4 = [2759.48, 1520.69, 2764.16, 1006.81]
    assert np.array_equal(E.undistort(img, K4, [0, 0, 0, 0], gpu_ctx), img)
    Kc = [2760.0, (cols - 1) / 2, 2760.0, (rows - 1) / 2]
    dist = [-0.11, 0.04, 0.0, 0.0]
    out = E.undistort(img, Kc, dist, gpu_ctx)
    out_rot = E.undistort(np.ascontiguousarray(img[::-1, ::-1]), Kc, dist, gpu_ctx)
    agree = (out_rot[::-1, ::-1] == out).mean()
    assert agree > 0.98            # rounding to 1/32 pixel is not symmetric about half-integers; the bulk agrees
    cy, cx = rows // 2, cols // 2
    assert np.array_equal(out[cy - 2:cy + 2, cx - 2:cx + 2], img[cy - 2:cy + 2, cx - 2:cx + 2])
    flat = np.full((rows, cols), 200, np.uint8)
    oflat = E.undistort(flat, Kc, [0.2, 0.0, 0.0, 0.0], gpu_ctx)     # pincushion: samples from outside near the border
    assert set(np.unique(oflat[200:-200, 300:-300])) == {200}
    assert oflat[0, 0] == 0


def test_undistort_rejects_bad_arguments(gpu_ctx):
    img = np.zeros((4, 4, 3), np.uint8)
    with pytest.raises(ValueError):
        E.undistort(np.zeros((4, 4, 2), np.uint8), [1, 1, 1, 1], [0, 0, 0, 0], gpu_ctx)
    with pytest.raises(ValueError):
        E.undistort(img, [1, 1, 1, 1], [0, 0, 0], gpu_ctx)

"""Host-side mirror of the reference's feature extraction over the C ABI:
``FeatureMatching::detectFeaturesSURF`` (cpp_code/src/feature_matching.cpp:43-69) and ``detectFeaturesORB`` (:14-41).  The
Hessian pyramid / FAST + Harris pyramid, the orientations and the descriptors are computed in libesfm_hip.so on the GPU.
ORB's 256 intensity tests use this library's own seeded point pairs: cv::ORB's learned table ships only inside OpenCV, so the
descriptors are ORB descriptors in kind, not bit-compatible with OpenCV's (esfm.h, oracle/orb_ref.c).  Also here: the image
undistortion that precedes detection (``MotionEstimator::doUnDistort``, cpp_code/src/estimate_motion.cpp:431-441) and
``DataIO::importDistort`` (cpp_code/src/data_io.cpp:97-125)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from ._lib import Context, check, default_context, lib
from .types import Frame


def surf_detect_and_compute(image, hessian_threshold: float = 100.0, max_keypoints: Optional[int] = None,
                            ctx: Optional[Context] = None) -> Tuple[np.ndarray, np.ndarray]:
    """esfm_surf_detect_and_compute.  image: [rows, cols] gray or [rows, cols, 3] BGR uint8.
    Returns (keypoints [n, 7] float32: x, y, size, angle, response, octave, class_id; descriptors [n, 64] float32)."""
    ctx = ctx or default_context()
    img = np.ascontiguousarray(image, np.uint8)
    if img.ndim == 2:
        rows, cols, ch = img.shape[0], img.shape[1], 1
    elif img.ndim == 3 and img.shape[2] == 3:
        rows, cols, ch = img.shape[0], img.shape[1], 3
    else:
        raise ValueError("image must be [rows, cols] or [rows, cols, 3] uint8")
    cap = int(max_keypoints) if max_keypoints is not None else rows * cols // 4 + 1024
    kp = np.zeros((max(cap, 1), 7), np.float32); desc = np.zeros((max(cap, 1), 64), np.float32)
    n = C.c_int32(0)
    check(lib().esfm_surf_detect_and_compute(ctx.handle, C.c_void_p(img.ctypes.data), rows, cols, ch, float(hessian_threshold), cap,
                                             C.c_void_p(kp.ctypes.data), C.c_void_p(desc.ctypes.data), C.byref(n)))
    return kp[:n.value].copy(), desc[:n.value].copy()


def detectFeaturesSURF(cur_frame: Frame, minHessian: int = 400, show: bool = False, ctx: Optional[Context] = None) -> bool:
    """feature_matching.cpp:43-69: fills cur_frame.keypoints (pt only, what the rest of the pipeline reads) and
    cur_frame.descriptors from cur_frame.rgb_image (BGR).  The full cv::KeyPoint fields are kept in cur_frame.keypoints_full."""
    if cur_frame.rgb_image is None:
        raise ValueError("frame has no image")
    kp, desc = surf_detect_and_compute(cur_frame.rgb_image, float(minHessian), None, ctx)
    cur_frame.keypoints = np.ascontiguousarray(kp[:, :2], np.float32)
    cur_frame.keypoints_full = kp
    cur_frame.descriptors = desc
    print(f"Found {len(kp)} features.")
    return True


def orb_detect_and_compute(image, nfeatures: int = 500, max_keypoints: Optional[int] = None,
                           ctx: Optional[Context] = None) -> Tuple[np.ndarray, np.ndarray]:
    """esfm_orb_detect_and_compute.  image: [rows, cols] gray or [rows, cols, 3] BGR uint8.
    Returns (keypoints [n, 7] float32: x, y, size, angle, response, octave, class_id; descriptors [n, 32] uint8)."""
    ctx = ctx or default_context()
    img = np.ascontiguousarray(image, np.uint8)
    if img.ndim == 2:
        rows, cols, ch = img.shape[0], img.shape[1], 1
    elif img.ndim == 3 and img.shape[2] == 3:
        rows, cols, ch = img.shape[0], img.shape[1], 3
    else:
        raise ValueError("image must be [rows, cols] or [rows, cols, 3] uint8")
    cap = int(max_keypoints) if max_keypoints is not None else 2 * int(nfeatures) + 4096      # retainBest keeps ties
    kp = np.zeros((max(cap, 1), 7), np.float32); desc = np.zeros((max(cap, 1), 32), np.uint8)
    n = C.c_int32(0)
    check(lib().esfm_orb_detect_and_compute(ctx.handle, C.c_void_p(img.ctypes.data), rows, cols, ch, int(nfeatures), cap,
                                            C.c_void_p(kp.ctypes.data), C.c_void_p(desc.ctypes.data), C.byref(n)))
    return kp[:n.value].copy(), desc[:n.value].copy()


def detectFeaturesORB(cur_frame: Frame, max_num: int = 5000, show: bool = False, ctx: Optional[Context] = None) -> bool:
    """feature_matching.cpp:14-41: cv::ORB::create(max_num) detect + compute on cur_frame.rgb_image (BGR)."""
    if cur_frame.rgb_image is None:
        raise ValueError("frame has no image")
    kp, desc = orb_detect_and_compute(cur_frame.rgb_image, int(max_num), None, ctx)
    cur_frame.keypoints = np.ascontiguousarray(kp[:, :2], np.float32)
    cur_frame.keypoints_full = kp
    cur_frame.descriptors = desc
    print(f"Found {len(kp)} features")
    return True


def undistort(image, K, dist, ctx: Optional[Context] = None) -> np.ndarray:
    """esfm_undistort: cv::undistort(image, out, K, dist) (estimate_motion.cpp:436).  image: [rows, cols] or [rows, cols, 3]
    uint8; K: 3 x 3 or (fx, cx, fy, cy); dist: k1, k2, p1, p2 as float64 (see import_distort for what the reference passes)."""
    ctx = ctx or default_context()
    img = np.ascontiguousarray(image, np.uint8)
    if img.ndim == 2:
        rows, cols, ch = img.shape[0], img.shape[1], 1
    elif img.ndim == 3 and img.shape[2] == 3:
        rows, cols, ch = img.shape[0], img.shape[1], 3
    else:
        raise ValueError("image must be [rows, cols] or [rows, cols, 3] uint8")
    K = np.asarray(K)
    if K.size == 9:
        K = K.reshape(3, 3)
        if K[0, 1] != 0 or K[1, 0] != 0 or K[2, 0] != 0 or K[2, 1] != 0 or K[2, 2] != 1:
            raise ValueError("camera matrix must be [fx 0 cx; 0 fy cy; 0 0 1]")
        k4 = np.array([K[0, 0], K[0, 2], K[1, 1], K[1, 2]], np.float64)     # eigen2cv of a Matrix3f, then convertTo CV_64F
    else:
        k4 = np.ascontiguousarray(K, np.float64).reshape(4)
    d4 = np.ascontiguousarray(dist, np.float64).reshape(-1)
    if d4.size != 4:
        raise ValueError("dist must hold k1, k2, p1, p2")
    out = np.empty_like(img)
    check(lib().esfm_undistort(ctx.handle, C.c_void_p(img.ctypes.data), rows, cols, ch, C.c_void_p(k4.ctypes.data),
                               C.c_void_p(d4.ctypes.data), C.c_void_p(out.ctypes.data)))
    return out


def import_distort(path: str) -> Optional[np.ndarray]:
    """DataIO::importDistort (data_io.cpp:97-125) as sfm.cpp:78-81 uses it.  Returns the 1 x 4 float64 coefficient matrix the
    reference hands to cv::undistort, or None when the file cannot be opened (the caller keeps zeros).  The file's k1 k2 p1 p2
    are read as floats (up to three groups of four, the last complete read wins) and stored with ``at<float>`` into a matrix that
    was created CV_64FC1 (sfm.cpp:78), so they land in the first 16 bytes: OpenCV sees k1' = the double whose low / high words
    are the bits of float k1 / k2, k2' likewise from (p1, p2), and p1' = p2' = 0.  Reproduced."""
    try:
        with open(path) as f:
            tokens = f.read().split()
    except OSError:
        return None
    vals = [0.0, 0.0, 0.0, 0.0]
    pos = 0
    for _ in range(3):
        for k in range(4):
            if pos >= len(tokens):
                break
            try:
                vals[k] = float(tokens[pos])
            except ValueError:
                vals[k] = 0.0          # a failed extraction stores zero (C++11) and stops the loop
                pos = len(tokens)
                break
            pos += 1
        if pos >= len(tokens):
            break
    coeff = np.zeros(4, np.float64)
    coeff.view(np.float32)[:4] = np.asarray(vals, np.float32)
    return coeff

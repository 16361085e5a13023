"""Host-side mirror of the reference's feature extraction over the C ABI (SURF half):
``FeatureMatching::detectFeaturesSURF`` (cpp_code/src/feature_matching.cpp:43-69).  The Hessian pyramid, the maxima, the
orientation and the 64-float descriptors are computed in libesfm_hip.so on the GPU.  ``detectFeaturesORB`` (:14-41) is not
built: cv::ORB's descriptor depends on a 256 x 4 learned sampling pattern that ships only inside OpenCV."""
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

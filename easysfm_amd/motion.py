"""Host-side mirror of the triangulation members of the reference's ``MotionEstimator`` over the C ABI (SURVEY.md section
8 row f-1, triangulation part): ``getDepthFast`` (cpp_code/src/estimate_motion.cpp:234-283), ``doTriangulation``
(:285-367) and ``outlierFilter`` (:476-505).  cv::triangulatePoints runs in libesfm_hip.so on the GPU
(``esfm_triangulate_points``); the statistical filter is the same kernel as ``CProceesing.SORFilter``;
``estimate2D2D_E5P_RANSAC`` (:27-97) = cv::findEssentialMat(RANSAC) + cv::recoverPose runs through ``esfm_find_essential_mat`` /
``esfm_recover_pose`` (hypotheses solved and scored on the GPU); ``estimate2D3D_P3P_RANSAC`` (:99-232) = cv::solvePnPRansac with
EPnP runs through ``esfm_solve_pnp_ransac``."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from ._lib import Context, check, default_context, lib
from .cloud import sor_filter
from .types import DMatch, Frame, SparsePointCloud


def triangulate_points(P1, P2, pts1, pts2, ctx: Optional[Context] = None) -> np.ndarray:
    """esfm_triangulate_points: cv::triangulatePoints(P1, P2, pts1, pts2) -> [n, 4] float32 homogeneous points."""
    ctx = ctx or default_context()
    P1 = np.ascontiguousarray(P1, np.float32).reshape(12); P2 = np.ascontiguousarray(P2, np.float32).reshape(12)
    a = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); b = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    if a.shape != b.shape:
        raise ValueError("point sets differ in size")
    n = a.shape[0]
    out = np.zeros((max(n, 1), 4), np.float32)
    check(lib().esfm_triangulate_points(ctx.handle, C.c_void_p(P1.ctypes.data), C.c_void_p(P2.ctypes.data), C.c_void_p(a.ctypes.data),
                                        C.c_void_p(b.ctypes.data), n, C.c_void_p(out.ctypes.data)))
    return out[:n]


def triangulate_pairs(P1s, P2s, point_offset, pts1, pts2, ctx: Optional[Context] = None) -> np.ndarray:
    """esfm_triangulate_pairs: many (P1, P2, point range) jobs in one launch."""
    ctx = ctx or default_context()
    P1s = np.ascontiguousarray(P1s, np.float32).reshape(-1, 12); P2s = np.ascontiguousarray(P2s, np.float32).reshape(-1, 12)
    off = np.ascontiguousarray(point_offset, np.int32)
    a = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); b = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    n = a.shape[0]
    out = np.zeros((max(n, 1), 4), np.float32)
    check(lib().esfm_triangulate_pairs(ctx.handle, P1s.shape[0], C.c_void_p(P1s.ctypes.data), C.c_void_p(P2s.ctypes.data),
                                       C.c_void_p(off.ctypes.data), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data),
                                       C.c_void_p(out.ctypes.data)))
    return out[:n]


def _k4(K) -> np.ndarray:
    K = np.asarray(K, np.float32)
    return np.array([K[0, 0], K[0, 2], K[1, 1], K[1, 2]], np.float32) if K.shape == (3, 3) else np.ascontiguousarray(K, np.float32).reshape(4)


def find_essential_mat(pts1, pts2, K, prob: float = 0.999, threshold: float = 1.0, ctx: Optional[Context] = None):
    """cv::findEssentialMat(pts1, pts2, K, RANSAC, prob, threshold, mask) -> (E [3,3] float64, mask [n] bool, iterations).
    K: 3x3 camera matrix or (fx, cx, fy, cy).  Raises EsfmError when OpenCV would return an empty matrix."""
    ctx = ctx or default_context()
    a = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); b = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    if a.shape != b.shape:
        raise ValueError("point sets differ in size")
    n = a.shape[0]
    k4 = _k4(K)
    E = np.zeros(9, np.float64); mask = np.zeros(max(n, 1), np.uint8); it = C.c_int32(0)
    check(lib().esfm_find_essential_mat(ctx.handle, C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), n, C.c_void_p(k4.ctypes.data),
                                        float(prob), float(threshold), C.c_void_p(E.ctypes.data), C.c_void_p(mask.ctypes.data), C.byref(it)))
    return E.reshape(3, 3), mask[:n].astype(bool), it.value


def find_essential_pairs(point_offset, pts1, pts2, K4_per_pair, prob: float = 0.999, threshold: float = 1.0, ctx: Optional[Context] = None):
    """esfm_find_essential_pairs: every pair's RANSAC in shared launches -> (E [p,3,3], mask [n_total] bool, status [p] bool,
    iterations [p])."""
    ctx = ctx or default_context()
    off = np.ascontiguousarray(point_offset, np.int32)
    n_pairs = len(off) - 1
    a = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); b = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    k4 = np.ascontiguousarray(K4_per_pair, np.float32).reshape(-1, 4)
    n = a.shape[0]
    E = np.zeros((max(n_pairs, 1), 9), np.float64); mask = np.zeros(max(n, 1), np.uint8)
    status = np.zeros(max(n_pairs, 1), np.int32); iters = np.zeros(max(n_pairs, 1), np.int32)
    check(lib().esfm_find_essential_pairs(ctx.handle, n_pairs, C.c_void_p(off.ctypes.data), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data),
                                          C.c_void_p(k4.ctypes.data), float(prob), float(threshold), C.c_void_p(E.ctypes.data),
                                          C.c_void_p(mask.ctypes.data), C.c_void_p(status.ctypes.data), C.c_void_p(iters.ctypes.data)))
    return E[:n_pairs].reshape(n_pairs, 3, 3), mask[:n].astype(bool), status[:n_pairs].astype(bool), iters[:n_pairs]


def recover_pose(E, pts1, pts2, K, mask=None, ctx: Optional[Context] = None):
    """cv::recoverPose(E, pts1, pts2, K, R, t, mask) -> (good, R [3,3], t [3], mask [n] bool or None)."""
    ctx = ctx or default_context()
    a = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); b = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    n = a.shape[0]
    k4 = _k4(K)
    Ev = np.ascontiguousarray(E, np.float64).reshape(9)
    R = np.zeros(9, np.float64); t = np.zeros(3, np.float64); good = C.c_int32(0)
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8)).copy()
    if m is not None and len(m) != n:
        raise ValueError("mask size")
    if m is not None and n == 0:
        m = np.zeros(1, np.uint8)
    check(lib().esfm_recover_pose(ctx.handle, C.c_void_p(Ev.ctypes.data), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), n,
                                  C.c_void_p(k4.ctypes.data), C.c_void_p(R.ctypes.data), C.c_void_p(t.ctypes.data),
                                  None if m is None else C.c_void_p(m.ctypes.data), C.byref(good)))
    return good.value, R.reshape(3, 3), t, (None if m is None else m[:n].astype(bool))


def recover_pose_pairs(point_offset, pts1, pts2, K4_per_pair, Es, mask=None, ctx: Optional[Context] = None):
    """esfm_recover_pose_pairs -> (good [p], R [p,3,3], t [p,3], mask [n_total] bool or None)."""
    ctx = ctx or default_context()
    off = np.ascontiguousarray(point_offset, np.int32)
    n_pairs = len(off) - 1
    a = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); b = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    k4 = np.ascontiguousarray(K4_per_pair, np.float32).reshape(-1, 4)
    Ev = np.ascontiguousarray(Es, np.float64).reshape(-1, 9)
    n = a.shape[0]
    R = np.zeros((max(n_pairs, 1), 9)); t = np.zeros((max(n_pairs, 1), 3)); good = np.zeros(max(n_pairs, 1), np.int32)
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8)).copy()
    if m is not None and len(m) == 0:
        m = np.zeros(1, np.uint8)
    check(lib().esfm_recover_pose_pairs(ctx.handle, n_pairs, C.c_void_p(off.ctypes.data), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data),
                                        C.c_void_p(k4.ctypes.data), C.c_void_p(Ev.ctypes.data), None if m is None else C.c_void_p(m.ctypes.data),
                                        C.c_void_p(R.ctypes.data), C.c_void_p(t.ctypes.data), C.c_void_p(good.ctypes.data)))
    return good[:n_pairs], R[:n_pairs].reshape(n_pairs, 3, 3), t[:n_pairs], (None if m is None else m[:n].astype(bool))


def solve_pnp_ransac(pts3d, pts2d, K, iterations_count: int = 100, reprojection_error: float = 8.0, confidence: float = 0.99,
                     ctx: Optional[Context] = None):
    """cv::solvePnPRansac(pts3d, pts2d, K, 0, rvec, tvec, false, iterationsCount, reprojectionError, confidence, inliers,
    SOLVEPNP_EPNP) -> (rvec [3], tvec [3], R [3,3], inlier mask [n] bool, iterations run).  Raises EsfmError where OpenCV
    returns false."""
    ctx = ctx or default_context()
    a = np.ascontiguousarray(pts3d, np.float32).reshape(-1, 3); b = np.ascontiguousarray(pts2d, np.float32).reshape(-1, 2)
    if a.shape[0] != b.shape[0]:
        raise ValueError("point sets differ in size")
    n = a.shape[0]
    k4 = _k4(K)
    rvec = np.zeros(3); tvec = np.zeros(3); R = np.zeros(9); mask = np.zeros(max(n, 1), np.uint8)
    ninl = C.c_int32(0); it = C.c_int32(0)
    check(lib().esfm_solve_pnp_ransac(ctx.handle, C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), n, C.c_void_p(k4.ctypes.data),
                                      int(iterations_count), float(reprojection_error), float(confidence), C.c_void_p(rvec.ctypes.data),
                                      C.c_void_p(tvec.ctypes.data), C.c_void_p(R.ctypes.data), C.c_void_p(mask.ctypes.data), C.byref(ninl),
                                      C.byref(it)))
    return rvec, tvec, R.reshape(3, 3), mask[:n].astype(bool), it.value


def ransac_sample_stream(count: int, n_samples: int) -> np.ndarray:
    """Host-only: the 5-index samples OpenCV's RANSAC draws for `count` points (esfm_ransac_sample_stream)."""
    idx = np.zeros((max(n_samples, 1), 5), np.int32)
    check(lib().esfm_ransac_sample_stream(int(count), int(n_samples), C.c_void_p(idx.ctypes.data)))
    return idx[:n_samples]


def five_point_models(q1, q2, ctx: Optional[Context] = None, host: bool = False, stages: bool = False):
    """esfm_five_point_models / esfm_five_point_models_host: the 5-point kernel alone on [n, 5, 2] normalised correspondences ->
    (E [n, 10, 3, 3], n_models [n]) (+ stages [n, 117] with stages=True).  host=True runs the host build of the kernels' routines
    (no GPU): the stage tests' middle leg between the GPU and the CPU restatement."""
    a = np.ascontiguousarray(q1, np.float64).reshape(-1, 5, 2); b = np.ascontiguousarray(q2, np.float64).reshape(-1, 5, 2)
    n = len(a)
    Es = np.zeros((max(n, 1), 10, 3, 3)); nm = np.zeros(max(n, 1), np.int32); st = np.zeros((max(n, 1), 117)) if stages else None
    stp = C.c_void_p(st.ctypes.data) if stages else None
    if host:
        check(lib().esfm_five_point_models_host(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), n, C.c_void_p(Es.ctypes.data), C.c_void_p(nm.ctypes.data), stp))
    else:
        ctx = ctx or default_context()
        check(lib().esfm_five_point_models(ctx.handle, C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), n, C.c_void_p(Es.ctypes.data),
                                           C.c_void_p(nm.ctypes.data), stp))
    return (Es[:n], nm[:n], st[:n]) if stages else (Es[:n], nm[:n])


def pixel2cam(p: np.ndarray, K: np.ndarray) -> np.ndarray:
    """estimate_motion.h:41-46, float arithmetic on the float K: ((u - cx) / fx, (v - cy) / fy)."""
    p = np.asarray(p, np.float32).reshape(-1, 2); K = np.asarray(K, np.float32)
    return np.stack([(p[:, 0] - K[0, 2]) / K[0, 0], (p[:, 1] - K[1, 2]) / K[1, 1]], axis=1).astype(np.float32)


def _dehomogenise(h: np.ndarray) -> np.ndarray:
    """pts_3d /= pts_3d.at<float>(3, 0) (estimate_motion.cpp:271, :341): float division."""
    return (h[:, :3] / h[:, 3:4]).astype(np.float32)


class MotionEstimator:
    """Mirror of p3dv::MotionEstimator (estimate_motion.h), triangulation members only."""

    def __init__(self, ctx: Optional[Context] = None):
        self._ctx = ctx

    def estimate2D2D_E5P_RANSAC(self, cur_frame_1: Frame, cur_frame_2: Frame, matches: Sequence[DMatch], inlier_matches: list,
                                ransac_thre: float = 1.0, ransac_prob: float = 0.99, show: bool = False) -> np.ndarray:
        """estimate_motion.cpp:27-97: essential matrix by 5-point RANSAC on the matched pixels (frame 1's K for both images,
        :43-44), inliers appended to `inlier_matches` from the RANSAC mask (:55-61), then recoverPose with that mask (:67).
        Returns T (4 x 4 float32, [R | t; 0 0 0 1], :78-85) mapping frame 1's camera coordinates to frame 2's."""
        k1 = np.asarray(cur_frame_1.keypoints, np.float32).reshape(-1, 2)[[m.queryIdx for m in matches]].reshape(-1, 2)
        k2 = np.asarray(cur_frame_2.keypoints, np.float32).reshape(-1, 2)[[m.trainIdx for m in matches]].reshape(-1, 2)
        E, mask, _ = find_essential_mat(k1, k2, cur_frame_1.K_cam, ransac_prob, ransac_thre, self._ctx)
        inlier_matches.extend(m for m, keep in zip(matches, mask) if keep)
        _, R, t, _ = recover_pose(E, k1, k2, cur_frame_1.K_cam, mask, self._ctx)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = R.astype(np.float32); T[:3, 3] = t.astype(np.float32)       # cv2eigen into Matrix3f / Vector3f (:76-79)
        print(f"Find [{int(mask.sum())}] inlier matches from [{len(matches)}] total matches.")
        return T

    def doUnDistort(self, cur_frame: Frame, distort_coeff) -> bool:
        """estimate_motion.cpp:431-441: cur_frame.rgb_image = cv::undistort(rgb_image, K_cam, distort_coeff)."""
        from .features import undistort
        if cur_frame.rgb_image is None:
            raise ValueError("frame has no image")
        cur_frame.rgb_image = undistort(cur_frame.rgb_image, np.asarray(cur_frame.K_cam, np.float32), distort_coeff, self._ctx)
        print("Undistort the image done.")
        return True

    def estimate2D3D_P3P_RANSAC(self, cur_frame: Frame, cur_map_3d: SparsePointCloud, ransac_thre: float = 2.5,
                                iterationsCount: int = 50000, ransac_prob: float = 0.99, show: bool = False) -> bool:
        """estimate_motion.cpp:99-232: 2-D/3-D correspondences by track id (every (keypoint, cloud point) pair with equal
        ids, keypoint-major, :122-151; cloud points beyond +-300 are skipped, :114,134), solvePnPRansac with EPnP (:161-162),
        pose written to cur_frame.pose_cam (:190-203), mean reprojection error over ALL correspondences (:207-219); returns
        False when that error > 10 and the inlier ratio < 0.5 (:226-230).  The reference's inlier bookkeeping reads the int
        index matrix as float (:169), so every entry addresses correspondence 0 and every other correspondence's cloud
        point gets is_inlier = 0; reproduced."""
        from .ba import angle_axis_to_rotation
        K = np.asarray(cur_frame.K_cam, np.float32)
        kp = np.asarray(cur_frame.keypoints, np.float32).reshape(-1, 2)
        xyz = np.asarray(cur_map_3d.xyz, np.float32).reshape(-1, 3)
        pid = np.asarray(cur_map_3d.unique_point_ids, np.int64)
        uid = np.asarray(cur_frame.unique_pixel_ids, np.int64)
        order = np.argsort(pid, kind="stable")                       # all cloud points with a given id, in cloud order
        lo = np.searchsorted(pid[order], uid, "left"); hi = np.searchsorted(pid[order], uid, "right")
        i2, j3 = [], []
        for i in range(len(uid)):
            for j in order[lo[i]:hi[i]]:
                if abs(xyz[j, 0]) < 300 and abs(xyz[j, 1]) < 300 and abs(xyz[j, 2]) < 300:
                    i2.append(i); j3.append(int(j))
        count = len(i2)
        print(f"{count} initial correspondences are used.")
        p2 = kp[i2].reshape(-1, 2); p3 = xyz[j3].reshape(-1, 3)
        rvec, tvec, R, mask, _ = solve_pnp_ransac(p3, p2, K, iterationsCount, ransac_thre, ransac_prob, self._ctx)
        n_inl = int(mask.sum())
        print(f"Inlier count: {n_inl}")
        index = list(j3)
        if n_inl > 0 and index:
            index[0] = -1                                            # inliers.at<float>(i, 0) on a CV_32S matrix: always 0 (:169)
        inl = np.asarray(cur_map_3d.is_inlier, np.int32).copy()
        for j in index:
            if j >= 0:
                inl[j] = 0
        cur_map_3d.is_inlier = inl
        R32 = angle_axis_to_rotation(rvec).astype(np.float32)        # cv::Rodrigues(r_vec, R_mat) then cv2eigen into Matrix3f (:184-190)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = R32; T[:3, 3] = tvec.astype(np.float32)
        cur_frame.pose_cam = T
        Xc = p3.astype(np.float64) @ angle_axis_to_rotation(rvec).T + tvec
        proj = np.stack([Xc[:, 0] / Xc[:, 2] * float(K[0, 0]) + float(K[0, 2]), Xc[:, 1] / Xc[:, 2] * float(K[1, 1]) + float(K[1, 2])], 1).astype(np.float32)
        reproj_err = float(np.mean(np.linalg.norm((proj - p2).astype(np.float64), axis=1))) if count else float("nan")
        inlier_ratio = n_inl / count if count else 0.0
        print(f"Mean reprojection error: {reproj_err}")
        return not (reproj_err > 10 and inlier_ratio < 0.5)

    def getDepthFast(self, cur_frame_1: Frame, cur_frame_2: Frame, T_21: np.ndarray, matches: Sequence[DMatch],
                     random_rate: int = 20) -> float:
        """estimate_motion.cpp:234-283: triangulate every `random_rate`-th match between the identity camera and T_21 and
        return the mean distance of the points from the first camera (in baseline lengths).  NaN for an empty sample, like
        the reference's 0 / 0."""
        T1 = np.eye(4, dtype=np.float32)[:3]
        T2 = (np.asarray(T_21, np.float32) @ np.eye(4, dtype=np.float32))[:3]
        sel = [m for i, m in enumerate(matches) if i % random_rate == 0]
        if not sel:
            return float("nan")
        k1 = np.asarray(cur_frame_1.keypoints, np.float32).reshape(-1, 2)[[m.queryIdx for m in sel]]
        k2 = np.asarray(cur_frame_2.keypoints, np.float32).reshape(-1, 2)[[m.trainIdx for m in sel]]
        h = triangulate_points(T1, T2, pixel2cam(k1, cur_frame_1.K_cam), pixel2cam(k2, cur_frame_1.K_cam), self._ctx)   # K of frame 1 for both (:245)
        p = _dehomogenise(h)
        depth_sum = 0.0
        for v in p:                                                     # Eigen::Vector3f::norm(), accumulated in double (:279)
            depth_sum += float(np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))
        return depth_sum / len(p)

    def doTriangulation(self, cur_frame_1: Frame, cur_frame_2: Frame, matches: Sequence[DMatch],
                        sparse_pointcloud: SparsePointCloud, show: bool = False, rgb_image: Optional[np.ndarray] = None) -> bool:
        """estimate_motion.cpp:285-367: matches whose track id (of frame 1's keypoint) is not in the cloud yet are
        triangulated from the two frames' poses and appended; colour from frame 1's image at the keypoint (BGR -> RGB,
        :347-353) when an image is given.  As in the reference the colour lookup indexes `matches[i]` with the index of
        the i-th NEW point (:345), which differs from the point's own match once any match was skipped."""
        T1 = np.asarray(cur_frame_1.pose_cam, np.float32)[:3]
        T2 = np.asarray(cur_frame_2.pose_cam, np.float32)[:3]
        known = set(int(v) for v in np.asarray(sparse_pointcloud.unique_point_ids).tolist())
        ids = np.asarray(cur_frame_1.unique_pixel_ids)
        new_q, new_t, new_ids = [], [], []
        for m in matches:
            uid = int(ids[m.queryIdx])
            if uid in known:
                continue
            known.add(uid)                                   # a repeated id inside this call is "already in world" too (:309-316)
            new_ids.append(uid); new_q.append(m.queryIdx); new_t.append(m.trainIdx)
        if new_ids:
            k1 = np.asarray(cur_frame_1.keypoints, np.float32).reshape(-1, 2)
            k2 = np.asarray(cur_frame_2.keypoints, np.float32).reshape(-1, 2)
            h = triangulate_points(T1, T2, pixel2cam(k1[new_q], cur_frame_1.K_cam), pixel2cam(k2[new_t], cur_frame_1.K_cam), self._ctx)
            xyz = _dehomogenise(h)
            rgb = np.zeros((len(new_ids), 3), np.uint8)
            if rgb_image is not None:
                for i in range(len(new_ids)):
                    px = k1[matches[i].queryIdx]                                  # :345, the reference's indexing
                    v = np.asarray(rgb_image)[int(px[1]), int(px[0])]
                    b, g, r = (v, v, v) if np.ndim(v) == 0 else v[:3]                # a gray image colours its points gray
                    rgb[i] = (r, g, b)
            old_xyz = np.asarray(sparse_pointcloud.xyz, np.float32).reshape(-1, 3)
            old_rgb = np.asarray(sparse_pointcloud.rgb, np.uint8).reshape(-1, 3)
            if len(old_rgb) != len(old_xyz):
                old_rgb = np.zeros((len(old_xyz), 3), np.uint8)
            sparse_pointcloud.xyz = np.concatenate([old_xyz, xyz])
            sparse_pointcloud.rgb = np.concatenate([old_rgb, rgb])
            sparse_pointcloud.unique_point_ids = np.concatenate([np.asarray(sparse_pointcloud.unique_point_ids, np.int64), np.array(new_ids, np.int64)])
            sparse_pointcloud.is_inlier = np.concatenate([np.asarray(sparse_pointcloud.is_inlier, np.int32), np.ones(len(new_ids), np.int32)])
        print(f"Triangulate [ {len(new_ids)} ] new points, [ {len(sparse_pointcloud.xyz)} ] points in total.")
        return True

    def outlierFilter(self, sparse_pointcloud: SparsePointCloud, MeanK: int = 40, std: float = 2.5) -> bool:
        """estimate_motion.cpp:476-505: statistical outlier removal that keeps ids and inlier flags aligned."""
        xyz = np.asarray(sparse_pointcloud.xyz, np.float32).reshape(-1, 3)
        keep, _, _ = sor_filter(xyz, MeanK, std, self._ctx)
        n0 = len(xyz)
        sparse_pointcloud.xyz = xyz[keep].copy()
        if len(sparse_pointcloud.rgb) == n0:
            sparse_pointcloud.rgb = np.asarray(sparse_pointcloud.rgb)[keep].copy()
        sparse_pointcloud.unique_point_ids = np.asarray(sparse_pointcloud.unique_point_ids)[keep].copy()
        sparse_pointcloud.is_inlier = np.asarray(sparse_pointcloud.is_inlier)[keep].copy()
        print(f"apply outlier filter: [ {n0} ] points before filtering, [ {int(keep.sum())} ] points after filtering.")
        return True

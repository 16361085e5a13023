"""Host-side mirror of the reference's bundle-adjustment interface over the C ABI.

``BundleAdjustment`` keeps the member names and argument meaning of cpp_code/include/ba.h:28-106
(initBA, setBAProblem, solveBA, doSFMBA); ``ba_solve`` is the array-level call that maps 1:1 onto
``esfm_ba_solve``.  The LM/Schur solve runs in libesfm_hip.so on the GPU; the only arithmetic done
here is the reference's own host-side packing (Rodrigues conversion, float truncation on write-back).
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._lib import ALLREDUCE_FN, BAOptions, BASummary, Context, ESFM_REDUCE_MAX, ESFM_REDUCE_SUM, check, default_context, lib
from .types import Frame, SparsePointCloud


# Bytes per observation the Jacobian sweep (ba_linearize_kernel) is BUILT to move: 16 in (cam / point index, uv) + Jc 96 + Jp 48 +
# res 16 out = SURVEY 8(d)'s compulsory 176 (round 2 also wrote W = F'E, 144 B more; round 3 re-forms it in the Schur kernels).
BA_SWEEP_BYTES_PER_OBS = 176


def ba_sweep_bytes_per_obs() -> int:
    return BA_SWEEP_BYTES_PER_OBS


def default_options() -> BAOptions:
    o = BAOptions()
    lib().esfm_ba_options_default(C.byref(o))
    return o


def _p(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


_NULL_ALLREDUCE = C.cast(None, ALLREDUCE_FN)


class _DevArray:
    """Zero-copy view of a device pointer for torch.as_tensor (__cuda_array_interface__)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def torch_allreduce_callback(group=None):
    """esfm_allreduce_fn backed by torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).
    The collective is enqueued on the hipStream_t the library passes (the solver's stream), wrapped as a
    torch ExternalStream, so it is ordered after the kernels that produced the buffer and before the ones that
    consume it whatever torch's current stream is.  The library's own RCCL path (esfm_comm_*, include/esfm.h)
    needs no callback at all; this one serves torch-managed process groups (and gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist

    def _cb(user, buf, count, op, stream):
        try:
            dev = torch.device("cuda", torch.cuda.current_device())
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)) if stream else contextlib.nullcontext():
                t = torch.as_tensor(_DevArray(buf, count), device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM if op == ESFM_REDUCE_SUM else dist.ReduceOp.MAX, group=group)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print(f"[easysfm_amd] all-reduce callback failed: {e!r}", flush=True)
            return 1

    return ALLREDUCE_FN(_cb)


class Comm:
    """esfm_comm: the library's own RCCL communicator (one rank per GPU / esfm_ctx).  Pass an instance as ``allreduce=`` to
    BAProblem.solve / ba_solve / ba_solve_ex: the solver then calls esfm_comm_allreduce (ncclAllReduce on the context's stream)
    itself, with no callback into Python.  ``unique_id()`` runs on rank 0; the 128 bytes reach the other ranks by whatever the
    host program has (bench.py broadcasts them through its torch.distributed store)."""

    def __init__(self, ctx: Context, uid: bytes, rank: int, world: int):
        if len(uid) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self._h = C.c_void_p()
        buf = C.create_string_buffer(bytes(uid), 128)
        check(lib().esfm_comm_create(ctx.handle, buf, self.rank, self.world, C.byref(self._h)))

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(lib().esfm_comm_get_unique_id(buf))
        return buf.raw

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def rccl_ranks(self) -> int:
        """ncclCommCount of the communicator: the ranks RCCL itself sees."""
        return int(lib().esfm_comm_rccl_ranks(self._h))

    def allreduce_(self, dev_ptr: int, count: int, op: int = ESFM_REDUCE_SUM) -> None:
        """In-place all-reduce of `count` doubles at a device pointer on the context's stream (tests)."""
        rc = lib().esfm_comm_allreduce(self._h, C.c_void_p(dev_ptr), int(count), int(op), C.c_void_p(self.ctx.stream))
        if rc != 0:
            raise RuntimeError("esfm_comm_allreduce failed: " + lib().esfm_last_error().decode("utf-8", "replace"))

    def close(self) -> None:
        """Collective teardown (ncclCommDestroy).  Call it explicitly: a communicator that is only garbage-collected at
        interpreter exit is left to the process teardown, because RCCL aborts when destroyed after the HIP runtime has gone."""
        if self._h:
            lib().esfm_comm_destroy(self._h)
            self._h = C.c_void_p()


class BAProblem:
    """Resident problem (esfm_ba_problem): upload once, solve/iterate many times.

    ``calib`` (fx, cx, fy, cy) switches to the free shared intrinsics of ReprojectErrorTerm_updatecalib (ba.h:170-222),
    bounded to ``calib +- calib_tol`` (ba.cpp:190-194); ``K4`` is then ignored.  ``ref_cam >= 0`` bounds that camera's
    six parameters to ``+-ref_threshold`` (ba.cpp:155-162)."""

    def __init__(self, cam_idx, pt_idx, uv, K4, cams, pts, ctx: Optional[Context] = None, calib=None, calib_tol: float = 0.0,
                 ref_cam: int = -1, ref_threshold: float = 1e-10):
        self.ctx = ctx or default_context()
        self.cam_idx = np.ascontiguousarray(cam_idx, np.int32).reshape(-1)
        self.pt_idx = np.ascontiguousarray(pt_idx, np.int32).reshape(-1)
        self.uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        cams = np.ascontiguousarray(cams, np.float64).reshape(-1, 6)
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        self.n_cam, self.n_pt, self.n_obs = cams.shape[0], pts.shape[0], self.cam_idx.shape[0]
        self.free_calib = calib is not None
        if self.pt_idx.shape[0] != self.n_obs or self.uv.shape[0] != self.n_obs:
            raise ValueError("inconsistent BA array sizes")
        self._h = C.c_void_p()
        if self.free_calib:
            cal = np.ascontiguousarray(calib, np.float64).reshape(4)
            check(lib().esfm_ba_problem_create_free_calib(self.ctx.handle, self.n_cam, self.n_pt, self.n_obs, _p(self.cam_idx),
                                                          _p(self.pt_idx), _p(self.uv), _p(cal), float(calib_tol), _p(cams), _p(pts),
                                                          C.byref(self._h)))
        else:
            self.K4 = np.ascontiguousarray(K4, np.float32).reshape(-1, 4)
            if self.K4.shape[0] != self.n_cam:
                raise ValueError("inconsistent BA array sizes")
            check(lib().esfm_ba_problem_create(self.ctx.handle, self.n_cam, self.n_pt, self.n_obs, _p(self.cam_idx), _p(self.pt_idx),
                                               _p(self.uv), _p(self.K4), _p(cams), _p(pts), C.byref(self._h)))
        if ref_cam >= 0:
            self.fix_camera(ref_cam, ref_threshold)

    def fix_camera(self, cam: int, threshold: float = 1e-10) -> None:
        """Hold camera `cam` inside [-threshold, threshold]^6 (the reference frame, ba.cpp:134, :155-162); cam < 0 releases."""
        check(lib().esfm_ba_problem_fix_camera(self._h, int(cam), float(threshold)))

    def set_calib(self, calib, calib_tol: float) -> None:
        cal = np.ascontiguousarray(calib, np.float64).reshape(4)
        check(lib().esfm_ba_problem_set_calib(self._h, _p(cal), float(calib_tol)))

    def get_calib(self) -> np.ndarray:
        cal = np.empty(4, np.float64)
        check(lib().esfm_ba_problem_get_calib(self._h, _p(cal)))
        return cal

    def set_params(self, cams, pts) -> None:
        cams = np.ascontiguousarray(cams, np.float64).reshape(-1, 6); pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        check(lib().esfm_ba_problem_set_params(self._h, _p(cams), _p(pts)))

    def solve(self, options: Optional[BAOptions] = None, allreduce=None) -> BASummary:
        summ = BASummary()
        opt = options if options is not None else default_options()
        user = None
        if isinstance(allreduce, Comm):         # the library's own RCCL path: a C function pointer and its communicator
            cb, user = C.cast(lib().esfm_comm_allreduce, ALLREDUCE_FN), allreduce.handle
        else:
            cb = allreduce if allreduce is not None else _NULL_ALLREDUCE
        check(lib().esfm_ba_problem_solve(self._h, C.byref(opt), cb, user, C.byref(summ)))
        return summ

    def get_params(self) -> Tuple[np.ndarray, np.ndarray]:
        cams = np.empty((self.n_cam, 6), np.float64); pts = np.empty((self.n_pt, 3), np.float64)
        check(lib().esfm_ba_problem_get_params(self._h, _p(cams), _p(pts)))
        return cams, pts

    def cost(self, cauchy_a: float = 0.5) -> float:
        c = C.c_double(0.0)
        check(lib().esfm_ba_problem_cost(self._h, float(cauchy_a), C.byref(c)))
        return c.value

    def close(self) -> None:
        if self._h:
            lib().esfm_ba_problem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ba_solve(cam_idx, pt_idx, uv, K4, cams, pts, options: Optional[BAOptions] = None, ctx: Optional[Context] = None,
             allreduce=None) -> Tuple[np.ndarray, np.ndarray, BASummary]:
    """esfm_ba_solve: returns (cams, pts, summary); the inputs are not modified."""
    prob = BAProblem(cam_idx, pt_idx, uv, K4, cams, pts, ctx)
    try:
        summ = prob.solve(options, allreduce)
        c, p = prob.get_params()
    finally:
        prob.close()
    return c, p, summ


def ba_solve_ex(cam_idx, pt_idx, uv, K4, cams, pts, calib=None, calib_tol: float = 0.0, ref_cam: int = -1,
                ref_threshold: float = 1e-10, options: Optional[BAOptions] = None, ctx: Optional[Context] = None, allreduce=None):
    """solveBA with the optional free intrinsics / reference camera (esfm_ba_solve_ex).
    Returns (cams, pts, calib or None, summary)."""
    prob = BAProblem(cam_idx, pt_idx, uv, K4, cams, pts, ctx, calib=calib, calib_tol=calib_tol, ref_cam=ref_cam,
                     ref_threshold=ref_threshold)
    try:
        summ = prob.solve(options, allreduce)
        c, p = prob.get_params()
        k = prob.get_calib() if prob.free_calib else None
    finally:
        prob.close()
    return c, p, k, summ


def line_search_next_step(f0: float, g0: float, prev, cur) -> float:
    """Host-only: next Armijo trial step (esfm_ba_line_search_next_step); prev / cur = (x, f, g) or None."""
    xp, fp, gp = prev if prev is not None else (0.0, 0.0, 0.0)
    xc, fc, gc = cur if cur is not None else (0.0, 0.0, 0.0)
    return float(lib().esfm_ba_line_search_next_step(f0, g0, xp, fp, gp, int(prev is not None), xc, fc, gc, int(cur is not None)))


def shard_points(n_pt: int, pt_idx, world: int) -> np.ndarray:
    """Point -> shard assignment balanced by observation count (host-only, no GPU)."""
    pt_idx = np.ascontiguousarray(pt_idx, np.int32)
    out = np.zeros(max(n_pt, 1), np.int32)
    check(lib().esfm_ba_shard_points(int(n_pt), len(pt_idx), _p(pt_idx), int(world), _p(out)))
    return out[:n_pt]


def reduced_plan(n_cam: int, n_pt: int, cam_idx, pt_idx, leaf_max: int = 0) -> dict:
    """esfm_ba_reduced_plan (host-only, no GPU): the structure-aware plan of the reduced camera system for an observation list --
    nested-dissection order of the cameras (col_src: original unknown of every permuted, tile-padded column; -1 = padding), the
    tiles of the symbolic fill, the dependency chain length.  What the tiled solve builds for itself; for tests and tools."""
    cam_idx = np.ascontiguousarray(cam_idx, np.int32); pt_idx = np.ascontiguousarray(pt_idx, np.int32)
    info = np.zeros(10, np.int32)
    check(lib().esfm_ba_reduced_plan(int(n_cam), int(n_pt), len(cam_idx), _p(cam_idx), _p(pt_idx), int(leaf_max), None, 0, None, 0, _p(info)))
    col_src = np.zeros(max(int(info[0]) * 64, 1), np.int32); tiles = np.zeros((max(int(info[1]), 1), 2), np.int32)
    check(lib().esfm_ba_reduced_plan(int(n_cam), int(n_pt), len(cam_idx), _p(cam_idx), _p(pt_idx), int(leaf_max), _p(col_src), len(col_src),
                                     _p(tiles), len(tiles), _p(info)))
    return dict(nb=int(info[0]), col_src=col_src[:int(info[0]) * 64], tiles=tiles[:int(info[1])], chain=int(info[2]), dense_nb=int(info[3]),
                worthwhile=bool(info[4]), update_steps=int(info[5]), supernodes=int(info[6]), workgroups=int(info[7]), covisible_blocks=int(info[8]))


# ------------------------------------------------------------------------------------------------
# cv::Rodrigues as the reference uses it (ba.cpp:82 matrix -> vector, :239 vector -> matrix).
def rotation_to_angle_axis(R: np.ndarray) -> np.ndarray:
    R = np.asarray(R, np.float64)
    # orthonormalise first, as OpenCV does (R = U V^T) [upstream calib3d Rodrigues]
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((r * r).sum() * 0.25)
    c = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (np.diag(R) + 1.0) * 0.5
        v = np.sqrt(np.maximum(t, 0.0))
        if R[0, 1] < 0: v[1] = -v[1]
        if R[0, 2] < 0: v[2] = -v[2]
        if (R[1, 2] > 0) != (v[1] * v[2] > 0): v[2] = -v[2]
        return v * (theta / max(np.linalg.norm(v), 1e-300))
    return r * (0.5 * theta / s)


def angle_axis_to_rotation(aa: np.ndarray) -> np.ndarray:
    aa = np.asarray(aa, np.float64)
    theta = np.linalg.norm(aa)
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = aa / theta
    c, s = np.cos(theta), np.sin(theta)
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * Kx


class BundleAdjustment:
    """Mirror of p3dv::BundleAdjustment (ba.h:28-106)."""

    def __init__(self, ctx: Optional[Context] = None, options: Optional[BAOptions] = None):
        self._ctx = ctx
        self.options = options
        self.initBA()

    def initBA(self) -> bool:
        """ba.h:59-75."""
        self.point_index_ = np.zeros(0, np.int32)
        self.camera_index_ = np.zeros(0, np.int32)
        self.points_2d_ = np.zeros((0, 2), np.float32)
        self.calibs_: List[np.ndarray] = []
        self.num_cameras_ = 0
        self.num_points_ = 0
        self.num_parameters_ = 0
        self.num_observations_ = 0
        self.ref_process_camera_id_ = -1
        self.parameters_ = np.zeros(0, np.float64)
        self.summary: Optional[BASummary] = None
        return True

    def setBAProblem(self, frames: Sequence[Frame], process_frame_id: Sequence[bool], sfm_sparse_points: SparsePointCloud,
                     fix_calib_tolerance_BA: float = 0.0, reference_frame_id: int = -1) -> bool:
        """ba.cpp:12-130.  Same observation list, in the same order (camera-major, point index
        ascending), as the reference's O(Ncam*Npts*Nkp) triple loop (:22-56), derived with an
        O(N log N) join: for every registered frame, each point picks the FIRST keypoint j with
        unique_pixel_has_match[j] and unique_pixel_ids[j] == unique_point_ids[k] (the `break` at :44)."""
        pid = np.asarray(sfm_sparse_points.unique_point_ids, np.int64)
        self.num_points_ = len(pid)
        cams_i, pts_i, uv = [], [], []
        self.num_cameras_ = 0
        for i, fr in enumerate(frames):
            if process_frame_id[i]:  # 0 = registered
                continue
            self.calibs_.append(np.asarray(fr.K_cam, np.float32))
            ids = np.asarray(fr.unique_pixel_ids, np.int64)
            cand = np.nonzero(np.asarray(fr.unique_pixel_has_match, bool))[0]
            if len(cand) and len(pid):
                cid = ids[cand]
                order = np.lexsort((cand, cid))            # by id, then lowest keypoint index first
                cid_s, cand_s = cid[order], cand[order]
                first = np.ones(len(cid_s), bool); first[1:] = cid_s[1:] != cid_s[:-1]
                u_id, u_j = cid_s[first], cand_s[first]
                pos = np.searchsorted(u_id, pid)
                pos_c = np.minimum(pos, len(u_id) - 1)
                hit = u_id[pos_c] == pid
                k_idx = np.nonzero(hit)[0]
                j_idx = u_j[pos_c[hit]]
                cams_i.append(np.full(len(k_idx), self.num_cameras_, np.int32))
                pts_i.append(k_idx.astype(np.int32))
                uv.append(np.asarray(fr.keypoints, np.float32).reshape(-1, 2)[j_idx])
            if i == reference_frame_id:
                self.ref_process_camera_id_ = self.num_cameras_
            self.num_cameras_ += 1
        self.camera_index_ = np.concatenate(cams_i) if cams_i else np.zeros(0, np.int32)
        self.point_index_ = np.concatenate(pts_i) if pts_i else np.zeros(0, np.int32)
        self.points_2d_ = np.concatenate(uv) if uv else np.zeros((0, 2), np.float32)
        self.num_observations_ = len(self.camera_index_)
        self.num_parameters_ = 6 * self.num_cameras_ + 3 * self.num_points_ + (4 if fix_calib_tolerance_BA != 0 else 0)
        par = np.zeros(self.num_parameters_, np.float64)
        k = 0
        for i, fr in enumerate(frames):
            if process_frame_id[i]:
                continue
            pose = np.asarray(fr.pose_cam, np.float32)
            rvec = rotation_to_angle_axis(pose[:3, :3]).astype(np.float32)   # cv::Rodrigues on a CV_32F matrix (:82)
            par[6 * k:6 * k + 3] = rvec.astype(np.float64)
            par[6 * k + 3:6 * k + 6] = pose[:3, 3].astype(np.float64)
            k += 1
        xyz = np.asarray(sfm_sparse_points.xyz, np.float32).reshape(-1, 3)
        par[6 * self.num_cameras_:6 * self.num_cameras_ + 3 * self.num_points_] = xyz.astype(np.float64).reshape(-1)
        if fix_calib_tolerance_BA != 0 and self.calibs_:
            K0 = self.calibs_[0]
            par[-4:] = [K0[0, 0], K0[0, 2], K0[1, 1], K0[1, 2]]
        self.parameters_ = par
        return 2 * self.num_observations_ > self.num_parameters_   # "Ready to solve" (:120-129)

    def solveBA(self, fix_calib_tolerance_BA: float = 0.0) -> bool:
        """ba.cpp:132-212: fixed intrinsics (:142-164) or the shared free intrinsics bounded to +-tolerance (:167-196);
        the reference frame, if one was named in setBAProblem, is bounded to +-1e-10 (:134, :155-162)."""
        fixed_threshold = 1e-10   # ba.cpp:134
        nc, npt = self.num_cameras_, self.num_points_
        K4 = np.array([[K[0, 0], K[0, 2], K[1, 1], K[1, 2]] for K in self.calibs_], np.float32).reshape(-1, 4)
        cams = self.parameters_[:6 * nc].reshape(nc, 6)
        pts = self.parameters_[6 * nc:6 * nc + 3 * npt].reshape(npt, 3)
        calib = self.parameters_[-4:].copy() if fix_calib_tolerance_BA != 0 else None
        c, p, k, summ = ba_solve_ex(self.camera_index_, self.point_index_, self.points_2d_, K4, cams, pts, calib=calib,
                                    calib_tol=float(fix_calib_tolerance_BA), ref_cam=self.ref_process_camera_id_,
                                    ref_threshold=fixed_threshold, options=self.options, ctx=self._ctx or default_context())
        self.parameters_[:6 * nc] = c.reshape(-1)
        self.parameters_[6 * nc:6 * nc + 3 * npt] = p.reshape(-1)
        if k is not None:
            self.parameters_[-4:] = k
        self.summary = summ
        return True

    def doSFMBA(self, frames: Sequence[Frame], process_frame_id: Sequence[bool], sfm_sparse_points: SparsePointCloud,
                fix_calib_tolerance_BA: float = 0.0, reference_frame_id: int = -1) -> bool:
        """ba.cpp:214-288: initBA, setBAProblem, solveBA, then the float write-back (:223-281)."""
        self.initBA()
        self.setBAProblem(frames, process_frame_id, sfm_sparse_points, fix_calib_tolerance_BA, reference_frame_id)
        self.solveBA(fix_calib_tolerance_BA)
        k = 0
        for i, fr in enumerate(frames):
            if process_frame_id[i]:
                continue
            rvec = self.parameters_[6 * k:6 * k + 3].astype(np.float32)                 # :235-237 (float)
            pose = np.array(fr.pose_cam, np.float32, copy=True)
            pose[:3, :3] = angle_axis_to_rotation(rvec.astype(np.float64)).astype(np.float32)
            pose[:3, 3] = self.parameters_[6 * k + 3:6 * k + 6].astype(np.float32)      # :244-246
            fr.pose_cam = pose
            k += 1
            if fix_calib_tolerance_BA != 0:                                            # :250-256 (float K_cam)
                K = np.array(fr.K_cam, np.float32, copy=True)
                K[0, 0], K[0, 2], K[1, 1], K[1, 2] = self.parameters_[-4:].astype(np.float32)
                fr.K_cam = K
        nc, npt = self.num_cameras_, self.num_points_
        sfm_sparse_points.xyz = self.parameters_[6 * nc:6 * nc + 3 * npt].reshape(npt, 3).astype(np.float32)  # :277-279
        return True

"""ctypes front-end of the CPU oracle (oracle/match_ref.c, oracle/ba_ref.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (easysfm_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None
_LIB_PATH: Optional[str] = None

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


class BAOptions(C.Structure):
    """Mirror of esfm_ba_options (include/esfm.h)."""
    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("verbose", C.c_int32),
        ("cauchy_a", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
    ]


class BAIteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32),
        ("step_is_valid", C.c_int32),
        ("step_is_successful", C.c_int32),
        ("line_search_steps", C.c_int32),
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double),
        ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
        ("model_cost_change", C.c_double),
    ]


BA_MAX_LOG = 256


class BASummary(C.Structure):
    _fields_ = [
        ("termination", C.c_int32),
        ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("num_unsuccessful_steps", C.c_int32),
        ("num_active_cameras", C.c_int32),
        ("num_active_points", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("solve_seconds", C.c_double),
        ("iterations", BAIteration * BA_MAX_LOG),
    ]


def build(arch: str = "x86-64-v3", out: str = "libesfm_oracle.so", force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Returns the .so path."""
    path = os.path.join(_HERE, out)
    srcs = [os.path.join(_HERE, f) for f in ("match_ref.c", "ba_ref.c", "cloud_ref.c", "geometry_ref.c", "ransac_ref.c", "pnp_ref.c", "surf_ref.c", "orb_ref.c", "undistort_ref.c", "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "esfm.h"))
    if force or not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        subprocess.run(["make", "-B", "-C", _HERE, f"ARCH={arch}", f"OUT={out}"], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return path


def load(path: Optional[str] = None) -> C.CDLL:
    """Load (building if needed) the oracle shared library and set signatures."""
    global _LIB, _LIB_PATH
    if path is None:
        path = os.path.join(_HERE, "libesfm_oracle.so")
        if not os.path.exists(path):
            build()
    if _LIB is not None and _LIB_PATH == path:
        return _LIB
    lib = C.CDLL(path)
    lib.esfm_ref_l2sqr.restype = C.c_float
    lib.esfm_ref_l2sqr.argtypes = [_f32p, _f32p, C.c_int]
    lib.esfm_ref_knn2_l2_f32.restype = None
    lib.esfm_ref_knn2_l2_f32.argtypes = [_f32p, C.c_int, _f32p, C.c_int, C.c_int, _i32p, _f32p]
    lib.esfm_ref_knn2_l2_f32_scalar.restype = None
    lib.esfm_ref_knn2_l2_f32_scalar.argtypes = [_f32p, C.c_int, _f32p, C.c_int, C.c_int, _i32p, _f32p]
    lib.esfm_ref_match_pairs_l2.restype = C.c_int
    lib.esfm_ref_match_pairs_l2.argtypes = [_f32p, _i32p, C.c_int, C.c_int, _i32p, C.c_int, C.c_double, _i32p, _i32p, _f32p, _i32p,
                                            np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")]
    lib.esfm_ref_knn2_hamming.restype = None
    lib.esfm_ref_knn2_hamming.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.c_int, _i32p, _f32p]
    lib.esfm_ref_ratio_filter.restype = C.c_int
    lib.esfm_ref_ratio_filter.argtypes = [_i32p, _f32p, C.c_int, C.c_double, _i32p, _i32p, _f32p]
    lib.esfm_ref_num_threads.restype = C.c_int
    lib.esfm_ref_set_num_threads.restype = None
    lib.esfm_ref_set_num_threads.argtypes = [C.c_int]
    lib.esfm_ref_ba_residual_jac.restype = None
    lib.esfm_ref_ba_residual_jac.argtypes = [_f64p, _f64p, _f32p, _f32p, _f64p, _f64p, _f64p]
    lib.esfm_ref_ba_cost.restype = C.c_double
    lib.esfm_ref_ba_cost.argtypes = [C.c_int, _i32p, _i32p, _f32p, _f32p, _f64p, _f64p, C.c_double]
    lib.esfm_ref_ba_options_default.restype = None
    lib.esfm_ref_ba_options_default.argtypes = [C.POINTER(BAOptions)]
    lib.esfm_ref_ba_solve.restype = C.c_int
    lib.esfm_ref_ba_solve.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f32p, _f32p, _f64p, _f64p,
                                      C.POINTER(BAOptions), C.POINTER(BASummary)]
    lib.esfm_ref_ba_solve_ex.restype = C.c_int
    lib.esfm_ref_ba_solve_ex.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f32p, C.c_void_p, _f64p, _f64p,
                                         C.c_void_p, C.c_double, C.c_int, C.c_double,
                                         C.POINTER(BAOptions), C.POINTER(BASummary)]
    lib.esfm_ref_ba_residual_jac_calib.restype = None
    lib.esfm_ref_ba_residual_jac_calib.argtypes = [_f64p, _f64p, _f64p, _f32p, _f64p, _f64p, _f64p, _f64p]
    lib.esfm_ref_ba_cost_calib.restype = C.c_double
    lib.esfm_ref_ba_cost_calib.argtypes = [C.c_int, _i32p, _i32p, _f32p, _f64p, _f64p, _f64p, C.c_double]
    lib.esfm_ref_ls_next_step.restype = C.c_double
    lib.esfm_ref_ls_next_step.argtypes = [C.c_double] * 2 + [C.c_double] * 3 + [C.c_int] + [C.c_double] * 3 + [C.c_int] + [C.c_double] * 2
    lib.esfm_ref_ba_partial_reduced.restype = C.c_int
    lib.esfm_ref_ba_partial_reduced.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f32p, _f32p, _f64p, _f64p,
                                                C.c_double, C.c_double, _f64p, _f64p, C.c_int, _f64p, _f64p]
    lib.esfm_ref_ba_column_sqnorms.restype = C.c_int
    lib.esfm_ref_ba_column_sqnorms.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f32p, _f32p, _f64p, _f64p,
                                               C.c_double, _f64p, _f64p]
    lib.esfm_ref_sor_mean_distances.restype = C.c_int
    lib.esfm_ref_sor_mean_distances.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p]
    lib.esfm_ref_sor_filter.restype = C.c_int
    lib.esfm_ref_sor_filter.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_double, _f32p, _u8p, C.POINTER(C.c_double)]
    lib.esfm_ref_triangulate_points.restype = None
    lib.esfm_ref_triangulate_points.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int, _f32p]
    lib.esfm_ref_ransac_samples.restype = None
    lib.esfm_ref_ransac_samples.argtypes = [C.c_int, C.c_int, _i32p]
    lib.esfm_ref_five_point.restype = C.c_int
    lib.esfm_ref_five_point.argtypes = [_f64p, _f64p, _f64p]
    lib.esfm_ref_five_point_stages.restype = C.c_int
    lib.esfm_ref_five_point_stages.argtypes = [_f64p, _f64p, _f64p]
    lib.esfm_ref_find_essential_ransac.restype = C.c_int
    lib.esfm_ref_find_essential_ransac.argtypes = [_f32p, _f32p, C.c_int, _f32p, C.c_double, C.c_double, _f64p, _u8p,
                                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.esfm_ref_decompose_essential.restype = None
    lib.esfm_ref_decompose_essential.argtypes = [_f64p, _f64p, _f64p, _f64p]
    lib.esfm_ref_recover_pose.restype = C.c_int
    lib.esfm_ref_recover_pose.argtypes = [_f64p, _f32p, _f32p, C.c_int, _f32p, _f64p, _f64p, _u8p]
    lib.esfm_ref_epnp.restype = C.c_double
    lib.esfm_ref_epnp.argtypes = [_f64p, _f64p, C.c_int, _f64p, _f64p, _f64p]
    lib.esfm_ref_rodrigues_to_vec.restype = None
    lib.esfm_ref_rodrigues_to_vec.argtypes = [_f64p, _f64p]
    lib.esfm_ref_solve_pnp_ransac.restype = C.c_int
    lib.esfm_ref_solve_pnp_ransac.argtypes = [_f32p, _f32p, C.c_int, _f32p, C.c_int, C.c_double, C.c_double, _f64p, _f64p, _f64p, _u8p,
                                              C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.esfm_ref_bgr2gray.restype = None
    lib.esfm_ref_bgr2gray.argtypes = [_u8p, C.c_int, _u8p]
    lib.esfm_ref_undistort.restype = C.c_int
    lib.esfm_ref_undistort.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _f64p, _f64p, _u8p]
    lib.esfm_ref_orb.restype = C.c_int
    lib.esfm_ref_orb.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _u8p]
    lib.esfm_ref_orb_pattern.restype = None
    lib.esfm_ref_orb_pattern.argtypes = [C.POINTER(C.c_int8)]
    lib.esfm_ref_surf.restype = C.c_int
    lib.esfm_ref_surf.argtypes = [_u8p, C.c_int, C.c_int, C.c_double, C.c_int, _f32p, _f32p]
    _LIB, _LIB_PATH = lib, path
    return lib


def usable_cpus() -> int:
    """CPUs this process can actually run on: the affinity mask capped by the cgroup CPU quota (cgroup v2 cpu.max, v1 cfs quota).
    os.cpu_count() alone is the machine's logical CPU count -- 256 on the GPU boxes, whose containers are capped at 16 CPUs' worth of
    time: 256 OpenMP threads there run at a fraction of 16 threads' speed (round 4's "256-core" CPU baseline: 193 pairs/s; 16
    threads: 436)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()[:2]
            quota = None if a == "max" else float(a) / float(b)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except Exception:
            quota = None
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return max(1, n)


def set_num_threads(n: int) -> None:
    """OpenMP threads of the oracle's parallel regions, capped at usable_cpus() (callers pass os.cpu_count())."""
    load().esfm_ref_set_num_threads(max(1, min(int(n), usable_cpus())))


def num_threads() -> int:
    return int(load().esfm_ref_num_threads())


# ----------------------------------------------------------------------------- matching
def knn2_l2(q: np.ndarray, t: np.ndarray):
    q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
    nq, dim = q.shape if q.ndim == 2 else (0, t.shape[1])
    nt = t.shape[0]
    idx = np.empty(2 * nq, np.int32); dist = np.empty(2 * nq, np.float32)
    load().esfm_ref_knn2_l2_f32(q.reshape(-1), nq, t.reshape(-1), nt, dim, idx, dist)
    return idx.reshape(nq, 2), dist.reshape(nq, 2)


def match_pairs_l2(sets, pairs, ratio: float):
    """The pair loop in one call, one parallel region over all (pair, query) items.  Returns [(queryIdx, trainIdx, distance)] per pair."""
    sets = [np.ascontiguousarray(s_, np.float32) for s_ in sets]
    dim = sets[0].shape[1]
    off = np.zeros(len(sets) + 1, np.int32)
    np.cumsum([s_.shape[0] for s_ in sets], out=off[1:])
    bank = np.ascontiguousarray(np.concatenate(sets, axis=0)).reshape(-1)
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    total = int(sum(sets[i].shape[0] for i, _ in pairs))
    qi = np.zeros(max(total, 1), np.int32); ti = np.zeros(max(total, 1), np.int32); d = np.zeros(max(total, 1), np.float32)
    n_out = np.zeros(max(len(pairs), 1), np.int32); out_off = np.zeros(len(pairs) + 1, np.int64)
    rc = load().esfm_ref_match_pairs_l2(bank, off, len(sets), dim, pairs.reshape(-1), len(pairs), float(ratio), qi, ti, d, n_out, out_off)
    if rc != 0:
        raise ValueError("esfm_ref_match_pairs_l2: bad arguments (dim must be a multiple of 8)")
    return [(qi[o:o + k].copy(), ti[o:o + k].copy(), d[o:o + k].copy()) for o, k in zip(out_off[:-1], n_out[:len(pairs)])]


def knn2_l2_scalar(q: np.ndarray, t: np.ndarray):
    """The plain (query, train) loop over esfm_ref_l2sqr: the definition knn2_l2's SIMD body is tested against."""
    q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
    nq, dim = q.shape if q.ndim == 2 else (0, t.shape[1])
    nt = t.shape[0]
    idx = np.empty(2 * nq, np.int32); dist = np.empty(2 * nq, np.float32)
    load().esfm_ref_knn2_l2_f32_scalar(q.reshape(-1), nq, t.reshape(-1), nt, dim, idx, dist)
    return idx.reshape(nq, 2), dist.reshape(nq, 2)


def knn2_hamming(q: np.ndarray, t: np.ndarray):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    nq, nb = q.shape
    nt = t.shape[0]
    idx = np.empty(2 * nq, np.int32); dist = np.empty(2 * nq, np.float32)
    load().esfm_ref_knn2_hamming(q.reshape(-1), nq, t.reshape(-1), nt, nb, idx, dist)
    return idx.reshape(nq, 2), dist.reshape(nq, 2)


def ratio_filter(idx: np.ndarray, dist: np.ndarray, ratio: float):
    nq = idx.shape[0]
    qi = np.empty(max(nq, 1), np.int32); ti = np.empty(max(nq, 1), np.int32); d = np.empty(max(nq, 1), np.float32)
    n = load().esfm_ref_ratio_filter(np.ascontiguousarray(idx, np.int32).reshape(-1),
                                     np.ascontiguousarray(dist, np.float32).reshape(-1), nq, float(ratio), qi, ti, d)
    return qi[:n].copy(), ti[:n].copy(), d[:n].copy()


def match_l2(q, t, ratio=0.5):
    """matchFeaturesSURF with exact brute force (feature_matching.cpp:115-142)."""
    idx, dist = knn2_l2(q, t)
    return ratio_filter(idx, dist, ratio)


def match_hamming(q, t, ratio=0.8):
    """matchFeaturesORB (feature_matching.cpp:71-97)."""
    idx, dist = knn2_hamming(q, t)
    return ratio_filter(idx, dist, ratio)


def l2sqr(a, b) -> float:
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return float(load().esfm_ref_l2sqr(a, b, a.size))


# ----------------------------------------------------------------------------- BA
def ba_default_options() -> BAOptions:
    o = BAOptions()
    load().esfm_ref_ba_options_default(C.byref(o))
    return o


def ba_residual_jac(cam, pt, K4, uv):
    r = np.empty(2); Jc = np.empty(12); Jp = np.empty(6)
    load().esfm_ref_ba_residual_jac(np.ascontiguousarray(cam, np.float64), np.ascontiguousarray(pt, np.float64),
                                    np.ascontiguousarray(K4, np.float32), np.ascontiguousarray(uv, np.float32),
                                    r, Jc, Jp)
    return r, Jc.reshape(2, 6), Jp.reshape(2, 3)


def ba_cost(cam_idx, pt_idx, uv, K4, cams, pts, cauchy_a=0.5) -> float:
    return float(load().esfm_ref_ba_cost(len(cam_idx), np.ascontiguousarray(cam_idx, np.int32),
                                         np.ascontiguousarray(pt_idx, np.int32),
                                         np.ascontiguousarray(uv, np.float32).reshape(-1),
                                         np.ascontiguousarray(K4, np.float32).reshape(-1),
                                         np.ascontiguousarray(cams, np.float64).reshape(-1),
                                         np.ascontiguousarray(pts, np.float64).reshape(-1), float(cauchy_a)))


def ba_solve(cam_idx, pt_idx, uv, K4, cams, pts, options: Optional[BAOptions] = None):
    """ceres::Solve restatement.  Returns (cams, pts, summary); inputs are not modified."""
    cams = np.array(cams, np.float64, copy=True, order="C").reshape(-1, 6)
    pts = np.array(pts, np.float64, copy=True, order="C").reshape(-1, 3)
    summ = BASummary()
    opt = options if options is not None else ba_default_options()
    rc = load().esfm_ref_ba_solve(cams.shape[0], pts.shape[0], len(cam_idx),
                                  np.ascontiguousarray(cam_idx, np.int32), np.ascontiguousarray(pt_idx, np.int32),
                                  np.ascontiguousarray(uv, np.float32).reshape(-1),
                                  np.ascontiguousarray(K4, np.float32).reshape(-1),
                                  cams.reshape(-1), pts.reshape(-1), C.byref(opt), C.byref(summ))
    if rc != 0:
        raise RuntimeError(f"esfm_ref_ba_solve failed with {rc}")
    return cams, pts, summ


def ba_residual_jac_calib(cam, pt, calib, uv):
    """Free-intrinsics functor (ba.h:170-222): r[2], Jc[2,6], Jp[2,3], Jk[2,4]."""
    r = np.empty(2); Jc = np.empty(12); Jp = np.empty(6); Jk = np.empty(8)
    load().esfm_ref_ba_residual_jac_calib(np.ascontiguousarray(cam, np.float64), np.ascontiguousarray(pt, np.float64),
                                          np.ascontiguousarray(calib, np.float64), np.ascontiguousarray(uv, np.float32),
                                          r, Jc, Jp, Jk)
    return r, Jc.reshape(2, 6), Jp.reshape(2, 3), Jk.reshape(2, 4)


def ba_cost_calib(cam_idx, pt_idx, uv, calib, cams, pts, cauchy_a=0.5) -> float:
    return float(load().esfm_ref_ba_cost_calib(len(cam_idx), np.ascontiguousarray(cam_idx, np.int32),
                                               np.ascontiguousarray(pt_idx, np.int32),
                                               np.ascontiguousarray(uv, np.float32).reshape(-1),
                                               np.ascontiguousarray(calib, np.float64).reshape(-1),
                                               np.ascontiguousarray(cams, np.float64).reshape(-1),
                                               np.ascontiguousarray(pts, np.float64).reshape(-1), float(cauchy_a)))


def ba_solve_ex(cam_idx, pt_idx, uv, K4, cams, pts, calib=None, calib_tol: float = 0.0, ref_cam: int = -1,
                ref_threshold: float = 1e-10, options: Optional[BAOptions] = None):
    """solveBA with the reference's optional pieces: free shared intrinsics bounded to +-calib_tol (ba.cpp:167-196)
    and/or a reference camera bounded to +-ref_threshold (ba.cpp:155-162).  Returns (cams, pts, calib, summary)."""
    cams = np.array(cams, np.float64, copy=True, order="C").reshape(-1, 6)
    pts = np.array(pts, np.float64, copy=True, order="C").reshape(-1, 3)
    cal = None if calib is None else np.array(calib, np.float64, copy=True, order="C").reshape(4)
    K = None if K4 is None else np.ascontiguousarray(K4, np.float32).reshape(-1)
    summ = BASummary()
    opt = options if options is not None else ba_default_options()
    rc = load().esfm_ref_ba_solve_ex(cams.shape[0], pts.shape[0], len(cam_idx),
                                     np.ascontiguousarray(cam_idx, np.int32), np.ascontiguousarray(pt_idx, np.int32),
                                     np.ascontiguousarray(uv, np.float32).reshape(-1),
                                     None if K is None else K.ctypes.data_as(C.c_void_p),
                                     cams.reshape(-1), pts.reshape(-1),
                                     None if cal is None else cal.ctypes.data_as(C.c_void_p), float(calib_tol),
                                     int(ref_cam), float(ref_threshold), C.byref(opt), C.byref(summ))
    if rc != 0:
        raise RuntimeError(f"esfm_ref_ba_solve_ex failed with {rc}")
    return cams, pts, cal, summ


def ls_next_step(f0, g0, prev, cur, min_step, max_step) -> float:
    """Next Armijo trial step; prev / cur = (x, value, gradient) or None when invalid."""
    xp, fp, gp = prev if prev is not None else (0.0, 0.0, 0.0)
    xc, fc, gc = cur if cur is not None else (0.0, 0.0, 0.0)
    return float(load().esfm_ref_ls_next_step(f0, g0, xp, fp, gp, int(prev is not None), xc, fc, gc,
                                              int(cur is not None), min_step, max_step))


def ba_partial_reduced(n_cam, n_pt, cam_idx, pt_idx, uv, K4, cams, pts, cauchy_a, radius, diag_c, diag_p,
                       add_cam_diag: bool):
    n = 6 * n_cam
    Sm = np.zeros(n * n); rhs = np.zeros(n)
    rc = load().esfm_ref_ba_partial_reduced(n_cam, n_pt, len(cam_idx), np.ascontiguousarray(cam_idx, np.int32),
                                            np.ascontiguousarray(pt_idx, np.int32),
                                            np.ascontiguousarray(uv, np.float32).reshape(-1),
                                            np.ascontiguousarray(K4, np.float32).reshape(-1),
                                            np.ascontiguousarray(cams, np.float64).reshape(-1),
                                            np.ascontiguousarray(pts, np.float64).reshape(-1),
                                            float(cauchy_a), float(radius),
                                            np.ascontiguousarray(diag_c, np.float64).reshape(-1),
                                            np.ascontiguousarray(diag_p, np.float64).reshape(-1),
                                            int(add_cam_diag), Sm, rhs)
    if rc != 0:
        raise RuntimeError("esfm_ref_ba_partial_reduced failed")
    return Sm.reshape(n, n), rhs


def ba_column_sqnorms(n_cam, n_pt, cam_idx, pt_idx, uv, K4, cams, pts, cauchy_a):
    nc = np.zeros(6 * n_cam); npp = np.zeros(3 * n_pt)
    rc = load().esfm_ref_ba_column_sqnorms(n_cam, n_pt, len(cam_idx), np.ascontiguousarray(cam_idx, np.int32),
                                           np.ascontiguousarray(pt_idx, np.int32),
                                           np.ascontiguousarray(uv, np.float32).reshape(-1),
                                           np.ascontiguousarray(K4, np.float32).reshape(-1),
                                           np.ascontiguousarray(cams, np.float64).reshape(-1),
                                           np.ascontiguousarray(pts, np.float64).reshape(-1), float(cauchy_a), nc, npp)
    if rc != 0:
        raise RuntimeError("esfm_ref_ba_column_sqnorms failed")
    return nc, npp


def iterations(summ: BASummary):
    return [summ.iterations[i] for i in range(min(summ.num_iterations + 1, BA_MAX_LOG))]


# ----------------------------------------------------------------------------- sparse-cloud filter
def sor_filter(points, mean_k: int = 50, std_mul: float = 2.0):
    """pcl::StatisticalOutlierRemoval as CProceesing::SORFilter configures it (cloudprocessing.hpp:24-36).
    points: [n, stride >= 3] float32, xyz first.  Returns (keep mask, mean distances, threshold)."""
    pts = np.ascontiguousarray(points, np.float32)
    n, stride = pts.shape
    md = np.zeros(max(n, 1), np.float32); keep = np.zeros(max(n, 1), np.uint8); thr = C.c_double(0.0)
    load().esfm_ref_sor_filter(pts.reshape(-1), n, stride, int(mean_k), float(std_mul), md, keep, C.byref(thr))
    return keep[:n].astype(bool), md[:n], thr.value


# ----------------------------------------------------------------------------- two-view triangulation
def triangulate_points(P1, P2, pts1, pts2) -> np.ndarray:
    """cv::triangulatePoints as estimate_motion.cpp:263/:333 call it.  P: [3,4] float32; pts: [n,2] float32 normalised
    image points.  Returns the homogeneous points [n,4] float32 (sign of each row arbitrary)."""
    P1 = np.ascontiguousarray(P1, np.float32).reshape(12); P2 = np.ascontiguousarray(P2, np.float32).reshape(12)
    a = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); b = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    n = a.shape[0]
    out = np.zeros(max(4 * n, 1), np.float32)
    load().esfm_ref_triangulate_points(P1, P2, a.reshape(-1), b.reshape(-1), n, out)
    return out[:4 * n].reshape(n, 4)


# ----------------------------------------------------------------------------- essential matrix RANSAC + pose
def ransac_samples(count: int, n_samples: int) -> np.ndarray:
    """The 5-index samples RANSACPointSetRegistrator::getSubset draws from cv::RNG((uint64)-1)."""
    idx = np.zeros(5 * max(n_samples, 1), np.int32)
    load().esfm_ref_ransac_samples(int(count), int(n_samples), idx)
    return idx[:5 * n_samples].reshape(n_samples, 5)


def five_point(q1, q2) -> np.ndarray:
    """EMEstimatorCallback::runKernel on 5 normalised correspondences: [k, 3, 3] essential matrices (k <= 10)."""
    a = np.ascontiguousarray(q1, np.float64).reshape(10); b = np.ascontiguousarray(q2, np.float64).reshape(10)
    out = np.zeros(90, np.float64)
    k = load().esfm_ref_five_point(a, b, out)
    return out[:9 * k].reshape(k, 3, 3)


def five_point_stages(q1, q2):
    """The restatement's intermediate values for one sample (esfm_ref_five_point_stages): 117 doubles in the layout of
    include/esfm.h's esfm_five_point_models (the last one = sweeps of the root iteration, -1: no polynomial)."""
    a = np.ascontiguousarray(q1, np.float64).reshape(10); b = np.ascontiguousarray(q2, np.float64).reshape(10)
    out = np.zeros(117, np.float64)
    out[116] = load().esfm_ref_five_point_stages(a, b, out[:116])
    return out


def find_essential_ransac(pts1, pts2, K4, prob: float = 0.99, threshold: float = 1.0):
    """cv::findEssentialMat(pts1, pts2, K, RANSAC, prob, threshold, mask).  Returns (ok, E [3,3], mask [n] bool,
    iterations run, inlier count)."""
    a = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); b = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    n = a.shape[0]
    E = np.zeros(9, np.float64); mask = np.zeros(max(n, 1), np.uint8)
    it = C.c_int32(0); cnt = C.c_int32(0)
    ok = load().esfm_ref_find_essential_ransac(a.reshape(-1), b.reshape(-1), n, np.ascontiguousarray(K4, np.float32).reshape(4),
                                               float(prob), float(threshold), E, mask, C.byref(it), C.byref(cnt))
    return bool(ok), E.reshape(3, 3), mask[:n].astype(bool), it.value, cnt.value


def recover_pose(E, pts1, pts2, K4, mask=None):
    """cv::recoverPose(E, pts1, pts2, K, R, t, mask).  Returns (good, R [3,3], t [3], mask)."""
    a = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); b = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    n = a.shape[0]
    m = np.ones(max(n, 1), np.uint8) if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8)).copy()
    if len(m) < max(n, 1):
        m = np.ones(max(n, 1), np.uint8)
    R = np.zeros(9, np.float64); t = np.zeros(3, np.float64)
    g = load().esfm_ref_recover_pose(np.ascontiguousarray(E, np.float64).reshape(9), a.reshape(-1), b.reshape(-1), n,
                                     np.ascontiguousarray(K4, np.float32).reshape(4), R, t, m)
    return int(g), R.reshape(3, 3), t, m[:n].astype(bool)


# ----------------------------------------------------------------------------- PnP
def epnp(pts3d, pix, K4):
    """epnp::compute_pose: pts3d [n,3], pix [n,2] (pixels), K4 = fx, cx, fy, cy -> (R [3,3], t [3], mean reprojection error)."""
    a = np.ascontiguousarray(pts3d, np.float64).reshape(-1, 3); b = np.ascontiguousarray(pix, np.float64).reshape(-1, 2)
    R = np.zeros(9); t = np.zeros(3)
    e = load().esfm_ref_epnp(a.reshape(-1), b.reshape(-1), a.shape[0], np.ascontiguousarray(K4, np.float64).reshape(4), R, t)
    return R.reshape(3, 3), t, float(e)


def solve_pnp_ransac(pts3d, pix, K4, iterations_count: int = 100, reproj_error: float = 8.0, confidence: float = 0.99):
    """cv::solvePnPRansac(pts3d, pix, K, 0, rvec, tvec, false, iterationsCount, reprojectionError, confidence, inliers,
    SOLVEPNP_EPNP) -> (ok, R, t, rvec, mask [n] bool, iterations run)."""
    a = np.ascontiguousarray(pts3d, np.float32).reshape(-1, 3); b = np.ascontiguousarray(pix, np.float32).reshape(-1, 2)
    n = a.shape[0]
    R = np.zeros(9); t = np.zeros(3); rv = np.zeros(3); mask = np.zeros(max(n, 1), np.uint8)
    it = C.c_int32(0); ni = C.c_int32(0)
    ok = load().esfm_ref_solve_pnp_ransac(a.reshape(-1), b.reshape(-1), n, np.ascontiguousarray(K4, np.float32).reshape(4), int(iterations_count),
                                          float(reproj_error), float(confidence), R, t, rv, mask, C.byref(it), C.byref(ni))
    return bool(ok), R.reshape(3, 3), t, rv, mask[:n].astype(bool), it.value


# ----------------------------------------------------------------------------- SURF
def bgr2gray(bgr) -> np.ndarray:
    b = np.ascontiguousarray(bgr, np.uint8)
    out = np.zeros(b.shape[:2], np.uint8)
    load().esfm_ref_bgr2gray(b.reshape(-1), out.size, out.reshape(-1))
    return out


def surf(gray, hessian_threshold: float = 100.0, max_kp: int = 200000):
    """SURF::detect + SURF::compute (feature_matching.cpp:43-58) on a gray uint8 image.
    Returns (keypoints [n, 7] float32: x, y, size, angle, response, octave, class_id; descriptors [n, 64] float32)."""
    g = np.ascontiguousarray(gray, np.uint8)
    kp = np.zeros((max_kp, 7), np.float32); desc = np.zeros((max_kp, 64), np.float32)
    n = load().esfm_ref_surf(g.reshape(-1), g.shape[0], g.shape[1], float(hessian_threshold), int(max_kp), kp.reshape(-1), desc.reshape(-1))
    return kp[:n].copy(), desc[:n].copy()


def orb(gray, nfeatures: int = 500, max_kp: int = 200000):
    """cv::ORB::create(nfeatures) detect + compute on a gray image: (keypoints[n, 7], descriptors[n, 32] uint8)."""
    g = np.ascontiguousarray(gray, np.uint8)
    cap = min(int(max_kp), 2 * int(nfeatures) + 4096)
    kp = np.zeros((cap, 7), np.float32); desc = np.zeros((cap, 32), np.uint8)
    n = load().esfm_ref_orb(g.reshape(-1), g.shape[0], g.shape[1], int(nfeatures), cap, kp.reshape(-1), desc.reshape(-1))
    return kp[:n].copy(), desc[:n].copy()


def orb_pattern() -> np.ndarray:
    """The 256 test point pairs (x0, y0, x1, y1) of the descriptor."""
    out = np.zeros(1024, np.int8)
    load().esfm_ref_orb_pattern(out.ctypes.data_as(C.POINTER(C.c_int8)))
    return out.reshape(256, 4)


def undistort(image, K4, dist4):
    """cv::undistort(image, out, K, dist) as MotionEstimator::doUnDistort calls it (estimate_motion.cpp:431-441).
    image: [rows, cols] or [rows, cols, 3] uint8; K4 = fx, cx, fy, cy; dist4 = k1, k2, p1, p2 (doubles)."""
    img = np.ascontiguousarray(image, np.uint8)
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty_like(img)
    k = np.ascontiguousarray(K4, np.float64).reshape(4)
    d = np.ascontiguousarray(dist4, np.float64).reshape(4)
    rc = load().esfm_ref_undistort(img.reshape(-1), img.shape[0], img.shape[1], ch, k, d, out.reshape(-1))
    if rc != 0:
        raise MemoryError("esfm_ref_undistort")
    return out

"""ctypes binding of libesfm_hip.so (include/esfm.h).

The shared library is the product: HIP kernels for gfx950 behind a C ABI.  This
module only loads it and declares signatures.  There is no CPU fallback: if the
library is missing, or no MI355X is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# ESFM_LIB: developer override to A/B an experimental build of the same ABI (never a CPU library)
LIB_PATH = os.environ.get("ESFM_LIB") or os.path.join(_HERE, "libesfm_hip.so")

ESFM_OK = 0
ESFM_L2_F32 = 0
ESFM_HAMMING = 1
ESFM_REDUCE_SUM = 0
ESFM_REDUCE_MAX = 1
K_L2_KNN, K_HAMMING_KNN, K_BA_LINEARIZE, K_BA_SCHUR, K_BA_SOLVE, K_L2_RESCAN, K_SOR_KNN, K_TRIANGULATE, K_RANSAC, K_SURF_DET, K_SURF_DESC, K_UNDISTORT, K_ORB_FAST, K_L2_SECOND = range(14)
BA_MAX_LOG = 256

STATUS_NAMES = {
    0: "ESFM_OK", -1: "ESFM_ERR_INVALID_ARG", -2: "ESFM_ERR_NO_DEVICE", -3: "ESFM_ERR_HIP", -4: "ESFM_ERR_OOM",
    -5: "ESFM_ERR_UNSUPPORTED", -6: "ESFM_ERR_NUMERIC", -7: "ESFM_ERR_COMM", -8: "ESFM_ERR_STALE_PREPARED",
}

# every symbol include/esfm.h declares (tests/test_abi.py checks the .so exports all of them)
EXPORTED_SYMBOLS = [
    "esfm_version", "esfm_last_error", "esfm_device_count", "esfm_ctx_create", "esfm_ctx_destroy",
    "esfm_ctx_synchronize", "esfm_ctx_stream", "esfm_ctx_set_kernel_timing", "esfm_ctx_kernel_time",
    "esfm_knn2_l2_f32", "esfm_knn2_hamming", "esfm_match_l2_f32", "esfm_match_hamming",
    "esfm_match_pairs_dev", "esfm_match_pairs", "esfm_knn2_pairs_dev", "esfm_knn2_pairs_screened_dev", "esfm_match_prepare_dev", "esfm_match_debug_counters", "esfm_match_release_prepared", "esfm_match_release_prepared_buffer", "esfm_match_prepared_buffer", "esfm_ctx_set_prepared_check", "esfm_match_last_stats", "esfm_match_last_second_pass", "esfm_ctx_set_l2_audit", "esfm_match_last_flagged",
    "esfm_shard_pair_list",
    "esfm_comm_get_unique_id", "esfm_comm_create", "esfm_comm_destroy", "esfm_comm_rank", "esfm_comm_world", "esfm_comm_rccl_ranks", "esfm_comm_allreduce",
    "esfm_ba_options_default", "esfm_ba_solve", "esfm_ba_problem_create", "esfm_ba_problem_set_params",
    "esfm_ba_problem_solve", "esfm_ba_problem_get_params", "esfm_ba_problem_destroy", "esfm_ba_problem_cost",
    "esfm_ba_shard_points", "esfm_ba_reduced_plan", "esfm_ba_problem_create_free_calib", "esfm_ba_problem_set_calib", "esfm_ba_problem_get_calib",
    "esfm_ba_problem_fix_camera", "esfm_ba_solve_ex", "esfm_ba_line_search_next_step",
    "esfm_sor_filter", "esfm_sor_mean_distances_dev", "esfm_triangulate_points", "esfm_triangulate_pairs",
    "esfm_find_essential_mat", "esfm_find_essential_pairs", "esfm_recover_pose", "esfm_recover_pose_pairs", "esfm_ransac_sample_stream", "esfm_five_point_models", "esfm_five_point_models_host",
    "esfm_solve_pnp_ransac", "esfm_surf_detect_and_compute", "esfm_orb_detect_and_compute", "esfm_undistort",
]


class EsfmError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class BAOptions(C.Structure):
    """esfm_ba_options (include/esfm.h)."""
    _fields_ = [
        ("max_num_iterations", C.c_int32), ("jacobi_scaling", C.c_int32),
        ("max_num_consecutive_invalid_steps", C.c_int32), ("verbose", C.c_int32),
        ("cauchy_a", C.c_double), ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
    ]


class BAIteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32), ("line_search_steps", C.c_int32),
        ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double), ("step_norm", C.c_double),
        ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double), ("model_cost_change", C.c_double),
    ]


class BASummary(C.Structure):
    _fields_ = [
        ("termination", C.c_int32), ("num_iterations", C.c_int32), ("num_successful_steps", C.c_int32),
        ("num_unsuccessful_steps", C.c_int32), ("num_active_cameras", C.c_int32), ("num_active_points", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("solve_seconds", C.c_double),
        ("iterations", BAIteration * BA_MAX_LOG),
    ]

    def log(self):
        return [self.iterations[i] for i in range(min(self.num_iterations + 1, BA_MAX_LOG))]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libesfm_hip.so (built by __graft_entry__.build() / easysfm_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C easysfm_amd/csrc`). "
            "easysfm_amd has no CPU fallback.")
    # Load order matters: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.  If the
    # system copy under /opt/rocm were mapped first (through this library's DT_NEEDED) and torch's
    # afterwards, the process would hold two HIP runtimes and the second one finds no device.
    # Importing torch first makes both resolve to the single copy torch ships.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing (device memory, streams, torch.distributed), not a requirement
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32p, f32p, f64p, i64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int64)
    L.esfm_version.restype = C.c_char_p
    L.esfm_last_error.restype = C.c_char_p
    L.esfm_device_count.restype = C.c_int
    L.esfm_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.esfm_ctx_destroy.argtypes = [vp]
    L.esfm_ctx_synchronize.argtypes = [vp]
    L.esfm_ctx_stream.argtypes = [vp]
    L.esfm_ctx_stream.restype = vp
    L.esfm_ctx_set_kernel_timing.argtypes = [vp, C.c_int]
    L.esfm_ctx_kernel_time.argtypes = [vp, C.c_int, f64p, i64p]
    L.esfm_knn2_l2_f32.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    L.esfm_knn2_hamming.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    L.esfm_match_l2_f32.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_double, vp, vp, vp, i32p]
    L.esfm_match_hamming.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_double, vp, vp, vp, i32p]
    L.esfm_match_pairs_dev.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_double, vp, vp, vp, vp, vp]
    L.esfm_knn2_pairs_dev.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp]
    L.esfm_knn2_pairs_screened_dev.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_double, vp, vp, vp]
    L.esfm_match_debug_counters.argtypes = [vp, vp]
    L.esfm_match_pairs.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_double, vp, vp, vp, vp, vp]
    L.esfm_match_prepare_dev.argtypes = [vp, C.c_int, vp, C.c_int64, C.c_int]
    L.esfm_match_release_prepared.argtypes = [vp]
    L.esfm_match_release_prepared_buffer.argtypes = [vp, vp]
    L.esfm_match_prepared_buffer.argtypes = [vp, vp]
    L.esfm_ctx_set_prepared_check.argtypes = [vp, C.c_int]
    L.esfm_match_last_stats.argtypes = [vp, i64p, i64p]
    L.esfm_match_last_second_pass.argtypes = [vp, i64p]
    L.esfm_ctx_set_l2_audit.argtypes = [vp, C.c_int]
    L.esfm_match_last_flagged.argtypes = [vp, vp, C.c_int64, i64p]
    L.esfm_shard_pair_list.argtypes = [C.c_int, vp, C.c_int, C.c_int, vp]
    L.esfm_comm_get_unique_id.argtypes = [vp]
    L.esfm_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.esfm_comm_destroy.argtypes = [vp]
    L.esfm_comm_rank.argtypes = [vp]
    L.esfm_comm_world.argtypes = [vp]
    L.esfm_comm_rccl_ranks.argtypes = [vp]
    L.esfm_comm_allreduce.argtypes = [vp, vp, C.c_int64, C.c_int, vp]
    L.esfm_ba_options_default.argtypes = [C.POINTER(BAOptions)]
    L.esfm_ba_options_default.restype = None
    L.esfm_ba_solve.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(BAOptions), ALLREDUCE_FN, vp,
                                C.POINTER(BASummary)]
    L.esfm_ba_problem_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.esfm_ba_problem_create_free_calib.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_double, vp, vp, C.POINTER(C.c_void_p)]
    L.esfm_ba_problem_set_calib.argtypes = [C.c_void_p, vp, C.c_double]
    L.esfm_ba_problem_get_calib.argtypes = [C.c_void_p, vp]
    L.esfm_ba_problem_fix_camera.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.esfm_ba_solve_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_double, C.c_int, C.c_double,
                                   C.POINTER(BAOptions), ALLREDUCE_FN, vp, C.POINTER(BASummary)]
    L.esfm_ba_line_search_next_step.restype = C.c_double
    L.esfm_ba_line_search_next_step.argtypes = [C.c_double] * 5 + [C.c_int] + [C.c_double] * 3 + [C.c_int]
    L.esfm_sor_filter.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, vp, vp, i32p, f64p]
    L.esfm_sor_mean_distances_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.esfm_triangulate_points.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp]
    L.esfm_triangulate_pairs.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.esfm_find_essential_mat.argtypes = [vp, vp, vp, C.c_int, vp, C.c_double, C.c_double, vp, vp, i32p]
    L.esfm_find_essential_pairs.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp, vp]
    L.esfm_recover_pose.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, i32p]
    L.esfm_recover_pose_pairs.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.esfm_ransac_sample_stream.argtypes = [C.c_int, C.c_int, vp]
    L.esfm_five_point_models.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp]
    L.esfm_five_point_models_host.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.esfm_surf_detect_and_compute.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, i32p]
    L.esfm_orb_detect_and_compute.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, i32p]
    L.esfm_undistort.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.esfm_solve_pnp_ransac.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_double, C.c_double, vp, vp, vp, vp, i32p, i32p]
    L.esfm_ba_problem_set_params.argtypes = [vp, vp, vp]
    L.esfm_ba_problem_solve.argtypes = [vp, C.POINTER(BAOptions), ALLREDUCE_FN, vp, C.POINTER(BASummary)]
    L.esfm_ba_problem_get_params.argtypes = [vp, vp, vp]
    L.esfm_ba_problem_destroy.argtypes = [vp]
    L.esfm_ba_problem_cost.argtypes = [vp, C.c_double, f64p]
    L.esfm_ba_shard_points.argtypes = [C.c_int, C.c_int, vp, C.c_int, vp]
    L.esfm_ba_reduced_plan.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("esfm_device_count",):
            pass
    _lib = L
    return L


def check(status: int) -> None:
    if status != ESFM_OK:
        raise EsfmError(status, lib().esfm_last_error().decode("utf-8", "replace"))


class Context:
    """esfm_ctx: one per host thread and GPU.  `stream` is a hipStream_t handle (int), e.g.
    torch.cuda.current_stream().cuda_stream; None lets the library own a stream."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        L = lib()
        self._h = C.c_void_p()
        self.torch_stream = None
        check(L.esfm_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = int(device)

    @classmethod
    def on_torch_stream(cls, device: int = 0) -> "Context":
        """Context bound to a dedicated torch.cuda.Stream (kept alive in .torch_stream), so torch ops
        and collectives issued under ``with torch.cuda.stream(ctx.torch_stream)`` are ordered with the
        library's kernels.  (torch's default stream has handle 0, which the C ABI reads as "create
        your own stream", hence the dedicated one.)"""
        import torch
        ts = torch.cuda.Stream(device=device)
        ctx = cls(device, ts.cuda_stream)
        ctx.torch_stream = ts
        return ctx

    @property
    def handle(self) -> C.c_void_p:
        if not self._h:
            raise RuntimeError("context destroyed")
        return self._h

    @property
    def stream(self) -> int:
        return int(lib().esfm_ctx_stream(self.handle) or 0)

    def synchronize(self) -> None:
        check(lib().esfm_ctx_synchronize(self.handle))

    def set_kernel_timing(self, enable: bool) -> None:
        check(lib().esfm_ctx_set_kernel_timing(self.handle, 1 if enable else 0))

    def kernel_time(self, kernel_id: int):
        """(total_ms, launches) of the hipEvent-bracketed launches of `kernel_id` since the last call."""
        ms = C.c_double(0.0); n = C.c_int64(0)
        check(lib().esfm_ctx_kernel_time(self.handle, int(kernel_id), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self) -> None:
        if self._h:
            lib().esfm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0, None)
    return _default_ctx

"""Parity of the bounds-constrained bundle adjustment against the CPU oracle and the
committed golden traces: free shared intrinsics within +-tolerance (reference ba.cpp:167-196, functor ba.h:170-222) and
the reference camera held at the origin by +-1e-10 bounds (ba.cpp:134, :155-162), both through Ceres' constrained
trust-region loop (projection, projected gradient norm, Armijo line search).

Tolerances as tests/test_ba_gpu.py: cost trace 1e-9 relative; radius 1e-6; step norm / model change / gradient 1e-3
(gauge-dependent while no camera is fixed); parameters after a few iterations 1e-6 + 1e-6; line-search contraction
counts and the accept/reject pattern exactly."""
import os

import numpy as np
import pytest

import easysfm_amd as E
from easysfm_amd import synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
K0 = np.array(synth.FOUNTAIN_K4, np.float64)


def _opts(oracle, max_iter, **kw):
    opt = E.default_options(); opt.max_num_iterations = max_iter
    ropt = oracle.ba_default_options(); ropt.max_num_iterations = max_iter
    for k, v in kw.items():
        setattr(opt, k, v); setattr(ropt, k, v)
    return opt, ropt


def _compare(summ, rs, oracle, step_tol=1e-3):
    a_log, b_log = summ.log(), oracle.iterations(rs)
    assert summ.termination == rs.termination and summ.num_iterations == rs.num_iterations
    assert [a.step_is_successful for a in a_log] == [b.step_is_successful for b in b_log]
    assert [a.line_search_steps for a in a_log] == [b.line_search_steps for b in b_log]
    for a, b in zip(a_log, b_log):
        assert a.step_is_valid == b.step_is_valid, a.iteration
        assert abs(a.cost - b.cost) <= 1e-9 * max(abs(b.cost), 1.0), (a.iteration, a.cost, b.cost)
        assert abs(a.trust_region_radius - b.trust_region_radius) <= 1e-6 * b.trust_region_radius, a.iteration
        for f in ("step_norm", "model_cost_change"):
            x, y = getattr(a, f), getattr(b, f)
            assert abs(x - y) <= step_tol * max(abs(y), 1e-6), (a.iteration, f, x, y)
        assert abs(a.gradient_max_norm - b.gradient_max_norm) <= step_tol * max(1.0, abs(b.gradient_max_norm)), a.iteration


@pytest.mark.parametrize("n_cam,n_pt,k,seed,tol", [(4, 60, 3, 41, 100.0), (6, 300, 4, 5, 20.0), (25, 2000, 8, 3, 15.0)])
def test_free_calib_trace_and_params_match_oracle(gpu_ctx, oracle_lib, n_cam, n_pt, k, seed, tol):
    sc = synth.ba_scene(n_cam, n_pt, k, seed=seed)
    calib0 = K0 * np.array([1.02, 0.99, 0.98, 1.01])
    opt, ropt = _opts(oracle_lib, 6)
    cams, pts, cal, summ = E.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, None, sc.cams0, sc.pts0, calib=calib0, calib_tol=tol,
                                         options=opt, ctx=gpu_ctx)
    rc, rp, rk, rs = oracle_lib.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, None, sc.cams0, sc.pts0, calib=calib0, calib_tol=tol,
                                            options=ropt)
    _compare(summ, rs, oracle_lib)
    assert np.allclose(cal, rk, rtol=1e-7, atol=1e-5)
    assert np.allclose(cams, rc, rtol=1e-6, atol=1e-6) and np.allclose(pts, rp, rtol=1e-6, atol=1e-6)
    assert np.all(cal >= calib0 - tol) and np.all(cal <= calib0 + tol)
    assert summ.final_cost < summ.initial_cost
    # the returned intrinsics are the ones the final cost was evaluated with
    c = oracle_lib.ba_cost_calib(sc.cam_idx, sc.pt_idx, sc.uv, cal, cams, pts)
    assert abs(c - summ.final_cost) <= 1e-10 * summ.final_cost
    # long run: gauge-invariant end results only (weak 3-view geometries leave fy / cy poorly determined)
    opt, ropt = _opts(oracle_lib, 50)
    cams, pts, cal, summ = E.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, None, sc.cams0, sc.pts0, calib=calib0, calib_tol=tol,
                                         options=opt, ctx=gpu_ctx)
    rc, rp, rk, rs = oracle_lib.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, None, sc.cams0, sc.pts0, calib=calib0, calib_tol=tol,
                                            options=ropt)
    assert abs(summ.final_cost - rs.final_cost) <= 1e-4 * rs.final_cost
    assert np.all(cal >= calib0 - tol) and np.all(cal <= calib0 + tol)


def test_free_calib_active_bound_and_line_search(gpu_ctx, oracle_lib):
    """Tight box: fx / fy run into their bounds, the projected step fails the Armijo test and the search contracts."""
    sc = synth.ba_scene(4, 60, 3, seed=41)
    calib0 = K0 * np.array([1.03, 0.99, 0.97, 1.01])
    opt, ropt = _opts(oracle_lib, 10)
    cams, pts, cal, summ = E.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, None, sc.cams0, sc.pts0, calib=calib0, calib_tol=6.0,
                                         options=opt, ctx=gpu_ctx)
    rc, rp, rk, rs = oracle_lib.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, None, sc.cams0, sc.pts0, calib=calib0, calib_tol=6.0,
                                            options=ropt)
    _compare(summ, rs, oracle_lib)
    assert sum(a.line_search_steps for a in summ.log()) > 0
    assert np.any(np.abs(np.abs(cal - calib0) - 6.0) < 1e-12)      # a bound is active, exactly
    assert np.allclose(cal, rk, rtol=1e-7, atol=1e-5)


@pytest.mark.parametrize("n_cam,n_pt,k,seed", [(4, 60, 3, 43), (12, 800, 5, 7)])
def test_reference_camera_is_held(gpu_ctx, oracle_lib, n_cam, n_pt, k, seed):
    """ba.cpp:155-162: the reference frame's pose is bounded to +-1e-10; with the gauge (almost) fixed the parameters of a
    longer run are comparable too."""
    sc = synth.in_reference_frame(synth.ba_scene(n_cam, n_pt, k, seed=seed), 0)
    opt, ropt = _opts(oracle_lib, 12)
    cams, pts, cal, summ = E.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ref_cam=0, options=opt, ctx=gpu_ctx)
    rc, rp, _, rs = oracle_lib.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ref_cam=0, options=ropt)
    assert cal is None
    _compare(summ, rs, oracle_lib)
    assert np.all(np.abs(cams[0]) <= 1e-10)
    assert np.allclose(cams, rc, rtol=1e-5, atol=1e-5) and np.allclose(pts, rp, rtol=1e-5, atol=1e-5)
    assert summ.final_cost < 0.5 * summ.initial_cost


def test_reference_camera_away_from_origin_is_projected(gpu_ctx, oracle_lib):
    """The bound is on the parameter VALUES (ba.cpp:159-160), so a reference camera that does not start at the origin is
    projected there at iteration 0 (TrustRegionMinimizer::IterationZero) -- same on both sides."""
    sc = synth.ba_scene(4, 60, 3, seed=45)
    opt, ropt = _opts(oracle_lib, 3)
    cams, pts, _, summ = E.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ref_cam=1, options=opt, ctx=gpu_ctx)
    rc, rp, _, rs = oracle_lib.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ref_cam=1, options=ropt)
    assert np.all(np.abs(cams[1]) <= 1e-10)
    assert abs(summ.initial_cost - rs.initial_cost) <= 1e-9 * rs.initial_cost
    assert [a.step_is_successful for a in summ.log()] == [b.step_is_successful for b in oracle_lib.iterations(rs)]


def test_both_bounds_squared_loss(gpu_ctx, oracle_lib):
    sc = synth.in_reference_frame(synth.ba_scene(5, 80, 4, seed=44, outlier_frac=0.0), 0)
    calib0 = K0 * np.array([0.98, 1.01, 1.02, 0.99])
    opt, ropt = _opts(oracle_lib, 8, cauchy_a=-1.0)
    cams, pts, cal, summ = E.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, None, sc.cams0, sc.pts0, calib=calib0, calib_tol=10.0,
                                         ref_cam=0, options=opt, ctx=gpu_ctx)
    rc, rp, rk, rs = oracle_lib.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, None, sc.cams0, sc.pts0, calib=calib0, calib_tol=10.0,
                                            ref_cam=0, options=ropt)
    _compare(summ, rs, oracle_lib)
    assert np.allclose(cal, rk, rtol=1e-7, atol=1e-5) and np.all(np.abs(cams[0]) <= 1e-10)


@pytest.mark.parametrize("tag", ["calib_loose", "calib_tight", "refcam", "both_squared"])
def test_constrained_golden_traces(gpu_ctx, tag):
    """The HIP path against the committed dense-LM traces directly (no oracle in the loop)."""
    z = np.load(os.path.join(GOLD, "ba_lm_constrained.npz"))
    n = len(z[f"{tag}.cost"]) - 1
    opt = E.default_options(); opt.max_num_iterations = n; opt.cauchy_a = float(z[f"{tag}.cauchy_a"])
    calib0 = z[f"{tag}.calib0"]
    free = calib0.size == 4
    cams, pts, cal, summ = E.ba_solve_ex(z[f"{tag}.cam_idx"], z[f"{tag}.pt_idx"], z[f"{tag}.uv"], None if free else z[f"{tag}.K4"],
                                         z[f"{tag}.cams0"], z[f"{tag}.pts0"], calib=calib0 if free else None,
                                         calib_tol=float(z[f"{tag}.calib_tol"]), ref_cam=int(z[f"{tag}.ref_cam"]), options=opt,
                                         ctx=gpu_ctx)
    log = summ.log()
    assert len(log) == n + 1
    assert [it.step_is_successful for it in log] == z[f"{tag}.ok"].tolist()
    assert [it.line_search_steps for it in log] == z[f"{tag}.ls"].tolist()
    for it, c, rad, sn in zip(log, z[f"{tag}.cost"], z[f"{tag}.radius"], z[f"{tag}.step_norm"]):
        assert abs(it.cost - c) <= (1e-8 if it.step_is_successful else 1e-6) * abs(c), (it.iteration, it.cost, c)
        assert abs(it.trust_region_radius - rad) <= 1e-6 * rad, it.iteration
        assert abs(it.step_norm - sn) <= 1e-3 * max(sn, 1e-6), it.iteration
    x = np.concatenate([cams.ravel(), pts.ravel()] + ([cal] if free else []))
    assert np.allclose(x, z[f"{tag}.x_final"], rtol=1e-4, atol=1e-5)


def test_free_calib_large_path(gpu_ctx, oracle_lib):
    """64 cameras: the reduced system (6 * 65 unknowns) is beyond the single-workgroup LDS solve, so the windowed Schur
    kernel, the intrinsics block row through global atomics and the multi-workgroup Cholesky carry the free intrinsics."""
    sc = synth.ba_scene(64, 3000, 5, seed=21)
    calib0 = K0 * np.array([1.01, 1.0, 0.99, 1.0])
    opt, ropt = _opts(oracle_lib, 4)
    cams, pts, cal, summ = E.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, None, sc.cams0, sc.pts0, calib=calib0, calib_tol=30.0,
                                         options=opt, ctx=gpu_ctx)
    rc, rp, rk, rs = oracle_lib.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, None, sc.cams0, sc.pts0, calib=calib0, calib_tol=30.0,
                                            options=ropt)
    _compare(summ, rs, oracle_lib)
    assert np.allclose(cal, rk, rtol=1e-7, atol=1e-5)


def test_mirror_doSFMBA_free_calib_and_reference_frame(gpu_ctx, oracle_lib):
    """doSFMBA(frames, ..., fix_calib_tolerance_BA, reference_frame_id) (ba.cpp:214-288): intrinsics written back into
    every registered frame's K_cam (:250-256), the reference frame's pose unchanged."""
    sc = synth.in_reference_frame(synth.ba_scene(5, 150, 4, seed=51), 0)
    frames, process = [], []
    K = np.array([[K0[0] * 1.02, 0, K0[1]], [0, K0[2] * 0.98, K0[3]], [0, 0, 1]], np.float32)
    for c in range(sc.n_cam):
        sel = np.nonzero(sc.cam_idx == c)[0]
        fr = E.Frame(frame_id=c, keypoints=sc.uv[sel])
        fr.unique_pixel_ids = sc.pt_idx[sel].astype(np.int64)
        fr.unique_pixel_has_match = np.ones(len(sel), bool)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = synth.aa_to_R(sc.cams0[c, :3]).astype(np.float32); pose[:3, 3] = sc.cams0[c, 3:].astype(np.float32)
        fr.pose_cam = pose; fr.K_cam = K.copy()
        frames.append(fr); process.append(False)
    cloud = E.SparsePointCloud(xyz=sc.pts0.astype(np.float32), unique_point_ids=np.arange(sc.n_pt))
    ba = E.BundleAdjustment(gpu_ctx)
    pose0 = frames[0].pose_cam.copy()
    ba.doSFMBA(frames, process, cloud, 25.0, 0)
    assert ba.ref_process_camera_id_ == 0
    assert ba.summary.final_cost < ba.summary.initial_cost
    assert np.allclose(frames[0].pose_cam, pose0, atol=1e-6)
    Kn = frames[3].K_cam
    assert all(np.array_equal(fr.K_cam, Kn) for fr in frames)
    assert abs(Kn[0, 0] - K0[0]) < 0.2 * abs(K[0, 0] - K0[0])      # fx was 2 % off; the flat camera ring barely constrains fy
    assert np.all(np.abs(np.array([Kn[0, 0], Kn[0, 2], Kn[1, 1], Kn[1, 2]]) - np.array([K[0, 0], K[0, 2], K[1, 1], K[1, 2]])) <= 25.0 + 1e-3)
    # same call through the oracle from the same packed parameters
    ba2 = E.BundleAdjustment.__new__(E.BundleAdjustment); ba2._ctx = None; ba2.options = None; ba2.initBA()
    for fr in frames:
        fr.K_cam = K.copy()
    # (poses / points were updated in place above; only the packing is checked here)
    ba2.setBAProblem(frames, process, cloud, 25.0, 0)
    assert ba2.num_parameters_ == 6 * 5 + 3 * sc.n_pt + 4
    assert np.allclose(ba2.parameters_[-4:], [K[0, 0], K[0, 2], K[1, 1], K[1, 2]])

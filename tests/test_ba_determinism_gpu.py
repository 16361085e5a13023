"""Run-to-run reproducibility of the HIP bundle adjustment.

No reduction of the solver depends on the order in which workgroups or waves arrive:
  * per-camera sums F'F / F'r: gathered per camera in observation order (ba_camacc_chunk_kernel), chunk sums added in order;
  * scalar sums (cost, model cost change, norms): per-workgroup partials added in block order by the last workgroup;
  * the Schur complement: addends rounded to a common power-of-two grid whose partial sums fit 53 bits, so the f64 atomics and
    the slab / all-reduce sums are exact (BADev::quant in easysfm_amd/csrc/ba_kernels.hpp);
  * max-reductions are order-independent; the Cholesky kernels have a fixed schedule.
Two solves of the same problem must therefore agree in every bit of the parameters and of the iteration log.
"""
import numpy as np
import pytest

import easysfm_amd as E
from easysfm_amd import synth

pytestmark = pytest.mark.gpu


def _log(summ):
    return [(it.cost, it.model_cost_change, it.step_norm, it.trust_region_radius, it.gradient_max_norm, it.step_is_successful)
            for it in summ.log()]


def _twice(sc, ctx, runs=3, **kw):
    out = []
    for _ in range(runs):
        opt = E.default_options()
        for k, v in kw.items():
            setattr(opt, k, v)
        out.append(E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt, ctx))
    c0, p0, s0 = out[0]
    for c, p, s in out[1:]:
        assert _log(s) == _log(s0)
        assert np.array_equal(c, c0) and np.array_equal(p, p0)
    return out[0]


@pytest.mark.parametrize("n_cam,n_pt,k,seed,iters", [
    (4, 50, 3, 1, 50),          # the scene whose long run used to stop an iteration apart about one run in twenty
    (25, 7776, 8, 3, 25),       # BA-25, the metric configuration: Schur complement in LDS slabs, one-workgroup Cholesky
    (40, 4000, 6, 5, 10),       # LDS Schur slabs, LDS Cholesky of the round-1 kernel
    (96, 9000, 6, 7, 8),        # windowed Schur (LDS window + global f64 atomics), tiled large Cholesky
])
def test_two_solves_are_bit_identical(gpu_ctx, n_cam, n_pt, k, seed, iters):
    _twice(synth.ba_scene(n_cam, n_pt, k, seed=seed), gpu_ctx, max_num_iterations=iters)


def test_squared_loss_and_no_jacobi_scaling_bit_identical(gpu_ctx):
    sc = synth.ba_scene(12, 1500, 5, seed=11, outlier_frac=0.0)
    _twice(sc, gpu_ctx, max_num_iterations=12, cauchy_a=-1.0)
    _twice(sc, gpu_ctx, max_num_iterations=12, jacobi_scaling=0)


def test_free_calibration_and_bounds_bit_identical(gpu_ctx):
    sc = synth.ba_scene(10, 1200, 5, seed=13)
    calib = np.array([sc.K4[0][0] * 1.01, sc.K4[0][1] + 2.0, sc.K4[0][2] * 0.99, sc.K4[0][3] - 1.5])
    outs = []
    for _ in range(3):
        opt = E.default_options(); opt.max_num_iterations = 12
        c, p, k, summ = E.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, calib=calib, calib_tol=20.0, ref_cam=0,
                                      options=opt, ctx=gpu_ctx)
        outs.append((c, p, k, _log(summ)))
    for o in outs[1:]:
        assert o[3] == outs[0][3]
        assert all(np.array_equal(a, b) for a, b in zip(o[:3], outs[0][:3]))


def test_exact_accumulation_matches_oracle_trace(gpu_ctx, oracle_lib):
    """Rounding the Schur addends to the common grid costs at most 2^-50 of the bound per addend: the cost trace still follows
    the oracle to the 1e-9 of tests/test_ba_gpu.py on a scene with 3 orders of magnitude between the block norms."""
    sc = synth.ba_scene(25, 3000, 8, seed=17)
    opt = E.default_options(); opt.max_num_iterations = 10
    ropt = oracle_lib.ba_default_options(); ropt.max_num_iterations = 10
    _, _, summ = E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt, gpu_ctx)
    _, _, rs = oracle_lib.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ropt)
    assert summ.num_iterations == rs.num_iterations
    for a, b in zip(summ.log(), oracle_lib.iterations(rs)):
        assert abs(a.cost - b.cost) <= 1e-9 * b.cost, a.iteration


@pytest.mark.parametrize("n_cam,n_pt,per,reps,iters", [(512, 300000, 10, 4, 6), (107, 9000, 6, 20, 6), (43, 4000, 6, 25, 5), (200, 40000, 8, 8, 6)])
def test_dataflow_cholesky_repeat_solves_bit_identical(gpu_ctx, n_cam, n_pt, per, reps, iters):
    """VERDICT r03 "weak" 11 / "next" 4c: the tiled dataflow Cholesky (chol3_kernel / chol2_back_kernel, ba_chol_large.hip) hands
    tiles between workgroups through flags and assumes that a workgroup it waits for has been dispatched (bounded spins turn a
    broken launch into an error, not a hang).  scratch/chol_stress.py's repeat-solve check as a test: many solves of the same
    problem on one resident esfm_ba_problem -- every flag, counter and buffer re-used -- must give bit-identical parameters and
    iteration counts, at BA-512's 48 block columns and at three mid sizes (2 - 19 block columns)."""
    sc = synth.ba_scene(n_cam, n_pt, per, radius=40.0 if n_cam > 150 else 15.0, extent=8.0 if n_cam > 150 else 3.0, seed=5000 + n_cam)
    prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, gpu_ctx)
    opt = E.default_options(); opt.function_tolerance = 0; opt.parameter_tolerance = 0; opt.gradient_tolerance = 0; opt.max_num_iterations = iters
    ref = None
    try:
        for r in range(reps):
            prob.set_params(sc.cams0, sc.pts0)
            s = prob.solve(opt); gpu_ctx.synchronize()
            cams, pts = prob.get_params()
            sig = (s.final_cost, s.num_iterations, s.num_successful_steps, cams.tobytes(), pts.tobytes())
            if ref is None:
                ref = sig
            assert sig == ref, (n_cam, r, s.final_cost, ref[0])
        assert ref[1] == iters and np.isfinite(ref[0])
    finally:
        prob.close()

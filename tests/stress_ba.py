"""Randomised stress of the BA path against the oracle's LM trace: camera counts across every kernel-path threshold (LDS slab Schur +
one-workgroup solve, small / tiled Cholesky, matrix-core / window / plain Schur kernels, structure-aware solve), random track
structure (consecutive windows, random subsets, far pairs, single observations, unobserved cameras / points), observation order
camera-major or shuffled, with and without free intrinsics.
usage: python tests/stress_ba.py [--seconds S | --cases N] [--seed K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easysfm_amd as E
from easysfm_amd import synth
import oracle

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=None, help="run until this much time has passed")
ap.add_argument("--cases", type=int, default=None, help="... or for exactly this many cases (deterministic in --seed)")
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
budget = args.seconds if args.seconds is not None else (1e9 if args.cases is not None else 120.0)
max_cases = args.cases if args.cases is not None else 1 << 60
seed = args.seed
rng = np.random.default_rng(seed)
oracle.build()
oracle.set_num_threads(min(16, os.cpu_count() or 1))
ctx = E.Context(0)
K = np.array(synth.FOUNTAIN_K4, np.float64)

def trace_equal(summ, rs, constrained):
    a_log, b_log = summ.log(), oracle.iterations(rs)
    assert summ.termination == rs.termination and summ.num_iterations == rs.num_iterations, (summ.termination, rs.termination, summ.num_iterations, rs.num_iterations)
    assert [a.step_is_successful for a in a_log] == [b.step_is_successful for b in b_log]
    if constrained:
        assert [a.line_search_steps for a in a_log] == [b.line_search_steps for b in b_log]
    for a, b in zip(a_log, b_log):
        assert a.step_is_valid == b.step_is_valid, a.iteration
        # (round-off differences between two correct f64 solvers grow along the free gauge by ~1/damping per iteration: tight early, looser later)
        # (free intrinsics on a handful of cameras with an active bound are worse still: the ORACLE's own cost after one iteration moves by
        # 2.6e-9 when the initial points are perturbed by 1e-13 relative -- seed 8, case 1059 of round 5's open-ended run: 3 cameras, 111 observations)
        tight = 1e-8 if constrained else 1e-9
        assert abs(a.cost - b.cost) <= (tight if a.iteration <= 3 else 1e-5) * max(abs(b.cost), 1.0), (a.iteration, a.cost, b.cost)
        assert abs(a.trust_region_radius - b.trust_region_radius) <= (1e-6 if a.iteration <= 4 else 1e-3) * b.trust_region_radius, (a.iteration, 'radius', a.trust_region_radius, b.trust_region_radius)

def explained_by_conditioning(ss, solve_oracle, base_rs, cams0, pts0):
    """A trace outside the tolerances: is the ORACLE's own trace just as sensitive?  Its solve is repeated on inputs perturbed by 1e-13
    relative (six draws); the GPU's deviation is a conditioning artefact of the problem -- points seen once or twice, a handful of cameras,
    no camera held: round-off along the free gauge grows by ~1/damping per iteration -- iff at every iteration the oracle moves at least half
    as far under that perturbation as the GPU is away from it (costs and radii), and a difference in the accept / validity pattern or
    the iteration count is one iff the perturbed oracle shows another pattern than the unperturbed one as well.  Returns (bool, text)."""
    prng = np.random.default_rng(12345)
    b_log, a_log = oracle.iterations(base_rs), ss.log()
    n = min(len(a_log), len(b_log))
    pat = lambda log, s_: (s_.termination, s_.num_iterations, tuple((x.step_is_successful, x.step_is_valid) for x in log))
    same_pattern = pat(a_log, ss) == pat(b_log, base_rs)
    d_gpu = np.array([abs(a_log[k].cost - b_log[k].cost) / max(abs(b_log[k].cost), 1.0) for k in range(n)])
    r_gpu = np.array([abs(a_log[k].trust_region_radius - b_log[k].trust_region_radius) / b_log[k].trust_region_radius for k in range(n)])
    d_or, r_or, pattern_moves = np.zeros(n), np.zeros(n), False
    for _ in range(6):
        rs = solve_oracle(cams0 * (1 + prng.uniform(-1e-13, 1e-13, cams0.shape)), pts0 * (1 + prng.uniform(-1e-13, 1e-13, pts0.shape)))
        p_log = oracle.iterations(rs)
        pattern_moves = pattern_moves or pat(p_log, rs) != pat(b_log, base_rs)
        for k in range(min(n, len(p_log))):
            d_or[k] = max(d_or[k], abs(p_log[k].cost - b_log[k].cost) / max(abs(b_log[k].cost), 1.0))
            r_or[k] = max(r_or[k], abs(p_log[k].trust_region_radius - b_log[k].trust_region_radius) / b_log[k].trust_region_radius)
    if not same_pattern:
        return pattern_moves, f"accept / validity pattern differs; the oracle's own pattern {'also changes' if pattern_moves else 'does NOT change'} under a 1e-13 perturbation"
    ok = bool(np.all((d_gpu <= 1e-9) | (d_or >= 0.5 * d_gpu)) and np.all((r_gpu <= 1e-6) | (r_or >= 0.5 * r_gpu)))
    k = int(np.argmax(d_gpu))
    return ok, f"largest cost difference {d_gpu[k]:.1e} at iteration {a_log[k].iteration}; the oracle's own cost moves {d_or[k]:.1e} there under a 1e-13 perturbation of its input"


t_end = time.time() + budget
n_cases = n_artefacts = 0
while time.time() < t_end and n_cases < max_cases:
    n_cam = int(rng.choice([3, 5, 8, 16, 25, 26, 27, 30, 42, 43, 44, 64, 90, 107, 150, 220]))
    n_pt = int(rng.integers(max(40, 4 * n_cam), 60 * n_cam + 200))
    base = synth.ba_scene(n_cam, n_pt, 3, radius=10.0 + 0.1 * n_cam, extent=2.0 + 0.01 * n_cam, seed=int(rng.integers(1, 1 << 30)))
    style = int(rng.integers(0, 4))
    cam, pt = [], []
    for p in range(n_pt):
        if style == 0:                       # consecutive window
            k = int(rng.integers(2, min(n_cam, 12) + 1)); c0 = int(rng.integers(0, n_cam))
            cams = (c0 + np.arange(k)) % n_cam
        elif style == 1:                     # random subset of a window of 13
            w = min(13, n_cam); k = int(rng.integers(1, w + 1)); c0 = int(rng.integers(0, n_cam))
            cams = (c0 + rng.choice(w, size=k, replace=False)) % n_cam
        elif style == 2:                     # anything
            cams = rng.choice(n_cam, size=int(rng.integers(1, min(n_cam, 9) + 1)), replace=False)
        else:                                # two groups + a few bridges
            half = max(2, n_cam // 2); g = 0 if p % 2 else n_cam - half
            cams = g + rng.choice(half, size=int(rng.integers(2, min(half, 8) + 1)), replace=False)
            if p % 41 == 0: cams = np.unique(np.concatenate([cams, rng.choice(n_cam, 1)]))
        if p % 301 == 7: continue            # a point nobody sees
        cams = np.unique(cams)
        cam += list(cams); pt += [p] * len(cams)
    cam = np.array(cam, np.int32); pt = np.array(pt, np.int32)
    if rng.random() < 0.5:
        o = np.lexsort((pt, cam)); cam, pt = cam[o], pt[o]
    elif rng.random() < 0.5:
        o = rng.permutation(len(cam)); cam, pt = cam[o], pt[o]
    Rs = np.stack([synth.aa_to_R(base.cams_gt[c, :3]) for c in range(n_cam)])
    Pc = np.einsum("nij,nj->ni", Rs[cam], base.pts_gt[pt]) + base.cams_gt[cam, 3:]
    uv = np.stack([Pc[:, 0] / Pc[:, 2] * K[0] + K[1], Pc[:, 1] / Pc[:, 2] * K[2] + K[3]], 1) + 0.5 * rng.standard_normal((len(cam), 2))
    out = rng.random(len(cam)) < 0.02
    uv[out] += rng.uniform(-50, 50, (int(out.sum()), 2))
    uv = uv.astype(np.float32)
    iters = int(rng.integers(2, 6))
    opt = E.default_options(); opt.max_num_iterations = iters
    ropt = oracle.ba_default_options(); ropt.max_num_iterations = iters
    calib = rng.random() < 0.25
    mode = [None, "dense", "sparse"][int(rng.integers(0, 3))]
    os.environ.pop("ESFM_BA_SOLVE", None)
    if mode: os.environ["ESFM_BA_SOLVE"] = mode
    tag = dict(seed=seed, case=n_cases, n_cam=n_cam, n_pt=n_pt, n_obs=len(cam), style=style, calib=calib, mode=mode, iters=iters)
    ss = rs = None
    try:
        if calib:
            c0 = K * np.array([1.02, 0.99, 0.98, 1.01]); tol = float(rng.choice([8.0, 20.0, 100.0]))
            cs, ps, cal, ss = E.ba_solve_ex(cam, pt, uv, None, base.cams0, base.pts0, calib=c0, calib_tol=tol, options=opt, ctx=ctx)
            solve_oracle = lambda c_, p_: oracle.ba_solve_ex(cam, pt, uv, None, c_, p_, calib=c0, calib_tol=tol, options=ropt)[3]
            rc, rp, rk, rs = oracle.ba_solve_ex(cam, pt, uv, None, base.cams0, base.pts0, calib=c0, calib_tol=tol, options=ropt)
        else:
            cs, ps, ss = E.ba_solve(cam, pt, uv, base.K4, base.cams0, base.pts0, opt, ctx)
            solve_oracle = lambda c_, p_: oracle.ba_solve(cam, pt, uv, base.K4, c_, p_, ropt)[2]
            rc, rp, rs = oracle.ba_solve(cam, pt, uv, base.K4, base.cams0, base.pts0, ropt)
        trace_equal(ss, rs, calib)
        if iters <= 3: assert np.allclose(cs, rc, rtol=1e-5, atol=1e-5) and np.allclose(ps, rp, rtol=1e-5, atol=1e-5), "parameters"
    except Exception as e:
        os.makedirs("gpurun_out", exist_ok=True); np.savez(f"gpurun_out/stress_ba_fail_{seed}_{n_cases}.npz", cam=cam, pt=pt, uv=uv, cams0=base.cams0, pts0=base.pts0, K4=base.K4)
        ok, why = (False, "the solve itself failed") if (ss is None or rs is None) else explained_by_conditioning(ss, solve_oracle, rs, base.cams0, base.pts0)
        if not ok:
            print("BA MISMATCH", tag, repr(e)[:300], "|", why, flush=True)
            sys.exit(1)
        print("  conditioning artefact", tag, repr(e)[:160], "|", why, flush=True)
        n_artefacts += 1
    n_cases += 1
os.environ.pop("ESFM_BA_SOLVE", None)
print(f"stress_ba seed {seed}: {n_cases} cases, traces and parameters equal to the oracle's within the tests' tolerances"
      + (f" ({n_artefacts} of them outside, each one matched by the oracle's OWN sensitivity to a 1e-13 perturbation of its input: conditioning artefacts, listed above)" if n_artefacts else ""))

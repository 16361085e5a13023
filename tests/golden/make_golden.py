#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ (run once in the build container; outputs are
committed).  The reference ships no tests or fixtures and cannot be built/imported here (SURVEY.md
section 8c), so these vectors come from a SECOND, independent restatement of the reference
semantics written in numpy / torch -- different code, different linear algebra -- against which the
C oracle (oracle/*.c) is pinned by tests/test_oracle_golden.py:

  match_hamming_cases.npz   hand-built ORB-like sets (forced ties for 1st/2nd place, all-equal rows,
                            nt in {0,1,2,3}, ratio boundary d0 == ratio*d1, sizes not multiples of 64);
                            expected 2-NN by stable integer argsort, ratio filter in float64
  match_l2_cases.npz        unit-norm 64-D sets with planted near-duplicates and exact duplicates;
                            expected 2-NN from a numpy float32 evaluation of the canonical summation
                            order (vectorised over pairs, same operation order), stable argsort
  ba_jacobian_cases.npz     residual + Jacobians of the reprojection functor (ba.h:113-153) by torch
                            autograd in float64, including the small-angle branch
  ba_lm_trace.npz           4 cameras / 50 points (Cauchy) and 5 cameras / 120 points (squared loss, far start, with
                            rejected steps): cost / radius / step norm /
                            accept flag per iteration from a dense (no Schur) numpy LM that solves
                            (J'J + D^2) y = J'r with numpy.linalg, plus the zero-noise known answer
  ba_lm_constrained.npz     the bounds-constrained variants of solveBA (ba.cpp:155-162 reference camera, ba.cpp:167-196
                            free shared intrinsics within +-tolerance): the same dense LM with the intrinsics as four more
                            columns (ba.h:170-222 by torch autograd), projection onto the box, projected gradient norm
                            and the Armijo line search with cubic interpolation (numpy.linalg.solve for the
                            interpolating polynomial, numpy.roots for its critical points)

  sor_cases.npz             statistical outlier removal (cloudprocessing.hpp:24-36 = pcl::StatisticalOutlierRemoval, MeanK 50,
                            StddevMulThresh 2.0): clustered clouds with planted outliers, exact duplicates, a non-finite
                            point, a cloud smaller than MeanK + 1; expected mean distances from a full float32 distance
                            matrix evaluated as ((dx*dx + dy*dy) + dz*dz), numpy sort, sequential float64 cumsum of float32
                            square roots; threshold from sequential cumsums
  triangulation_cases.npz   cv::triangulatePoints (estimate_motion.cpp:263, :333): random two-view geometries incl. a short
                            baseline and noisy points; expected homogeneous points from numpy.linalg.svd of the f64 DLT matrix
  ransac_cases.npz          two-view verification (estimate_motion.cpp:49-67): ten exact 5-point samples with their
                            ground-truth essential matrix (known answer: it must be among the solver's models, and every
                            model must satisfy the epipolar, determinant and trace constraints); recoverPose cases whose
                            expected R, t, cheirality mask come from a numpy restatement (numpy.linalg.svd for the
                            decomposition and for each point's 4 x 4 DLT system)
  fountain_pair_half.npz    INPUT ONLY: the reference's first two test images (test_data/images_25/0000.png, 0001.png), decoded
                            with PIL, BGR order, every second pixel (384 x 256 x 3 uint8) -- real texture for the SURF tests
                            (written by a one-off snippet, not by this script: the PNGs live in /root/reference)

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from easysfm_amd import synth  # noqa: E402  (seeded generators only; no GPU)


# ------------------------------------------------------------------------------------------------ matching
def knn2_from_matrix(D: np.ndarray):
    """Two smallest per row by (value, column) -- what an ascending scan with strict-< insertion yields."""
    nq, nt = D.shape
    idx = np.full((nq, 2), -1, np.int32)
    dist = np.full((nq, 2), np.finfo(np.float32).max, np.float32)
    if nt:
        order = np.argsort(D, axis=1, kind="stable")[:, :2]
        k = order.shape[1]
        idx[:, :k] = order
        dist[:, :k] = np.take_along_axis(D, order, 1).astype(np.float32)
    return idx, dist


def ratio_filter(idx, dist, ratio):
    keep = (idx[:, 0] >= 0) & (idx[:, 1] >= 0) & (dist[:, 0].astype(np.float64) < ratio * dist[:, 1].astype(np.float64))
    q = np.nonzero(keep)[0].astype(np.int32)
    return q, idx[keep, 0], dist[keep, 0]


def hamming_matrix(q, t):
    if len(t) == 0:
        return np.zeros((len(q), 0), np.int64)
    return np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(-1).astype(np.int64)


def l2_matrix_canonical(q, t):
    """float32 distances, canonical order: 8 partial sums over blocks of 8 (mul and add separate),
    (acc[c]+acc[c+4]) summed left to right, scalar tail, then sqrt -- vectorised over all pairs."""
    q = q.astype(np.float32); t = t.astype(np.float32)
    nq, dim = q.shape
    nt = t.shape[0]
    if nt == 0:
        return np.zeros((nq, 0), np.float32)
    acc = np.zeros((nq, nt, 8), np.float32)
    nb = dim // 8
    for b in range(nb):
        d = q[:, None, 8 * b:8 * b + 8] - t[None, :, 8 * b:8 * b + 8]
        acc = acc + d * d
    s = [acc[..., c] + acc[..., c + 4] for c in range(4)]
    d2 = ((s[0] + s[1]) + s[2]) + s[3]
    for j in range(8 * nb, dim):
        d = q[:, None, j] - t[None, :, j]
        d2 = d2 + d * d
    return np.sqrt(d2).astype(np.float32)


def make_hamming():
    rng = np.random.default_rng(12345)
    cases = {}
    # 0: forced ties
    t = np.zeros((70, 32), np.uint8); t[::3, 0] = 1; t[1::3, 1] = 3
    q = np.zeros((37, 32), np.uint8); q[20:, 5] = 0xFF
    cases["ties"] = (q, t)
    # 1: all rows equal
    cases["all_equal"] = (np.full((9, 32), 0xA5, np.uint8), np.full((13, 32), 0xA5, np.uint8))
    # 2..5: tiny train sets
    for nt in (0, 1, 2, 3):
        cases[f"nt{nt}"] = (rng.integers(0, 256, (5, 32), dtype=np.uint8), rng.integers(0, 256, (nt, 32), dtype=np.uint8))
    # 6: ratio boundary: d0 = 4, d1 = 5 (0.8 * 5 == 4 exactly -> must be rejected), d0 = 3, d1 = 5 accepted
    t = np.zeros((2, 32), np.uint8); t[0, 0] = 0x0F; t[1, 0] = 0x1F
    q = np.zeros((2, 32), np.uint8); q[1, 0] = 0x01
    cases["boundary"] = (q, t)
    # 7: random, sizes not multiples of 64
    o = synth.orb_like_sets(2, 131, pool=97, seed_base=7000)
    cases["random"] = (o[1], o[0][:101])
    out = {}
    for name, (q, t) in cases.items():
        idx, dist = knn2_from_matrix(hamming_matrix(q, t))
        out[f"{name}.q"] = q; out[f"{name}.t"] = t; out[f"{name}.idx"] = idx; out[f"{name}.dist"] = dist
        for r in (0.5, 0.8, 1.0):
            a, b, c = ratio_filter(idx, dist, r)
            out[f"{name}.m{r}.q"] = a; out[f"{name}.m{r}.t"] = b; out[f"{name}.m{r}.d"] = c
    np.savez_compressed(os.path.join(HERE, "match_hamming_cases.npz"), **out)


def make_l2():
    rng = np.random.default_rng(54321)
    cases = {}
    s = synth.surf_like_sets(2, 90, pool=64, seed_base=8000)
    cases["surf_like"] = (s[1], s[0][:77])
    base = s[0][:20]
    t = np.concatenate([base, base[:10], base[:5] * np.float32(1.0 + 1e-7), rng.standard_normal((30, 64)).astype(np.float32) * 0.2])
    t = t[rng.permutation(len(t))].astype(np.float32)
    q = np.concatenate([base[:12], base[:12] + 1e-4 * rng.standard_normal((12, 64)).astype(np.float32)]).astype(np.float32)
    cases["duplicates"] = (q, t)
    for nt in (0, 1, 2):
        cases[f"nt{nt}"] = (s[1][:4], s[0][:nt])
    cases["dim37"] = (rng.standard_normal((11, 37)).astype(np.float32), rng.standard_normal((23, 37)).astype(np.float32))
    out = {}
    near = []
    for name, (q, t) in cases.items():
        D = l2_matrix_canonical(q, t)
        idx, dist = knn2_from_matrix(D)
        out[f"{name}.q"] = q; out[f"{name}.t"] = t; out[f"{name}.idx"] = idx; out[f"{name}.dist"] = dist
        for r in (0.5, 0.8):
            a, b, c = ratio_filter(idx, dist, r)
            out[f"{name}.m{r}.q"] = a; out[f"{name}.m{r}.t"] = b; out[f"{name}.m{r}.d"] = c
        if D.shape[1] >= 2:
            u = dist.view(np.uint32).astype(np.int64)
            near.append(int(np.sum(np.abs(u[:, 0] - u[:, 1]) < 4)))
    out["near_tie_audit"] = np.array(near)   # queries whose two best distances are < 4 ulp apart, per case
    np.savez_compressed(os.path.join(HERE, "match_l2_cases.npz"), **out)


# ------------------------------------------------------------------------------------------------ BA
def torch_residual(cam, pt, K4, uv):
    """ReprojectErrorTerm_fixcalib (ba.h:113-153) in torch float64, both AngleAxisRotatePoint branches."""
    import torch
    aa, tr = cam[:3], cam[3:]
    theta2 = (aa * aa).sum()
    if theta2.item() > np.finfo(np.float64).eps:
        theta = torch.sqrt(theta2)
        w = aa / theta
        p = pt * torch.cos(theta) + torch.linalg.cross(w, pt) * torch.sin(theta) + w * (w @ pt) * (1.0 - torch.cos(theta))
    else:
        p = pt + torch.linalg.cross(aa, pt)
    p = p + tr
    u = p[0] / p[2] * float(K4[0]) + float(K4[1])
    v = p[1] / p[2] * float(K4[2]) + float(K4[3])
    return torch.stack([float(uv[0]) - u, float(uv[1]) - v])


def make_ba_jacobians():
    import torch
    rng = np.random.default_rng(777)
    K4 = np.array(synth.FOUNTAIN_K4, np.float32)
    cams, pts, uvs, rs, Jcs, Jps = [], [], [], [], [], []
    for i in range(24):
        cam = np.concatenate([rng.standard_normal(3) * (0.8 if i % 3 else 1e-3), rng.standard_normal(3)])
        if i in (0, 1):
            cam[:3] = 0.0 if i == 0 else np.array([1e-9, -2e-9, 5e-10])   # small-angle branch
        pt = rng.uniform(-2, 2, 3) + np.array([0, 0, 8.0])
        uv = rng.uniform(0, 700, 2).astype(np.float32)
        c = torch.tensor(cam, dtype=torch.float64, requires_grad=True)
        p = torch.tensor(pt, dtype=torch.float64, requires_grad=True)
        r = torch_residual(c, p, K4, uv)
        Jc = np.zeros((2, 6)); Jp = np.zeros((2, 3))
        for k in range(2):
            g = torch.autograd.grad(r[k], [c, p], retain_graph=True)
            Jc[k] = g[0].numpy(); Jp[k] = g[1].numpy()
        cams.append(cam); pts.append(pt); uvs.append(uv); rs.append(r.detach().numpy()); Jcs.append(Jc); Jps.append(Jp)
    np.savez_compressed(os.path.join(HERE, "ba_jacobian_cases.npz"), cams=np.array(cams), pts=np.array(pts), uvs=np.array(uvs),
                        K4=K4, r=np.array(rs), Jc=np.array(Jcs), Jp=np.array(Jps))


def dense_lm(sc, max_iter, cauchy_a=0.5):
    """Dense restatement of Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy (no Schur
    complement: the damped normal equations are solved whole).  Rules: oracle/ba_ref.c header."""
    import torch
    n_cam, n_pt, n_obs = sc.n_cam, sc.n_pt, sc.n_obs
    npar = 6 * n_cam + 3 * n_pt

    def evaluate(x, jac):
        cams = x[:6 * n_cam].reshape(n_cam, 6); pts = x[6 * n_cam:].reshape(n_pt, 3)
        r = np.zeros(2 * n_obs); J = np.zeros((2 * n_obs, npar)) if jac else None
        cost = 0.0
        for k in range(n_obs):
            c, p = int(sc.cam_idx[k]), int(sc.pt_idx[k])
            ct = torch.tensor(cams[c], dtype=torch.float64, requires_grad=jac)
            pt = torch.tensor(pts[p], dtype=torch.float64, requires_grad=jac)
            res = torch_residual(ct, pt, sc.K4[c], sc.uv[k])
            rv = res.detach().numpy()
            s = float(rv @ rv)
            if cauchy_a > 0:
                b = cauchy_a ** 2
                rho0 = b * np.log1p(s / b); rho1 = max(1.0 / (1.0 + s / b), np.finfo(np.float64).tiny)
            else:
                rho0, rho1 = s, 1.0
            cost += 0.5 * rho0
            sq = np.sqrt(rho1)
            r[2 * k:2 * k + 2] = rv * sq
            if jac:
                for i in range(2):
                    g = torch.autograd.grad(res[i], [ct, pt], retain_graph=True)
                    J[2 * k + i, 6 * c:6 * c + 6] = g[0].numpy() * sq
                    J[2 * k + i, 6 * n_cam + 3 * p:6 * n_cam + 3 * p + 3] = g[1].numpy() * sq
        return cost, r, J

    x = np.concatenate([sc.cams0.ravel(), sc.pts0.ravel()])
    cost, r, J = evaluate(x, True)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))
    J = J * scale
    radius, nu = 1e4, 2.0
    x_norm = np.linalg.norm(x)
    log = [dict(cost=cost, radius=radius, step_norm=0.0, ok=1, gmax=np.abs((J / scale).T @ r).max())]
    diag = None
    reuse = False
    for it in range(1, max_iter + 1):
        if not reuse:
            diag = np.clip((J * J).sum(0), 1e-6, 1e32)
        A = J.T @ J + np.diag(diag / radius)
        y = np.linalg.solve(A, J.T @ r)
        step = -y
        reuse = True
        m = J @ step
        mcc = -m @ (r + m / 2.0)
        cand = x + step * scale
        ccost, _, _ = evaluate(cand, False)
        step_norm = np.linalg.norm(x - cand)
        if step_norm <= 1e-8 * (x_norm + 1e-8) or abs(cost - ccost) <= 1e-6 * cost:
            log.append(dict(cost=cost, radius=radius, step_norm=step_norm, ok=0, gmax=log[-1]["gmax"], stop=1)); break
        rho = (cost - ccost) / mcc
        if rho > 1e-3:
            x = cand; x_norm = np.linalg.norm(x)
            cost, r, J = evaluate(x, True)
            g = np.abs(J.T @ r).max()
            J = J * scale
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3)); nu = 2.0; reuse = False
            log.append(dict(cost=cost, radius=radius, step_norm=step_norm, ok=1, gmax=g))
        else:
            radius /= nu; nu *= 2.0
            log.append(dict(cost=ccost, radius=radius, step_norm=step_norm, ok=0, gmax=log[-1]["gmax"]))
    return x, log


def torch_residual_calib(cam, pt, calib, uv):
    """ReprojectErrorTerm_updatecalib (ba.h:170-216): calib = fx, cx, fy, cy as a differentiable tensor."""
    import torch
    aa, tr = cam[:3], cam[3:]
    theta2 = (aa * aa).sum()
    if theta2.item() > np.finfo(np.float64).eps:
        theta = torch.sqrt(theta2)
        w = aa / theta
        p = pt * torch.cos(theta) + torch.linalg.cross(w, pt) * torch.sin(theta) + w * (w @ pt) * (1.0 - torch.cos(theta))
    else:
        p = pt + torch.linalg.cross(aa, pt)
    p = p + tr
    u = p[0] / p[2] * calib[0] + calib[1]
    v = p[1] / p[2] * calib[2] + calib[3]
    return torch.stack([float(uv[0]) - u, float(uv[1]) - v])


def armijo_next_step(f0, g0, prev, cur, lo, hi):
    """line_search.cc InterpolatingPolynomialMinimizingStepSize (CUBIC) + polynomial.cc MinimizePolynomial.
    prev / cur = (x, f, g) or None."""
    if cur is None:
        raise ValueError
    rows, rhs = [], []
    samples = [(0.0, f0, g0), cur] + ([prev] if prev is not None else [])
    n = 2 * len(samples)
    for (x, f, g) in samples:
        rows.append([x ** j for j in range(n)]); rhs.append(f)
        rows.append([0.0] + [j * x ** (j - 1) for j in range(1, n)]); rhs.append(g)
    coef = np.linalg.solve(np.array(rows), np.array(rhs))          # lowest degree first
    poly = np.polynomial.Polynomial(coef)
    cands = [(lo + hi) / 2.0, lo, hi]
    droots = np.roots(poly.deriv().coef[::-1])
    cands += [float(np.real(z)) for z in droots if lo <= np.real(z) <= hi]   # Ceres tries real parts of all roots
    best = cands[0]
    for c in cands[1:]:
        if poly(c) < poly(best):
            best = c
    return best


def dense_lm_constrained(sc, max_iter, cauchy_a, calib0=None, calib_tol=0.0, ref_cam=-1, ref_thr=1e-10):
    """dense_lm + Ceres' bounds handling (trust_region_minimizer.cc: IterationZero projection, projected gradient
    norm, DoLineSearch, Plus() projects).  Unknown vector: cameras | points | (fx, cx, fy, cy)."""
    import torch
    n_cam, n_pt, n_obs = sc.n_cam, sc.n_pt, sc.n_obs
    free = calib0 is not None
    npar = 6 * n_cam + 3 * n_pt + (4 if free else 0)
    ko = 6 * n_cam + 3 * n_pt
    lo = np.full(npar, -np.inf); up = np.full(npar, np.inf)
    if ref_cam >= 0:
        lo[6 * ref_cam:6 * ref_cam + 6] = -ref_thr; up[6 * ref_cam:6 * ref_cam + 6] = ref_thr
    if free:
        lo[ko:] = np.asarray(calib0) - calib_tol; up[ko:] = np.asarray(calib0) + calib_tol

    def plus(x, d):
        return np.minimum(np.maximum(x + d, lo), up)

    def evaluate(x, jac):
        cams = x[:6 * n_cam].reshape(n_cam, 6); pts = x[6 * n_cam:ko].reshape(n_pt, 3)
        r = np.zeros(2 * n_obs); J = np.zeros((2 * n_obs, npar)) if jac else None
        cost = 0.0
        kt = torch.tensor(x[ko:], dtype=torch.float64, requires_grad=jac) if free else None
        for k in range(n_obs):
            c, p = int(sc.cam_idx[k]), int(sc.pt_idx[k])
            ct = torch.tensor(cams[c], dtype=torch.float64, requires_grad=jac)
            pt = torch.tensor(pts[p], dtype=torch.float64, requires_grad=jac)
            res = torch_residual_calib(ct, pt, kt, sc.uv[k]) if free else torch_residual(ct, pt, sc.K4[c], sc.uv[k])
            rv = res.detach().numpy()
            sq_norm = float(rv @ rv)
            if cauchy_a > 0:
                b = cauchy_a ** 2
                rho0 = b * np.log1p(sq_norm / b); rho1 = max(1.0 / (1.0 + sq_norm / b), np.finfo(np.float64).tiny)
            else:
                rho0, rho1 = sq_norm, 1.0
            cost += 0.5 * rho0
            sq = np.sqrt(rho1)
            r[2 * k:2 * k + 2] = rv * sq
            if jac:
                for i in range(2):
                    g = torch.autograd.grad(res[i], [ct, pt] + ([kt] if free else []), retain_graph=True)
                    J[2 * k + i, 6 * c:6 * c + 6] = g[0].numpy() * sq
                    J[2 * k + i, 6 * n_cam + 3 * p:6 * n_cam + 3 * p + 3] = g[1].numpy() * sq
                    if free:
                        J[2 * k + i, ko:] = g[2].numpy() * sq
        return cost, r, J

    def pgrad_norm(x, g):
        return np.abs(x - plus(x, -g)).max()

    x = np.concatenate([sc.cams0.ravel(), sc.pts0.ravel()] + ([np.asarray(calib0, np.float64)] if free else []))
    x = plus(x, 0.0)
    cost, r, J = evaluate(x, True)
    grad = J.T @ r
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))
    J = J * scale
    radius, nu = 1e4, 2.0
    x_norm = np.linalg.norm(x)
    log = [dict(cost=cost, radius=radius, step_norm=0.0, ok=1, gmax=pgrad_norm(x, grad), ls=0)]
    reuse = False
    diag = None
    for it in range(1, max_iter + 1):
        if not reuse:
            diag = np.clip((J * J).sum(0), 1e-6, 1e32)
        A = J.T @ J + np.diag(diag / radius)
        step = -np.linalg.solve(A, J.T @ r)
        reuse = True
        m = J @ step
        mcc = -m @ (r + m / 2.0)
        delta = step * scale
        # DoLineSearch (Armijo, from step size 1)
        g0 = grad @ delta
        dmax = np.abs(delta).max()

        def phi(t):
            xt = plus(x, t * delta)
            f, rr, JJ = evaluate(xt, True)
            return f, (JJ.T @ rr) @ delta

        prev, cur, n_ls, ls_ok = None, None, 0, True
        f, g = phi(1.0); cur = (1.0, f, g)
        while cur[1] > cost + 1e-4 * g0 * cur[0]:
            n_ls += 1
            if n_ls >= 20:
                ls_ok = False; break
            t = armijo_next_step(cost, g0, prev, cur, 1e-3 * cur[0], 0.6 * cur[0])
            if t * dmax < 1e-9:
                ls_ok = False; break
            prev = cur
            f, g = phi(t); cur = (t, f, g)
        if ls_ok:
            delta = delta * cur[0]
        cand = plus(x, delta)
        ccost, _, _ = evaluate(cand, False)
        step_norm = np.linalg.norm(x - cand)
        if step_norm <= 1e-8 * (x_norm + 1e-8) or abs(cost - ccost) <= 1e-6 * cost:
            log.append(dict(cost=cost, radius=radius, step_norm=step_norm, ok=0, gmax=log[-1]["gmax"], ls=n_ls, stop=1)); break
        rho = (cost - ccost) / mcc
        if rho > 1e-3:
            x = cand; x_norm = np.linalg.norm(x)
            cost, r, J = evaluate(x, True)
            grad = J.T @ r
            J = J * scale
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3)); nu = 2.0; reuse = False
            log.append(dict(cost=cost, radius=radius, step_norm=step_norm, ok=1, gmax=pgrad_norm(x, grad), ls=n_ls))
        else:
            radius /= nu; nu *= 2.0
            log.append(dict(cost=ccost, radius=radius, step_norm=step_norm, ok=0, gmax=log[-1]["gmax"], ls=n_ls))
    return x, log


def make_ba_constrained():
    out = {}
    K = np.array(synth.FOUNTAIN_K4, np.float64)
    cases = (
        # free intrinsics, loose box (bounds never active), Cauchy
        ("calib_loose", 0.5, (4, 60, 3), 8, dict(seed=41), K * np.array([1.02, 0.99, 0.98, 1.01]), 100.0, -1),
        # free intrinsics, tight box (fx / fy hit the bound; the line search contracts)
        ("calib_tight", 0.5, (4, 60, 3), 10, dict(seed=41), K * np.array([1.03, 0.99, 0.97, 1.01]), 6.0, -1),
        # reference camera held at the origin by +-1e-10 bounds, fixed intrinsics
        ("refcam", 0.5, (4, 60, 3), 8, dict(seed=43), None, 0.0, 0),
        # both, squared loss
        ("both_squared", -1.0, (5, 80, 4), 8, dict(seed=44, outlier_frac=0.0), K * np.array([0.98, 1.01, 1.02, 0.99]), 10.0, 0),
    )
    for tag, a, shape, iters, kw, calib0, tol, ref in cases:
        sc = synth.ba_scene(*shape, **kw)
        if ref >= 0:
            sc = synth.in_reference_frame(sc, ref)   # the reference pipeline keeps its reference frame at the origin
        x, log = dense_lm_constrained(sc, iters, a, calib0, tol, ref)
        for f in ("cam_idx", "pt_idx", "uv", "K4", "cams0", "pts0"):
            out[f"{tag}.{f}"] = getattr(sc, f)
        out[f"{tag}.cauchy_a"] = np.float64(a)
        out[f"{tag}.calib0"] = np.zeros(0) if calib0 is None else np.asarray(calib0, np.float64)
        out[f"{tag}.calib_tol"] = np.float64(tol)
        out[f"{tag}.ref_cam"] = np.int32(ref)
        for key in ("cost", "radius", "step_norm", "ok", "gmax", "ls"):
            out[f"{tag}.{key}"] = np.array([e[key] for e in log])
        out[f"{tag}.x_final"] = x
        print(tag, "pattern", "".join(str(e["ok"]) for e in log), "ls", [e["ls"] for e in log],
              "cost", log[0]["cost"], "->", log[-1]["cost"])
    np.savez_compressed(os.path.join(HERE, "ba_lm_constrained.npz"), **out)


def make_ba_trace():
    out = {}
    for tag, a, shape, iters, kw in (("cauchy", 0.5, (4, 50, 3), 10, dict(seed=31)),
                                     ("squared_rejects", -1.0, (5, 120, 4), 12,
                                      dict(seed=12, start_noise=(0.5, 2.0, 2.0), outlier_frac=0.0))):
        sc = synth.ba_scene(*shape, **kw)
        x, log = dense_lm(sc, iters, a)
        for f in ("cam_idx", "pt_idx", "uv", "K4", "cams0", "pts0"):
            out[f"{tag}.{f}"] = getattr(sc, f)
        out[f"{tag}.cauchy_a"] = np.float64(a)
        out[f"{tag}.cost"] = np.array([e["cost"] for e in log])
        out[f"{tag}.radius"] = np.array([e["radius"] for e in log])
        out[f"{tag}.step_norm"] = np.array([e["step_norm"] for e in log])
        out[f"{tag}.ok"] = np.array([e["ok"] for e in log])
        out[f"{tag}.gmax"] = np.array([e["gmax"] for e in log])
        out[f"{tag}.x_final"] = x
        print(tag, "pattern", "".join(str(e["ok"]) for e in log), "cost", log[0]["cost"], "->", log[-1]["cost"])
    # known answer: exact observations at ground truth
    sc = synth.ba_scene(4, 30, 3, seed=32, uv_noise=0.0, outlier_frac=0.0, start_noise=(0, 0, 0))
    for f in ("cam_idx", "pt_idx", "uv", "K4", "cams_gt", "pts_gt"):
        out[f"zero.{f}"] = getattr(sc, f)
    np.savez_compressed(os.path.join(HERE, "ba_lm_trace.npz"), **out)


# ------------------------------------------------------------------------------------------------ cloud
def sor_numpy(P, mean_k, std_mul):
    P = np.asarray(P, np.float32)
    n = len(P)
    fin = np.isfinite(P[:, :3]).all(axis=1)
    Q = P[fin, :3]
    md = np.zeros(n, np.float32)
    dx = P[:, None, 0] - Q[None, :, 0]; dy = P[:, None, 1] - Q[None, :, 1]; dz = P[:, None, 2] - Q[None, :, 2]
    with np.errstate(invalid="ignore"):
        D = ((dx * dx + dy * dy) + dz * dz).astype(np.float32)
    for i in range(n):
        if not fin[i]:
            continue
        row = np.sort(D[i])[1:mean_k + 1]                      # entry 0 is the point itself
        md[i] = np.float32(np.cumsum(np.sqrt(row).astype(np.float64))[-1] / mean_k) if len(row) else np.float32(0)
    valid = int(fin.sum())
    s = np.cumsum(md.astype(np.float64))[-1]
    sq = np.cumsum((md * md).astype(np.float64))[-1]
    mean = s / valid
    var = (sq - s * s / valid) / (valid - 1.0)
    thr = mean + std_mul * np.sqrt(var)
    keep = ~(md.astype(np.float64) > thr)
    return md, thr, keep


def make_sor():
    rng = np.random.default_rng(4242)
    out = {}
    cases = {}
    blob = np.concatenate([rng.normal(0, 1.0, (1500, 3)), rng.normal([6, 0, 0], 0.3, (500, 3)), rng.uniform(-15, 15, (40, 3))])
    cases["blobs"] = (blob.astype(np.float32), 50, 2.0)
    dup = rng.normal(0, 1, (300, 3)).astype(np.float32)
    dup[50:60] = dup[0]                                        # exact duplicates: zero distances beyond entry 0
    dup[100] = [np.nan, 0, 0]; dup[101] = [0, np.inf, 0]       # non-finite points: excluded from the search, distance 0
    cases["dups_nonfinite"] = (dup, 50, 2.0)
    cases["small"] = (rng.normal(0, 1, (30, 3)).astype(np.float32), 50, 2.0)    # fewer than MeanK + 1 points
    g = np.stack(np.meshgrid(np.arange(8.0), np.arange(8.0), np.arange(8.0)), -1).reshape(-1, 3).astype(np.float32)
    cases["grid_ties"] = (g, 20, 1.0)                          # massive distance ties at the k-th place
    xyzrgb = np.zeros((400, 8), np.float32); xyzrgb[:, :3] = rng.normal(0, 2, (400, 3)); xyzrgb[:, 3] = 1.0; xyzrgb[:, 4] = rng.random(400)
    cases["stride8"] = (xyzrgb, 10, 1.5)                       # pcl::PointXYZRGB layout: 8 floats per point
    for tag, (P, k, m) in cases.items():
        md, thr, keep = sor_numpy(P, k, m)
        out[f"{tag}.points"] = P; out[f"{tag}.mean_k"] = np.int32(k); out[f"{tag}.std_mul"] = np.float64(m)
        out[f"{tag}.mean_dist"] = md; out[f"{tag}.threshold"] = np.float64(thr); out[f"{tag}.keep"] = keep
        print("sor", tag, len(P), "kept", int(keep.sum()), "thr", thr)
    np.savez_compressed(os.path.join(HERE, "sor_cases.npz"), **out)


def make_triangulation():
    rng = np.random.default_rng(606)
    out = {}
    for tag, n, base, noise in (("wide", 300, 1.0, 0.0), ("noisy", 300, 1.0, 2e-3), ("short_baseline", 200, 0.05, 2e-4)):
        R = synth.aa_to_R(rng.normal(0, 0.15, 3)); t = np.array([base, 0.1 * base, -0.05 * base])
        P1 = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)
        P2 = np.hstack([R, t[:, None]]).astype(np.float32)
        X = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0, 7.0])
        x1 = (X[:, :2] / X[:, 2:3] + rng.normal(0, noise, (n, 2))).astype(np.float32)
        Xc = X @ R.T + t
        x2 = (Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, noise, (n, 2))).astype(np.float32)
        H = np.zeros((n, 4))
        for i in range(n):
            A = np.array([x1[i, 0].astype(np.float64) * P1[2].astype(np.float64) - P1[0], x1[i, 1].astype(np.float64) * P1[2].astype(np.float64) - P1[1],
                          x2[i, 0].astype(np.float64) * P2[2].astype(np.float64) - P2[0], x2[i, 1].astype(np.float64) * P2[2].astype(np.float64) - P2[1]], np.float64)
            H[i] = np.linalg.svd(A)[2][3]
        out[f"{tag}.P1"] = P1; out[f"{tag}.P2"] = P2; out[f"{tag}.x1"] = x1; out[f"{tag}.x2"] = x2
        out[f"{tag}.points4d"] = H.astype(np.float32); out[f"{tag}.X"] = X
    np.savez_compressed(os.path.join(HERE, "triangulation_cases.npz"), **out)


def recover_pose_numpy(E, p1, p2, K4, mask):
    """cv::recoverPose restated with numpy.linalg.svd (decomposeEssentialMat + cheirality with distanceThresh 50)."""
    fx, cx, fy, cy = [float(v) for v in K4]
    a = np.stack([(p1[:, 0].astype(np.float64) - cx) / fx, (p1[:, 1].astype(np.float64) - cy) / fy], 1)
    b = np.stack([(p2[:, 0].astype(np.float64) - cx) / fx, (p2[:, 1].astype(np.float64) - cy) / fy], 1)
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0: U = -U
    if np.linalg.det(Vt) < 0: Vt = -Vt
    W = np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]])
    R1, R2, t = U @ W @ Vt, U @ W.T @ Vt, U[:, 2]
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    res = []
    for R, tt in ((R1, t), (R2, t), (R1, -t), (R2, -t)):
        P = np.hstack([R, tt[:, None]])
        m = np.zeros(len(a), bool)
        for i in range(len(a)):
            A = np.array([a[i, 0] * P0[2] - P0[0], a[i, 1] * P0[2] - P0[1], b[i, 0] * P[2] - P[0], b[i, 1] * P[2] - P[1]])
            Q = np.linalg.svd(A)[2][3]
            ok = Q[2] * Q[3] > 0
            X = Q[:3] / Q[3]
            ok = ok and X[2] < 50
            z2 = P[2, :3] @ X + P[2, 3]
            m[i] = ok and z2 > 0 and z2 < 50 and mask[i]
        res.append((int(m.sum()), R, tt, m))
    g = [r[0] for r in res]
    if g[0] >= g[1] and g[0] >= g[2] and g[0] >= g[3]: k = 0
    elif g[1] >= g[0] and g[1] >= g[2] and g[1] >= g[3]: k = 1
    elif g[2] >= g[0] and g[2] >= g[1] and g[2] >= g[3]: k = 2
    else: k = 3
    return res[k]


def make_ransac():
    rng = np.random.default_rng(515)
    out = {}
    K4 = np.array(synth.FOUNTAIN_K4, np.float32)
    q1s, q2s, Es = [], [], []
    for i in range(10):
        R = synth.aa_to_R(rng.normal(0, 0.3, 3)); t = rng.normal(0, 1, 3); t /= np.linalg.norm(t)
        X = rng.uniform(-2, 2, (5, 3)) + np.array([0, 0, 6.0])
        Xc = X @ R.T + t
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Egt = tx @ R
        q1s.append(X[:, :2] / X[:, 2:3]); q2s.append(Xc[:, :2] / Xc[:, 2:3]); Es.append(Egt / np.linalg.norm(Egt))
    out["five.q1"] = np.array(q1s); out["five.q2"] = np.array(q2s); out["five.E"] = np.array(Es)
    for tag, n, frac in (("pose_a", 200, 0.2), ("pose_b", 60, 0.5)):
        R = synth.aa_to_R(rng.normal(0, 0.2, 3)); t = np.array([1.0, 0.2, -0.1]); t /= np.linalg.norm(t)
        X = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0, 8.0])
        Xc = X @ R.T + t
        p1 = (X[:, :2] / X[:, 2:3] * [K4[0], K4[2]] + [K4[1], K4[3]]).astype(np.float32)
        p2 = (Xc[:, :2] / Xc[:, 2:3] * [K4[0], K4[2]] + [K4[1], K4[3]] + rng.normal(0, 0.2, (n, 2))).astype(np.float32)
        bad = rng.choice(n, int(frac * n), replace=False)
        p2[bad] += rng.uniform(-50, 50, (len(bad), 2)).astype(np.float32)
        mask = np.ones(n, bool); mask[bad[: len(bad) // 2]] = False
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        E = tx @ R; E /= np.linalg.norm(E)
        E = E + rng.normal(0, 1e-4, (3, 3))                      # a slightly perturbed (not exactly essential) matrix, like RANSAC's
        good, Rr, tr, m = recover_pose_numpy(E, p1, p2, K4, mask)
        out[f"{tag}.E"] = E; out[f"{tag}.p1"] = p1; out[f"{tag}.p2"] = p2; out[f"{tag}.mask_in"] = mask
        out[f"{tag}.R"] = Rr; out[f"{tag}.t"] = tr; out[f"{tag}.mask"] = m; out[f"{tag}.good"] = np.int32(good)
        print("ransac", tag, "good", good, "of", n)
    out["K4"] = K4
    np.savez_compressed(os.path.join(HERE, "ransac_cases.npz"), **out)


def undistort_numpy(img, K4, dist):
    """cv::undistort (estimate_motion.cpp:436) restated with whole-array numpy operations, independently of oracle/undistort_ref.c:
    per stripe the inverse of the shifted camera matrix by cofactors, x numerators by a running sum along the row, the k1 k2 p1
    p2 model in float64, positions rounded (half to even) to 1/32 pixel, integer bilinear weights, zero border."""
    img = np.asarray(img, np.uint8)
    rows, cols = img.shape[:2]
    src = img.reshape(rows, cols, -1).astype(np.int64)
    fx, u0, fy, v0 = [float(v) for v in K4]
    k1, k2, p1, p2 = [float(v) for v in dist]
    stripe0 = min(max(1, 4096 // cols), rows)
    out = np.zeros_like(src)
    padded = np.zeros((rows + 2, cols + 2, src.shape[2]), np.int64)      # zero border, index shifted by one
    padded[1:-1, 1:-1] = src
    for y0 in range(0, rows, stripe0):
        sr = min(stripe0, rows - y0)
        S = np.array([[fx, 0.0, u0], [0.0, fy, v0 - y0], [0.0, 0.0, 1.0]])
        det = S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0]) \
            + S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0])
        d = 1.0 / det
        cof = lambda a, b, c, e: (S[a] * S[b] - S[c] * S[e]) * d
        ir = np.array([cof((1, 1), (2, 2), (1, 2), (2, 1)), cof((0, 2), (2, 1), (0, 1), (2, 2)), cof((0, 1), (1, 2), (0, 2), (1, 1)),
                       cof((1, 2), (2, 0), (1, 0), (2, 2)), cof((0, 0), (2, 2), (0, 2), (2, 0)), cof((0, 2), (1, 0), (0, 0), (1, 2)),
                       cof((1, 0), (2, 1), (1, 1), (2, 0)), cof((0, 1), (2, 0), (0, 0), (2, 1)), cof((0, 0), (1, 1), (0, 1), (1, 0))])
        i = np.arange(sr, dtype=np.float64)[:, None]
        run = lambda first, inc: np.cumsum(np.concatenate([first, np.broadcast_to(inc, (sr, cols - 1))], axis=1), axis=1)
        X = run(i * ir[1] + ir[2], ir[0]); Y = run(i * ir[4] + ir[5], ir[3]); W = run(i * ir[7] + ir[8], ir[6])
        w = 1.0 / W; x = X * w; y = Y * w
        x2 = x * x; y2 = y * y; r2 = x2 + y2; xy2 = 2 * x * y
        kr = (1 + (k2 * r2 + k1) * r2) / 1.0
        xd = x * kr + p1 * xy2 + p2 * (r2 + 2 * x2)
        yd = y * kr + p1 * (r2 + 2 * y2) + p2 * xy2
        iu = np.rint((fx * xd + u0) * 32).astype(np.int64); iv = np.rint((fy * yd + v0) * 32).astype(np.int64)
        sx = iu >> 5; sy = iv >> 5; ax = iu & 31; ay = iv & 31
        outside = (sx >= cols) | (sx + 1 < 0) | (sy >= rows) | (sy + 1 < 0)
        cx = np.clip(sx, -1, cols - 1) + 1; cy = np.clip(sy, -1, rows - 1) + 1      # into the padded image
        acc = np.zeros((sr, cols, src.shape[2]), np.int64)
        for dy, dx, wgt in ((0, 0, (32 - ax) * (32 - ay)), (0, 1, ax * (32 - ay)), (1, 0, (32 - ax) * ay), (1, 1, ax * ay)):
            acc += padded[cy + dy, cx + dx] * (wgt * 32)[..., None]
        res = (acc + 16384) >> 15
        res[outside] = 0
        out[y0:y0 + sr] = res
    return out.reshape(img.shape).astype(np.uint8)


def make_undistort():
    """Golden undistortion cases: a colour and a gray image, mild and strong coefficients, a wide image (one-row stripes) and
    the coefficients the reference's importDistort quirk produces from a typical file (float pairs read as doubles)."""
    rng = np.random.default_rng(np.random.PCG64(97))
    out = {}
    yy, xx = np.mgrid[0:96, 0:160]
    smooth = (127 + 80 * np.sin(xx / 7.0) * np.cos(yy / 5.0)).astype(np.uint8)
    color = np.stack([smooth, np.roll(smooth, 5, 1), 255 - smooth], axis=2) ^ rng.integers(0, 8, (96, 160, 3), dtype=np.uint8)
    wide = rng.integers(0, 256, (6, 4500), dtype=np.uint8)
    quirk = np.zeros(4, np.float64)
    quirk.view(np.float32)[:4] = np.array([-0.28, 1.35, 0.0007, -0.0004], np.float32)     # k1 k2 p1 p2 written through at<float>
    cases = [
        ("color_mild", color, [140.0, 79.4, 138.0, 47.6], [-0.12, 0.03, 0.001, -0.0005]),
        ("color_strong", color, [90.0, 81.0, 92.0, 50.0], [0.45, -0.2, 0.01, 0.02]),
        ("gray_barrel", smooth, [120.0, 80.0, 120.0, 48.0], [-0.35, 0.12, 0.0, 0.0]),
        ("gray_zero", smooth, [131.7, 77.77, 129.3, 45.21], [0.0, 0.0, 0.0, 0.0]),
        ("wide_rows", wide, [2200.0, 2250.5, 2200.0, 2.5], [-0.08, 0.01, 0.0, 0.0]),
        ("quirk_coeffs", color, [140.0, 79.4, 138.0, 47.6], quirk),
    ]
    for name, img, K4, dist in cases:
        out[name + "_image"] = img
        out[name + "_K4"] = np.asarray(K4, np.float64)
        out[name + "_dist"] = np.asarray(dist, np.float64)
        out[name + "_out"] = undistort_numpy(img, K4, dist)
    out["names"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "undistort_cases.npz"), **out)


if __name__ == "__main__":
    make_hamming()
    make_l2()
    make_ba_jacobians()
    make_ba_trace()
    make_ba_constrained()
    make_sor()
    make_triangulation()
    make_ransac()
    make_undistort()
    print("golden vectors written to", HERE)

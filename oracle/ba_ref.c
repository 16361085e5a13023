/*
 * oracle/ba_ref.c -- CPU restatement of the reference's bundle adjustment.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include, link,
 * call or execute this file (see oracle/match_ref.c header).
 *
 * PARITY UNPINNED: the reference holds no BA tests/fixtures and cannot be
 * built here; the arithmetic lives in un-vendored Ceres (version unpinned,
 * cpp_code/CMakeLists.txt:45; a 2019 checkout, i.e. Ceres 1.14-era).  This file
 * restates
 *   - the cost functor ReprojectErrorTerm_fixcalib::operator()
 *     (cpp_code/include/ba.h:113-153) evaluated with forward-mode dual numbers
 *     exactly as ceres::AutoDiffCostFunction<...,2,6,3> (ba.h:155-159) would,
 *     including ceres::AngleAxisRotatePoint's two branches [upstream rotation.h];
 *   - the problem set-up of BundleAdjustment::solveBA (cpp_code/src/ba.cpp:140-151):
 *     one residual block + CauchyLoss(0.5) per observation; with a reference frame its six
 *     parameters are bounded to +-1e-10 (ba.cpp:155-162); with ba_calib_change_tolerance != 0 the
 *     functor is ReprojectErrorTerm_updatecalib (ba.h:170-222) over a shared intrinsics block
 *     bounded to its initial value +- tolerance (ba.cpp:167-196) -- esfm_ref_ba_solve_ex;
 *   - the solver options ba.cpp:201-204 (DENSE_SCHUR, 50 iterations) on top of
 *     Ceres defaults [upstream, from memory of ceres 1.14
 *     trust_region_minimizer.cc / levenberg_marquardt_strategy.cc /
 *     schur_eliminator_impl.h / corrector.cc / loss_function.cc]:
 *       cost = 1/2 sum rho(|r|^2); CauchyLoss: rho = b log(1+s/b), b = a^2;
 *       corrector with rho'' <= 0: r and J scaled by sqrt(rho');
 *       Jacobi scaling 1/(1+||col||) fixed at iteration 0;
 *       LM: D^2 = clamp(diag(J'J), 1e-6, 1e32)/radius, step = -(J'J+D^2)^-1 J'r
 *       through point-block Schur elimination and a dense Cholesky of the reduced
 *       camera system; model_cost_change = -(J s).(r + J s/2);
 *       parameter / function tolerance tested on the candidate BEFORE acceptance
 *       (a step that triggers them is NOT applied); accept iff
 *       relative_decrease > 1e-3; radius /= max(1/3, 1-(2q-1)^3) on accept,
 *       radius /= nu, nu *= 2 on reject; gradient tolerance after accepted steps.
 *   - parameter blocks with no observation are removed from the problem (Ceres
 *     drops unused blocks): untouched and excluded from the norms.
 *
 * The parameter layout is the reference's parameters_ array split in two:
 * cams = 6 doubles per camera (angle-axis, translation; ba.cpp:88-93),
 * pts = 3 per point (ba.cpp:101-103); observations are float pixels
 * (points_2d_, ba.cpp:37) and the four used K entries are float (ba.h:142-143).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/esfm.h"

/* ------------------------------------------------------------------------- */
/* forward-mode dual numbers with 13 partials (6 camera + 3 point + 4 intrinsics), ceres::Jet.
 * The fixed-intrinsics functor uses the first 9 (AutoDiffCostFunction<...,2,6,3>, ba.h:155-159), the
 * free-intrinsics functor all 13 (AutoDiffCostFunction<...,2,6,3,4>, ba.h:218-222). */
#define NJ 13
typedef struct { double a; double v[NJ]; } jet;

static inline jet jconst(double a) { jet r; r.a = a; for (int i = 0; i < NJ; ++i) r.v[i] = 0.0; return r; }
static inline jet jvar(double a, int k) { jet r = jconst(a); r.v[k] = 1.0; return r; }
static inline jet jadd(jet x, jet y) { jet r; r.a = x.a + y.a; for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
static inline jet jsub(jet x, jet y) { jet r; r.a = x.a - y.a; for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
static inline jet jmul(jet x, jet y) { jet r; r.a = x.a * y.a; for (int i = 0; i < NJ; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
/* ceres jet.h: h = f/g, h' = (f' - h g') / g */
static inline jet jdiv(jet x, jet y) { jet r; double gi = 1.0 / y.a; r.a = x.a * gi; for (int i = 0; i < NJ; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * gi; return r; }
static inline jet jsqrt(jet x) { jet r; r.a = sqrt(x.a); double t = 1.0 / (2.0 * r.a); for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] * t; return r; }
static inline jet jcos(jet x) { jet r; r.a = cos(x.a); double t = -sin(x.a); for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] * t; return r; }
static inline jet jsin(jet x) { jet r; r.a = sin(x.a); double t = cos(x.a); for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] * t; return r; }

/* ceres::AngleAxisRotatePoint [upstream rotation.h], called at ba.h:131. */
static void jet_angle_axis_rotate(const jet aa[3], const jet pt[3], jet out[3])
{
    jet theta2 = jadd(jadd(jmul(aa[0], aa[0]), jmul(aa[1], aa[1])), jmul(aa[2], aa[2]));
    if (theta2.a > DBL_EPSILON) {
        jet theta = jsqrt(theta2);
        jet costheta = jcos(theta);
        jet sintheta = jsin(theta);
        jet theta_inverse = jdiv(jconst(1.0), theta);
        jet w[3] = { jmul(aa[0], theta_inverse), jmul(aa[1], theta_inverse), jmul(aa[2], theta_inverse) };
        jet wxp[3] = { jsub(jmul(w[1], pt[2]), jmul(w[2], pt[1])),
                       jsub(jmul(w[2], pt[0]), jmul(w[0], pt[2])),
                       jsub(jmul(w[0], pt[1]), jmul(w[1], pt[0])) };
        jet tmp = jmul(jadd(jadd(jmul(w[0], pt[0]), jmul(w[1], pt[1])), jmul(w[2], pt[2])),
                       jsub(jconst(1.0), costheta));
        for (int k = 0; k < 3; ++k)
            out[k] = jadd(jadd(jmul(pt[k], costheta), jmul(wxp[k], sintheta)), jmul(w[k], tmp));
    } else {
        jet wxp[3] = { jsub(jmul(aa[1], pt[2]), jmul(aa[2], pt[1])),
                       jsub(jmul(aa[2], pt[0]), jmul(aa[0], pt[2])),
                       jsub(jmul(aa[0], pt[1]), jmul(aa[1], pt[0])) };
        for (int k = 0; k < 3; ++k) out[k] = jadd(pt[k], wxp[k]);
    }
}

/* ReprojectErrorTerm_fixcalib::operator() (ba.h:113-153) and ReprojectErrorTerm_updatecalib::operator()
 * (ba.h:170-216) on jets: the two functors differ only in where fx, cx, fy, cy come from -- constants taken from
 * the float calibration matrix (ba.h:142-143), or the shared 4-parameter block cam_calib = fx, cx, fy, cy
 * (ba.h:199-202).  cam[6] = angle-axis, translation; uv = observed pixel (float, ba.h:145-146 / 208-209).
 * r[2], Jc[2][6], Jp[2][3], Jk[2][4] (Jk only when calib != NULL). */
static void residual_jac_any(const double *cam, const double *pt, const float *K4, const double *calib,
                             const float *uv, double *r, double *Jc, double *Jp, double *Jk)
{
    jet aa[3], tr[3], X[3], p[3], in[4];
    for (int k = 0; k < 3; ++k) { aa[k] = jvar(cam[k], k); tr[k] = jvar(cam[3 + k], 3 + k); X[k] = jvar(pt[k], 6 + k); }
    for (int k = 0; k < 4; ++k) in[k] = calib ? jvar(calib[k], 9 + k) : jconst((double)K4[k]);
    jet_angle_axis_rotate(aa, X, p);
    for (int k = 0; k < 3; ++k) p[k] = jadd(p[k], tr[k]);
    jet x = jdiv(p[0], p[2]);
    jet y = jdiv(p[1], p[2]);
    jet u = jadd(jmul(x, in[0]), in[1]);
    jet v = jadd(jmul(y, in[2]), in[3]);
    jet r0 = jsub(jconst((double)uv[0]), u);
    jet r1 = jsub(jconst((double)uv[1]), v);
    r[0] = r0.a; r[1] = r1.a;
    if (Jc) for (int k = 0; k < 6; ++k) { Jc[k] = r0.v[k]; Jc[6 + k] = r1.v[k]; }
    if (Jp) for (int k = 0; k < 3; ++k) { Jp[k] = r0.v[6 + k]; Jp[3 + k] = r1.v[6 + k]; }
    if (Jk && calib) for (int k = 0; k < 4; ++k) { Jk[k] = r0.v[9 + k]; Jk[4 + k] = r1.v[9 + k]; }
}

void esfm_ref_ba_residual_jac(const double *cam, const double *pt, const float *K4, const float *uv,
                              double *r, double *Jc, double *Jp)
{
    residual_jac_any(cam, pt, K4, NULL, uv, r, Jc, Jp, NULL);
}

/* free-intrinsics functor (ba.h:170-222): calib = fx, cx, fy, cy doubles; Jk[2][4] */
void esfm_ref_ba_residual_jac_calib(const double *cam, const double *pt, const double *calib, const float *uv,
                                    double *r, double *Jc, double *Jp, double *Jk)
{
    residual_jac_any(cam, pt, NULL, calib, uv, r, Jc, Jp, Jk);
}

/* Residual only (candidate-cost evaluation), plain doubles, same formulas. */
static void residual_only(const double *cam, const double *pt, const double *in4, const float *uv, double *r)
{
    double theta2 = cam[0] * cam[0] + cam[1] * cam[1] + cam[2] * cam[2];
    double p[3];
    if (theta2 > DBL_EPSILON) {
        double theta = sqrt(theta2), c = cos(theta), s = sin(theta), ti = 1.0 / theta;
        double w[3] = { cam[0] * ti, cam[1] * ti, cam[2] * ti };
        double wxp[3] = { w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0] };
        double tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (1.0 - c);
        for (int k = 0; k < 3; ++k) p[k] = pt[k] * c + wxp[k] * s + w[k] * tmp;
    } else {
        double wxp[3] = { cam[1] * pt[2] - cam[2] * pt[1], cam[2] * pt[0] - cam[0] * pt[2], cam[0] * pt[1] - cam[1] * pt[0] };
        for (int k = 0; k < 3; ++k) p[k] = pt[k] + wxp[k];
    }
    p[0] += cam[3]; p[1] += cam[4]; p[2] += cam[5];
    double x = p[0] / p[2], y = p[1] / p[2];
    r[0] = (double)uv[0] - (x * in4[0] + in4[1]);
    r[1] = (double)uv[1] - (y * in4[2] + in4[3]);
}

/* ceres::CauchyLoss::Evaluate [upstream loss_function.cc]; a <= 0: trivial loss. */
static inline void loss_eval(double a, double s, double rho[3])
{
    if (a <= 0.0) { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; return; }
    double b = a * a, c = 1.0 / b;
    double sum = 1.0 + s * c, inv = 1.0 / sum;
    rho[0] = b * log(sum);
    rho[1] = inv > DBL_MIN ? inv : DBL_MIN;
    rho[2] = -c * (inv * inv);
}

/* Robustified cost 1/2 sum rho(|r|^2) (ceres ResidualBlock::Evaluate). Returns
 * DBL_MAX when any residual is non-finite (Ceres: evaluation fails -> candidate
 * cost = max double, trust_region_minimizer.cc ComputeCandidatePointAndEvaluateCost). */
static double cost_any(int n_obs, const int32_t *cam_idx, const int32_t *pt_idx, const float *obs_uv,
                       const float *K4, const double *calib, const double *cams, const double *pts, double cauchy_a)
{
    double cost = 0.0; int bad = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : cost) reduction(| : bad) schedule(static)
#endif
    for (int k = 0; k < n_obs; ++k) {
        double r[2], rho[3], in4[4];
        if (calib) { in4[0] = calib[0]; in4[1] = calib[1]; in4[2] = calib[2]; in4[3] = calib[3]; }
        else { const float *K = K4 + 4 * (size_t)cam_idx[k]; in4[0] = (double)K[0]; in4[1] = (double)K[1]; in4[2] = (double)K[2]; in4[3] = (double)K[3]; }
        residual_only(cams + 6 * (size_t)cam_idx[k], pts + 3 * (size_t)pt_idx[k], in4, obs_uv + 2 * (size_t)k, r);
        double s = r[0] * r[0] + r[1] * r[1];
        if (!isfinite(s)) { bad |= 1; continue; }
        loss_eval(cauchy_a, s, rho);
        cost += 0.5 * rho[0];
    }
    return bad ? DBL_MAX : cost;
}

double esfm_ref_ba_cost(int n_obs, const int32_t *cam_idx, const int32_t *pt_idx, const float *obs_uv,
                        const float *K4, const double *cams, const double *pts, double cauchy_a)
{
    return cost_any(n_obs, cam_idx, pt_idx, obs_uv, K4, NULL, cams, pts, cauchy_a);
}

double esfm_ref_ba_cost_calib(int n_obs, const int32_t *cam_idx, const int32_t *pt_idx, const float *obs_uv,
                              const double *calib, const double *cams, const double *pts, double cauchy_a)
{
    return cost_any(n_obs, cam_idx, pt_idx, obs_uv, NULL, calib, cams, pts, cauchy_a);
}

/* ------------------------------------------------------------------------- */
/* The unknowns are split as Ceres' Schur ordering splits them: e-blocks = points, f-blocks = cameras followed,
 * when the intrinsics are free (ba.cpp:169-196), by the shared 4-parameter block fx, cx, fy, cy.  All "f" arrays
 * below have nf = 6 n_cam + 4 has_calib entries: camera c at [6c, 6c+6), intrinsics at [6 n_cam, 6 n_cam + 4). */
typedef struct {
    int n_cam, n_pt, n_obs;
    int has_calib, nf;
    const int32_t *cam_idx, *pt_idx; const float *uv, *K4;
    /* observations grouped by point (CSR) */
    int32_t *pt_start; /* n_pt+1 */
    int32_t *order;    /* n_obs: original observation index, grouped by point */
    int32_t *cam_nobs;
    /* linearisation at x (corrected by the loss, columns scaled) */
    double *Jc;  /* 12 per obs, [2][6] */
    double *Jp;  /* 6 per obs, [2][3]  */
    double *Jk;  /* 8 per obs, [2][4]; intrinsics columns (has_calib only) */
    double *r;   /* 2 per obs */
    double *scale_c, *scale_p; /* Jacobi scaling, nf, 3*n_pt */
    double *diag_c, *diag_p;   /* LM diagonal (squared column norms, clamped) */
    double *grad_c, *grad_p;   /* gradient J'r of the UNSCALED problem at the last linearisation */
} ba_state;

/* ParameterBlock::Plus tail [upstream parameter_block.h]: project onto the box, lower bound first. */
static inline double clamp_box(double v, const double *lo, const double *up, size_t i)
{
    if (lo) v = fmax(v, lo[i]);
    if (up) v = fmin(v, up[i]);
    return v;
}

/* Evaluate residuals + Jacobians at (xf, pts), apply the corrector
 * (corrector.cc: rho''<=0 -> scale both by sqrt(rho')), return cost; gradient
 * max-norm of the UNSCALED problem in *gmax (trust_region_minimizer.cc
 * EvaluateGradientAndJacobian: gradient before column scaling; with bounds the
 * norm of x - Plus(x, -gradient), i.e. the projected gradient step). */
static double linearize(ba_state *S, const double *xf, const double *pts, double cauchy_a,
                        int apply_scaling, const double *lo, const double *up, double *gmax_out, int *ok)
{
    double cost = 0.0; int bad = 0;
    const double *calib = S->has_calib ? xf + 6 * (size_t)S->n_cam : NULL;
    double *gc = S->grad_c, *gp = S->grad_p;
    memset(gc, 0, sizeof(double) * (size_t)S->nf);
    memset(gp, 0, sizeof(double) * 3 * (size_t)S->n_pt);
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : cost) reduction(| : bad) schedule(static)
#endif
    for (int k = 0; k < S->n_obs; ++k) {
        double r[2], Jc[12], Jp[6], Jk[8], rho[3];
        int c = S->cam_idx[k], p = S->pt_idx[k];
        residual_jac_any(xf + 6 * (size_t)c, pts + 3 * (size_t)p, S->K4 ? S->K4 + 4 * (size_t)c : NULL, calib,
                         S->uv + 2 * (size_t)k, r, Jc, Jp, Jk);
        double s = r[0] * r[0] + r[1] * r[1];
        int fin = isfinite(s);
        for (int i = 0; i < 12; ++i) fin &= isfinite(Jc[i]);
        for (int i = 0; i < 6; ++i) fin &= isfinite(Jp[i]);
        if (calib) for (int i = 0; i < 8; ++i) fin &= isfinite(Jk[i]);
        if (!fin) { bad |= 1; continue; }
        loss_eval(cauchy_a, s, rho);
        cost += 0.5 * rho[0];
        double sq = sqrt(rho[1]);
        for (int i = 0; i < 12; ++i) S->Jc[12 * (size_t)k + i] = Jc[i] * sq;
        for (int i = 0; i < 6; ++i) S->Jp[6 * (size_t)k + i] = Jp[i] * sq;
        if (calib) for (int i = 0; i < 8; ++i) S->Jk[8 * (size_t)k + i] = Jk[i] * sq;
        S->r[2 * (size_t)k] = r[0] * sq; S->r[2 * (size_t)k + 1] = r[1] * sq;
    }
    *ok = !bad;
    if (bad) return DBL_MAX;
    /* gradient g = J' r (serial: deterministic order) */
    for (int k = 0; k < S->n_obs; ++k) {
        const double *Jc = S->Jc + 12 * (size_t)k, *Jp = S->Jp + 6 * (size_t)k, *r = S->r + 2 * (size_t)k;
        int c = S->cam_idx[k], p = S->pt_idx[k];
        for (int i = 0; i < 6; ++i) gc[6 * (size_t)c + i] += Jc[i] * r[0] + Jc[6 + i] * r[1];
        for (int i = 0; i < 3; ++i) gp[3 * (size_t)p + i] += Jp[i] * r[0] + Jp[3 + i] * r[1];
        if (calib) {
            const double *Jk = S->Jk + 8 * (size_t)k;
            for (int i = 0; i < 4; ++i) gc[6 * (size_t)S->n_cam + i] += Jk[i] * r[0] + Jk[4 + i] * r[1];
        }
    }
    double gmax = 0.0;
    for (size_t i = 0; i < (size_t)S->nf; ++i) {
        double a = (lo || up) ? fabs(xf[i] - clamp_box(xf[i] - gc[i], lo, up, i)) : fabs(gc[i]);
        if (a > gmax) gmax = a;
    }
    for (size_t i = 0; i < (size_t)3 * S->n_pt; ++i) if (fabs(gp[i]) > gmax) gmax = fabs(gp[i]);
    *gmax_out = gmax;
    if (apply_scaling) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
        for (int k = 0; k < S->n_obs; ++k) {
            int c = S->cam_idx[k], p = S->pt_idx[k];
            for (int row = 0; row < 2; ++row) {
                for (int i = 0; i < 6; ++i) S->Jc[12 * (size_t)k + 6 * row + i] *= S->scale_c[6 * (size_t)c + i];
                for (int i = 0; i < 3; ++i) S->Jp[6 * (size_t)k + 3 * row + i] *= S->scale_p[3 * (size_t)p + i];
                if (calib) for (int i = 0; i < 4; ++i) S->Jk[8 * (size_t)k + 4 * row + i] *= S->scale_c[6 * (size_t)S->n_cam + i];
            }
        }
    }
    return cost;
}

/* Squared column norms of the stored Jacobian. */
static void column_sqnorms(const ba_state *S, double *nc, double *np)
{
    memset(nc, 0, sizeof(double) * (size_t)S->nf);
    memset(np, 0, sizeof(double) * 3 * (size_t)S->n_pt);
    for (int k = 0; k < S->n_obs; ++k) {
        const double *Jc = S->Jc + 12 * (size_t)k, *Jp = S->Jp + 6 * (size_t)k;
        int c = S->cam_idx[k], p = S->pt_idx[k];
        for (int i = 0; i < 6; ++i) nc[6 * (size_t)c + i] += Jc[i] * Jc[i] + Jc[6 + i] * Jc[6 + i];
        for (int i = 0; i < 3; ++i) np[3 * (size_t)p + i] += Jp[i] * Jp[i] + Jp[3 + i] * Jp[3 + i];
        if (S->has_calib) {
            const double *Jk = S->Jk + 8 * (size_t)k;
            for (int i = 0; i < 4; ++i) nc[6 * (size_t)S->n_cam + i] += Jk[i] * Jk[i] + Jk[4 + i] * Jk[4 + i];
        }
    }
}

/* 3x3 symmetric positive definite inverse via Cholesky (ceres InvertPSDMatrix). */
static int inv3_spd(const double A[9], double Ai[9])
{
    double l00 = A[0]; if (!(l00 > 0.0)) return 0; l00 = sqrt(l00);
    double l10 = A[3] / l00, l20 = A[6] / l00;
    double l11 = A[4] - l10 * l10; if (!(l11 > 0.0)) return 0; l11 = sqrt(l11);
    double l21 = (A[7] - l20 * l10) / l11;
    double l22 = A[8] - l20 * l20 - l21 * l21; if (!(l22 > 0.0)) return 0; l22 = sqrt(l22);
    /* inverse of L */
    double m00 = 1.0 / l00, m11 = 1.0 / l11, m22 = 1.0 / l22;
    double m10 = -l10 * m00 * m11;
    double m21 = -l21 * m11 * m22;
    double m20 = -(l20 * m00 + l21 * m10) * m22;
    /* A^-1 = M' M */
    Ai[0] = m00 * m00 + m10 * m10 + m20 * m20;
    Ai[1] = Ai[3] = m10 * m11 + m20 * m21;
    Ai[2] = Ai[6] = m20 * m22;
    Ai[4] = m11 * m11 + m21 * m21;
    Ai[5] = Ai[7] = m21 * m22;
    Ai[8] = m22 * m22;
    return 1;
}

/* Dense Cholesky solve A x = b, A n x n symmetric (row-major, lower used),
 * in place on copies.  Returns 0 if A is not positive definite. */
static int chol_solve(double *A, double *b, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0) || !isfinite(d)) return 0;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (n - j > 256)
#endif
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    return 1;
}

/* The f-block columns one observation touches: its camera's 6 and, with free intrinsics, the shared 4.
 * h0/h1 = the two Jacobian rows restricted to those columns. */
static inline int obs_fcols(const ba_state *S, int k, int cols[10], double h0[10], double h1[10])
{
    const double *Jc = S->Jc + 12 * (size_t)k;
    const int c = S->cam_idx[k];
    for (int a = 0; a < 6; ++a) { cols[a] = 6 * c + a; h0[a] = Jc[a]; h1[a] = Jc[6 + a]; }
    if (!S->has_calib) return 6;
    const double *Jk = S->Jk + 8 * (size_t)k;
    for (int a = 0; a < 4; ++a) { cols[6 + a] = 6 * S->n_cam + a; h0[6 + a] = Jk[a]; h1[6 + a] = Jk[4 + a]; }
    return 10;
}

/*
 * Reduced f-block system of the damped normal equations (schur_eliminator_impl.h):
 *   S   = sum F'F + D_f^2 - sum_p W_p (E'E + D_p^2)^-1 W_p',  W_p = sum F_i'E_i
 *   rhs = sum F'r - sum_p W_p (E'E + D_p^2)^-1 E'r
 * over the observations of this state only (a shard builds a partial system;
 * D_f^2 is added iff add_cam_diag).  S is nf x nf row-major, full symmetric.
 * F_i is the observation's Jacobian over its f-block columns (camera, and the intrinsics when free).
 * EtEinv (9/pt) and Etr (3/pt) are kept for the back-substitution.
 */
static void reduced_point(const ba_state *S, int p, double radius, double *Sm, double *rhs,
                          double *EtEinv, double *Etr, int *ok)
{
    const int n = S->nf;
    int b = S->pt_start[p], e = S->pt_start[p + 1];
    if (b == e) return;
    double A[9] = {0}, g[3] = {0};
    for (int t = b; t < e; ++t) {
        int k = S->order[t];
        const double *Jp = S->Jp + 6 * (size_t)k, *r = S->r + 2 * (size_t)k;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) A[3 * i + j] += Jp[i] * Jp[j] + Jp[3 + i] * Jp[3 + j];
            g[i] += Jp[i] * r[0] + Jp[3 + i] * r[1];
        }
    }
    for (int i = 0; i < 3; ++i) A[4 * i] += S->diag_p[3 * (size_t)p + i] / radius;
    double Ai[9];
    if (!inv3_spd(A, Ai)) { *ok = 0; return; }
    memcpy(EtEinv + 9 * (size_t)p, Ai, sizeof(Ai));
    memcpy(Etr + 3 * (size_t)p, g, sizeof(g));
    double Aig[3];
    for (int i = 0; i < 3; ++i) Aig[i] = Ai[3 * i] * g[0] + Ai[3 * i + 1] * g[1] + Ai[3 * i + 2] * g[2];
    for (int t = b; t < e; ++t) {
        int k = S->order[t];
        const double *Jp = S->Jp + 6 * (size_t)k, *r = S->r + 2 * (size_t)k;
        int ci[10]; double h0[10], h1[10];
        const int m = obs_fcols(S, k, ci, h0, h1);
        /* W_i = F_i' E_i (m x 3); F'F and F'r */
        double W[30];
        for (int a = 0; a < m; ++a)
            for (int j = 0; j < 3; ++j) W[3 * a + j] = h0[a] * Jp[j] + h1[a] * Jp[3 + j];
        for (int a = 0; a < m; ++a) {
            for (int c2 = 0; c2 < m; ++c2)
                Sm[(size_t)ci[a] * n + ci[c2]] += h0[a] * h0[c2] + h1[a] * h1[c2];
            rhs[ci[a]] += h0[a] * r[0] + h1[a] * r[1];
            rhs[ci[a]] -= W[3 * a] * Aig[0] + W[3 * a + 1] * Aig[1] + W[3 * a + 2] * Aig[2];
        }
        /* Y = W_i Ai (m x 3) */
        double Y[30];
        for (int a = 0; a < m; ++a)
            for (int j = 0; j < 3; ++j)
                Y[3 * a + j] = W[3 * a] * Ai[j] + W[3 * a + 1] * Ai[3 + j] + W[3 * a + 2] * Ai[6 + j];
        for (int t2 = b; t2 < e; ++t2) {
            int k2 = S->order[t2];
            const double *Jp2 = S->Jp + 6 * (size_t)k2;
            int cj[10]; double g0[10], g1[10];
            const int m2 = obs_fcols(S, k2, cj, g0, g1);
            double W2[30]; /* W_j[c2][m] */
            for (int c2 = 0; c2 < m2; ++c2)
                for (int q = 0; q < 3; ++q) W2[3 * c2 + q] = g0[c2] * Jp2[q] + g1[c2] * Jp2[3 + q];
            for (int a = 0; a < m; ++a)
                for (int c2 = 0; c2 < m2; ++c2)
                    Sm[(size_t)ci[a] * n + cj[c2]] -=
                        Y[3 * a] * W2[3 * c2] + Y[3 * a + 1] * W2[3 * c2 + 1] + Y[3 * a + 2] * W2[3 * c2 + 2];
        }
    }
}

static int build_reduced(const ba_state *S, double radius, int add_cam_diag,
                         double *Sm, double *rhs, double *EtEinv, double *Etr)
{
    const int n = S->nf;
    const size_t nn = (size_t)n * n;
    memset(Sm, 0, sizeof(double) * nn);
    memset(rhs, 0, sizeof(double) * (size_t)n);
    int ok = 1;
    int nthr = 1;
#ifdef _OPENMP
    nthr = omp_get_max_threads();
    if (nn * sizeof(double) * (size_t)nthr > ((size_t)1 << 30)) nthr = 1; /* thread-local S copies <= 1 GiB */
#endif
    if (nthr <= 1) {
        for (int p = 0; p < S->n_pt; ++p) reduced_point(S, p, radius, Sm, rhs, EtEinv, Etr, &ok);
    } else {
#ifdef _OPENMP
        double *loc = (double *)calloc((nn + (size_t)n) * (size_t)nthr, sizeof(double));
#pragma omp parallel num_threads(nthr)
        {
            int tid = omp_get_thread_num(), myok = 1;
            double *lS = loc + (nn + (size_t)n) * (size_t)tid, *lr = lS + nn;
#pragma omp for schedule(static)
            for (int p = 0; p < S->n_pt; ++p) reduced_point(S, p, radius, lS, lr, EtEinv, Etr, &myok);
            if (!myok) {
#pragma omp atomic write
                ok = 0;
            }
        }
        for (int t = 0; t < nthr; ++t) { /* fixed thread order: deterministic for a fixed thread count */
            const double *lS = loc + (nn + (size_t)n) * (size_t)t, *lr = lS + nn;
            for (size_t i = 0; i < nn; ++i) Sm[i] += lS[i];
            for (int i = 0; i < n; ++i) rhs[i] += lr[i];
        }
        free(loc);
#endif
    }
    if (add_cam_diag)
        for (int i = 0; i < n; ++i) Sm[(size_t)i * n + i] += S->diag_c[i] / radius;
    return ok;
}

static void group_by_point(ba_state *S)
{
    S->pt_start = (int32_t *)calloc((size_t)S->n_pt + 1, sizeof(int32_t));
    S->order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S->n_obs > 0 ? S->n_obs : 1));
    S->cam_nobs = (int32_t *)calloc((size_t)S->n_cam > 0 ? S->n_cam : 1, sizeof(int32_t));
    for (int k = 0; k < S->n_obs; ++k) { S->pt_start[S->pt_idx[k] + 1]++; S->cam_nobs[S->cam_idx[k]]++; }
    for (int p = 0; p < S->n_pt; ++p) S->pt_start[p + 1] += S->pt_start[p];
    int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * ((size_t)S->n_pt + 1));
    memcpy(fill, S->pt_start, sizeof(int32_t) * ((size_t)S->n_pt + 1));
    for (int k = 0; k < S->n_obs; ++k) S->order[fill[S->pt_idx[k]]++] = k;
    free(fill);
}

static void state_alloc(ba_state *S)
{
    size_t no = S->n_obs > 0 ? S->n_obs : 1;
    S->nf = 6 * S->n_cam + (S->has_calib ? 4 : 0);
    size_t nf = S->nf > 0 ? (size_t)S->nf : 1, np3 = 3 * (size_t)(S->n_pt > 0 ? S->n_pt : 1);
    S->Jc = (double *)malloc(sizeof(double) * 12 * no);
    S->Jp = (double *)malloc(sizeof(double) * 6 * no);
    S->Jk = S->has_calib ? (double *)malloc(sizeof(double) * 8 * no) : NULL;
    S->r = (double *)malloc(sizeof(double) * 2 * no);
    S->scale_c = (double *)malloc(sizeof(double) * nf);
    S->scale_p = (double *)malloc(sizeof(double) * np3);
    S->diag_c = (double *)calloc(nf, sizeof(double));
    S->diag_p = (double *)calloc(np3, sizeof(double));
    S->grad_c = (double *)calloc(nf, sizeof(double));
    S->grad_p = (double *)calloc(np3, sizeof(double));
    for (size_t i = 0; i < nf; ++i) S->scale_c[i] = 1.0;
    for (size_t i = 0; i < np3; ++i) S->scale_p[i] = 1.0;
}

static void state_free(ba_state *S)
{
    free(S->pt_start); free(S->order); free(S->cam_nobs);
    free(S->Jc); free(S->Jp); free(S->Jk); free(S->r); free(S->scale_c); free(S->scale_p); free(S->diag_c); free(S->diag_p);
    free(S->grad_c); free(S->grad_p);
}

/* Exported for the sharding test (tests/test_sharding_gloo.py): partial
 * reduced system + LM diagonal pieces of an observation subset, linearised at
 * (cams, pts) with NO Jacobi scaling and a caller-supplied LM diagonal
 * (diag_c, diag_p = squared column norms, already clamped). */
int esfm_ref_ba_partial_reduced(int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx,
                                const float *obs_uv, const float *K4, const double *cams, const double *pts,
                                double cauchy_a, double radius, const double *diag_c, const double *diag_p,
                                int add_cam_diag, double *Sm, double *rhs)
{
    ba_state S; memset(&S, 0, sizeof(S));
    S.n_cam = n_cam; S.n_pt = n_pt; S.n_obs = n_obs; S.cam_idx = cam_idx; S.pt_idx = pt_idx; S.uv = obs_uv; S.K4 = K4;
    group_by_point(&S); state_alloc(&S);
    double gmax; int ok;
    linearize(&S, cams, pts, cauchy_a, 0, NULL, NULL, &gmax, &ok);
    memcpy(S.diag_c, diag_c, sizeof(double) * 6 * (size_t)n_cam);
    memcpy(S.diag_p, diag_p, sizeof(double) * 3 * (size_t)n_pt);
    double *EtEinv = (double *)malloc(sizeof(double) * 9 * (size_t)n_pt);
    double *Etr = (double *)malloc(sizeof(double) * 3 * (size_t)n_pt);
    int ok2 = ok && build_reduced(&S, radius, add_cam_diag, Sm, rhs, EtEinv, Etr);
    free(EtEinv); free(Etr); state_free(&S);
    return ok2 ? 0 : -1;
}

/* Squared column norms of the loss-corrected, unscaled Jacobian at (cams, pts)
 * (what the Jacobi scaling and the first LM diagonal are computed from). */
int esfm_ref_ba_column_sqnorms(int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx,
                               const float *obs_uv, const float *K4, const double *cams, const double *pts,
                               double cauchy_a, double *nc, double *np)
{
    ba_state S; memset(&S, 0, sizeof(S));
    S.n_cam = n_cam; S.n_pt = n_pt; S.n_obs = n_obs; S.cam_idx = cam_idx; S.pt_idx = pt_idx; S.uv = obs_uv; S.K4 = K4;
    group_by_point(&S); state_alloc(&S);
    double gmax; int ok;
    linearize(&S, cams, pts, cauchy_a, 0, NULL, NULL, &gmax, &ok);
    column_sqnorms(&S, nc, np);
    state_free(&S);
    return ok ? 0 : -1;
}

static double now_sec(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void esfm_ref_ba_options_default(esfm_ba_options *o)
{
    o->max_num_iterations = 50;                  /* ba.cpp:202 */
    o->jacobi_scaling = 1;
    o->max_num_consecutive_invalid_steps = 5;
    o->verbose = 0;
    o->cauchy_a = 0.5;                           /* ba.cpp:150 */
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
}

/* ------------------------------------------------------------------------- */
/* Armijo line search of the bounds-constrained trust-region loop [upstream line_search.cc ArmijoLineSearch::DoSearch,
 * polynomial.cc MinimizeInterpolatingPolynomial; Ceres defaults: CUBIC interpolation, sufficient decrease 1e-4,
 * step contraction within [1e-3, 0.6], min step size 1e-9, at most 20 iterations].
 * Polynomials are stored lowest degree first. */
typedef struct { double x, value, gradient; int value_valid, gradient_valid; } ls_sample;

static double poly_eval(const double *c, int deg, double x)
{
    double v = c[deg];
    for (int i = deg - 1; i >= 0; --i) v = v * x + c[i];
    return v;
}

/* solve the (n x n) system A c = b with full pivoting (FindInterpolatingPolynomial uses fullPivLu) */
static int solve_full_pivot(double *A, double *b, int n, double *x)
{
    int perm[8];
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int pr = k, pc = k; double best = 0.0;
        for (int i = k; i < n; ++i) for (int j = k; j < n; ++j) if (fabs(A[i * n + j]) > best) { best = fabs(A[i * n + j]); pr = i; pc = j; }
        if (best == 0.0) return 0;
        if (pr != k) { for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[pr * n + j]; A[pr * n + j] = t; } double t = b[k]; b[k] = b[pr]; b[pr] = t; }
        if (pc != k) { for (int i = 0; i < n; ++i) { double t = A[i * n + k]; A[i * n + k] = A[i * n + pc]; A[i * n + pc] = t; } int t = perm[k]; perm[k] = perm[pc]; perm[pc] = t; }
        for (int i = k + 1; i < n; ++i) {
            double f = A[i * n + k] / A[k * n + k];
            for (int j = k; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
            b[i] -= f * b[k];
        }
    }
    double y[8];
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < n; ++j) s -= A[i * n + j] * y[j];
        y[i] = s / A[i * n + i];
    }
    for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
    return 1;
}

/* polynomial through the samples' values (and gradients where valid); returns the degree */
static int fit_polynomial(const ls_sample *smp, int ns, double *coef)
{
    int ncon = 0;
    for (int i = 0; i < ns; ++i) ncon += (smp[i].value_valid ? 1 : 0) + (smp[i].gradient_valid ? 1 : 0);
    const int n = ncon;
    double A[64], b[8];
    int row = 0;
    for (int i = 0; i < ns; ++i) {
        if (smp[i].value_valid) {
            double pw = 1.0;
            for (int j = 0; j < n; ++j) { A[row * n + j] = pw; pw *= smp[i].x; }
            b[row++] = smp[i].value;
        }
        if (smp[i].gradient_valid) {
            double pw = 1.0;
            A[row * n] = 0.0;
            for (int j = 1; j < n; ++j) { A[row * n + j] = (double)j * pw; pw *= smp[i].x; }
            b[row++] = smp[i].gradient;
        }
    }
    if (!solve_full_pivot(A, b, n, coef)) return -1;
    return n - 1;
}

/* real roots of q (degree deg, lowest first) inside [a, b]; returns the count.  Roots are isolated between
 * consecutive critical points and refined by bisection. */
static int real_roots_in(const double *q, int deg, double a, double b, double *roots)
{
    while (deg > 0 && q[deg] == 0.0) --deg;
    if (deg <= 0) return 0;
    if (deg == 1) { double r = -q[0] / q[1]; if (r >= a && r <= b) { roots[0] = r; return 1; } return 0; }
    double dq[8];
    for (int i = 1; i <= deg; ++i) dq[i - 1] = (double)i * q[i];
    double crit[8];
    int nc = real_roots_in(dq, deg - 1, a, b, crit);
    for (int i = 1; i < nc; ++i) { double v = crit[i]; int j = i - 1; while (j >= 0 && crit[j] > v) { crit[j + 1] = crit[j]; --j; } crit[j + 1] = v; }
    double brk[10]; int nb = 0;
    brk[nb++] = a; for (int i = 0; i < nc; ++i) brk[nb++] = crit[i]; brk[nb++] = b;
    int nr = 0;
    for (int i = 0; i + 1 < nb; ++i) {
        double l = brk[i], u = brk[i + 1];
        if (!(u > l)) continue;
        double fl = poly_eval(q, deg, l), fu = poly_eval(q, deg, u);
        if (fl == 0.0) { if (nr == 0 || roots[nr - 1] != l) roots[nr++] = l; continue; }
        if (fu == 0.0) { if (i + 2 == nb) roots[nr++] = u; continue; }
        if ((fl < 0.0) == (fu < 0.0)) continue;
        for (int it = 0; it < 200 && u - l > 0.0; ++it) {
            double mid = 0.5 * (l + u);
            if (mid == l || mid == u) break;
            double fm = poly_eval(q, deg, mid);
            if (fm == 0.0) { l = u = mid; break; }
            if ((fm < 0.0) == (fl < 0.0)) { l = mid; fl = fm; } else { u = mid; }
        }
        roots[nr++] = 0.5 * (l + u);
    }
    return nr;
}

/* polynomial.cc MinimizePolynomial: midpoint, the two ends, then the derivative's roots inside the interval.
 * For a quadratic derivative the candidates follow FindQuadraticPolynomialRoots (including the real part of a
 * complex pair, which Ceres also tries); for higher degrees only the real roots are tried. */
static double minimize_polynomial(const double *c, int deg, double xmin, double xmax)
{
    double best_x = (xmin + xmax) / 2.0, best_v = poly_eval(c, deg, best_x);
    double v = poly_eval(c, deg, xmin); if (v < best_v) { best_v = v; best_x = xmin; }
    v = poly_eval(c, deg, xmax); if (v < best_v) { best_v = v; best_x = xmax; }
    if (deg <= 1) return best_x;
    double d[8]; int dd = deg - 1;
    for (int i = 1; i <= deg; ++i) d[i - 1] = (double)i * c[i];
    while (dd > 0 && d[dd] == 0.0) --dd;
    double roots[8]; int nr = 0;
    if (dd == 1) { roots[nr++] = -d[0] / d[1]; }
    else if (dd == 2) {
        const double qa = d[2], qb = d[1], qc = d[0];
        const double D = qb * qb - 4.0 * qa * qc, sD = sqrt(fabs(D));
        if (D >= 0.0) {
            if (qb >= 0.0) { roots[nr++] = (-qb - sD) / (2.0 * qa); roots[nr++] = (2.0 * qc) / (-qb - sD); }
            else { roots[nr++] = (2.0 * qc) / (-qb + sD); roots[nr++] = (-qb + sD) / (2.0 * qa); }
        } else { roots[nr++] = -qb / (2.0 * qa); }
    } else if (dd > 2) nr = real_roots_in(d, dd, xmin, xmax, roots);
    for (int i = 0; i < nr; ++i) {
        if (!(roots[i] >= xmin && roots[i] <= xmax)) continue;
        v = poly_eval(c, deg, roots[i]);
        if (v < best_v) { best_v = v; best_x = roots[i]; }
    }
    return best_x;
}

/* line_search.cc InterpolatingPolynomialMinimizingStepSize, CUBIC */
static double ls_next_step(const ls_sample *initial, const ls_sample *previous, const ls_sample *current,
                           double min_step, double max_step)
{
    if (!current->value_valid || !current->gradient_valid)
        return fmin(fmax(current->x * 0.5, min_step), max_step);
    ls_sample smp[3]; int ns = 0;
    smp[ns++] = *initial; smp[ns++] = *current;
    if (previous->value_valid) smp[ns++] = *previous;
    double coef[8];
    int deg = fit_polynomial(smp, ns, coef);
    if (deg < 0) return fmin(fmax(current->x * 0.5, min_step), max_step);
    return minimize_polynomial(coef, deg, min_step, max_step);
}

/* exported so the product's host-side restatement of the same search can be checked on polynomials alone */
double esfm_ref_ls_next_step(double f0, double g0, double xp, double fp, double gp, int prev_valid,
                             double xc, double fc, double gc, int cur_valid, double min_step, double max_step)
{
    ls_sample ini = { 0.0, f0, g0, 1, 1 }, prev = { xp, fp, gp, prev_valid, prev_valid }, cur = { xc, fc, gc, cur_valid, cur_valid };
    return ls_next_step(&ini, &prev, &cur, min_step, max_step);
}

/* value and directional derivative of the cost along delta at Plus(x, t delta) [line_search.cc
 * LineSearchFunction::Evaluate: gradient of the objective at the projected point, dotted with the direction] */
static int ls_evaluate(const ba_state *S, const double *xf, const double *xp, const double *df, const double *dp,
                       const double *lo, const double *up, double t, double cauchy_a,
                       double *tf, double *tp, double *value, double *gradient)
{
    for (int i = 0; i < S->nf; ++i) tf[i] = clamp_box(xf[i] + t * df[i], lo, up, (size_t)i);
    for (size_t i = 0; i < 3 * (size_t)S->n_pt; ++i) tp[i] = xp[i] + t * dp[i];
    const double *calib = S->has_calib ? tf + 6 * (size_t)S->n_cam : NULL;
    double cost = 0.0, g = 0.0; int bad = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : cost, g) reduction(| : bad) schedule(static)
#endif
    for (int k = 0; k < S->n_obs; ++k) {
        double r[2], Jc[12], Jp[6], Jk[8], rho[3];
        int c = S->cam_idx[k], p = S->pt_idx[k];
        residual_jac_any(tf + 6 * (size_t)c, tp + 3 * (size_t)p, S->K4 ? S->K4 + 4 * (size_t)c : NULL, calib,
                         S->uv + 2 * (size_t)k, r, Jc, Jp, Jk);
        double s = r[0] * r[0] + r[1] * r[1];
        if (!isfinite(s)) { bad |= 1; continue; }
        loss_eval(cauchy_a, s, rho);
        cost += 0.5 * rho[0];
        double m0 = 0.0, m1 = 0.0;
        for (int a = 0; a < 6; ++a) { m0 += Jc[a] * df[6 * c + a]; m1 += Jc[6 + a] * df[6 * c + a]; }
        for (int a = 0; a < 3; ++a) { m0 += Jp[a] * dp[3 * (size_t)p + a]; m1 += Jp[3 + a] * dp[3 * (size_t)p + a]; }
        if (calib) for (int a = 0; a < 4; ++a) { m0 += Jk[a] * df[6 * S->n_cam + a]; m1 += Jk[4 + a] * df[6 * S->n_cam + a]; }
        g += rho[1] * (r[0] * m0 + r[1] * m1);
    }
    *value = cost; *gradient = g;
    return !bad && isfinite(cost) && isfinite(g);
}

/* ceres::Solve for the problem ba.cpp:140-196 builds (see header).
 *   calib      NULL: fixed intrinsics K4 per camera (ba.cpp:142-164)
 *              else: shared free intrinsics fx, cx, fy, cy (ba.cpp:167-196), in/out, bounded to the initial value
 *              +- calib_tol (ba.cpp:190-194; calib_tol <= 0 is rejected: lower >= upper is infeasible for Ceres)
 *   ref_cam    >= 0: every parameter of that camera is bounded to [-ref_thr, +ref_thr] (ba.cpp:155-162 / 181-188;
 *              the reference uses 1e-10), < 0: none
 * With any bound the trust-region loop is Ceres' constrained variant: x projected at iteration 0, projected
 * gradient norm, Plus() projects, and an Armijo line search along the LM step before the candidate is evaluated
 * (trust_region_minimizer.cc DoLineSearch). */
int esfm_ref_ba_solve_ex(int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx,
                         const float *obs_uv, const float *K4, double *cams, double *pts,
                         double *calib, double calib_tol, int ref_cam, double ref_thr,
                         const esfm_ba_options *opt_in, esfm_ba_summary *sum)
{
    esfm_ba_options opt;
    if (opt_in) opt = *opt_in; else esfm_ref_ba_options_default(&opt);
    esfm_ba_summary local; if (!sum) sum = &local;
    memset(sum, 0, sizeof(*sum));
    for (int k = 0; k < n_obs; ++k)
        if (cam_idx[k] < 0 || cam_idx[k] >= n_cam || pt_idx[k] < 0 || pt_idx[k] >= n_pt) return -1;
    if (calib && !(calib_tol > 0.0)) return -1;
    if (!calib && !K4) return -1;
    if (ref_cam >= n_cam) return -1;
    if (ref_cam >= 0 && !(ref_thr > 0.0)) return -1;

    ba_state S; memset(&S, 0, sizeof(S));
    S.n_cam = n_cam; S.n_pt = n_pt; S.n_obs = n_obs; S.cam_idx = cam_idx; S.pt_idx = pt_idx; S.uv = obs_uv;
    S.K4 = calib ? NULL : K4; S.has_calib = calib != NULL;
    group_by_point(&S); state_alloc(&S);
    const int n = S.nf;
    const size_t n1 = n > 0 ? (size_t)n : 1, np3 = 3 * (size_t)(n_pt > 0 ? n_pt : 1);
    int nact_c = 0, nact_p = 0;
    for (int c = 0; c < n_cam; ++c) nact_c += S.cam_nobs[c] > 0;
    for (int p = 0; p < n_pt; ++p) nact_p += S.pt_start[p + 1] > S.pt_start[p];
    sum->num_active_cameras = nact_c; sum->num_active_points = nact_p;

    double *x_c = (double *)malloc(sizeof(double) * n1); /* f-block unknowns: cameras | intrinsics */
    double *x_p = pts;                                   /* points live in the caller's array */
    memcpy(x_c, cams, sizeof(double) * 6 * (size_t)n_cam);
    if (calib) memcpy(x_c + 6 * (size_t)n_cam, calib, sizeof(double) * 4);
    char *act_f = (char *)calloc(n1, 1);
    for (int c = 0; c < n_cam; ++c) if (S.cam_nobs[c] > 0) for (int i = 0; i < 6; ++i) act_f[6 * c + i] = 1;
    if (calib && n_obs > 0) for (int i = 0; i < 4; ++i) act_f[6 * n_cam + i] = 1;
    /* bounds (only on active blocks: Ceres drops unused blocks together with their bounds) */
    double *lo = NULL, *up = NULL;
    int constrained = 0;
    if ((calib && n_obs > 0) || (ref_cam >= 0 && S.cam_nobs[ref_cam] > 0)) {
        constrained = 1;
        lo = (double *)malloc(sizeof(double) * n1); up = (double *)malloc(sizeof(double) * n1);
        for (int i = 0; i < n; ++i) { lo[i] = -INFINITY; up[i] = INFINITY; }
        if (ref_cam >= 0 && S.cam_nobs[ref_cam] > 0) for (int i = 0; i < 6; ++i) { lo[6 * ref_cam + i] = -ref_thr; up[6 * ref_cam + i] = ref_thr; }
        if (calib && n_obs > 0) for (int i = 0; i < 4; ++i) { lo[6 * n_cam + i] = calib[i] - calib_tol; up[6 * n_cam + i] = calib[i] + calib_tol; }
    }
    double *cand_c = (double *)malloc(sizeof(double) * n1);
    double *cand_p = (double *)malloc(sizeof(double) * np3);
    double *Sm = (double *)malloc(sizeof(double) * n1 * n1);
    double *rhs = (double *)malloc(sizeof(double) * n1);
    double *EtEinv = (double *)malloc(sizeof(double) * 3 * np3);
    double *Etr = (double *)malloc(sizeof(double) * np3);
    double *step_c = (double *)malloc(sizeof(double) * n1);
    double *step_p = (double *)malloc(sizeof(double) * np3);
    double *delta_c = (double *)malloc(sizeof(double) * n1);
    double *delta_p = (double *)malloc(sizeof(double) * np3);
    double *nc = (double *)malloc(sizeof(double) * n1);
    double *np = (double *)malloc(sizeof(double) * np3);

    double t0 = now_sec();
    int rc = 0;
    /* ---- iteration 0 (TrustRegionMinimizer::IterationZero) ---- */
    if (constrained) for (int i = 0; i < n; ++i) if (act_f[i]) x_c[i] = clamp_box(x_c[i], lo, up, (size_t)i);
    double x_norm = 0.0;
    for (int i = 0; i < n; ++i) if (act_f[i]) x_norm += x_c[i] * x_c[i];
    for (int p = 0; p < n_pt; ++p) if (S.pt_start[p + 1] > S.pt_start[p]) for (int i = 0; i < 3; ++i) x_norm += x_p[3 * (size_t)p + i] * x_p[3 * (size_t)p + i];
    x_norm = sqrt(x_norm);
    double gmax = 0.0; int ok = 1;
    double x_cost = linearize(&S, x_c, x_p, opt.cauchy_a, 0, lo, up, &gmax, &ok);
    if (!ok) { sum->termination = ESFM_BA_FAILURE; rc = -6; goto done; }
    if (opt.jacobi_scaling) {
        column_sqnorms(&S, nc, np);
        for (int i = 0; i < n; ++i) S.scale_c[i] = 1.0 / (1.0 + sqrt(nc[i]));
        for (size_t i = 0; i < 3 * (size_t)n_pt; ++i) S.scale_p[i] = 1.0 / (1.0 + sqrt(np[i]));
        for (int k = 0; k < n_obs; ++k) {
            int c = cam_idx[k], p = pt_idx[k];
            for (int row = 0; row < 2; ++row) {
                for (int i = 0; i < 6; ++i) S.Jc[12 * (size_t)k + 6 * row + i] *= S.scale_c[6 * (size_t)c + i];
                for (int i = 0; i < 3; ++i) S.Jp[6 * (size_t)k + 3 * row + i] *= S.scale_p[3 * (size_t)p + i];
                if (calib) for (int i = 0; i < 4; ++i) S.Jk[8 * (size_t)k + 4 * row + i] *= S.scale_c[6 * (size_t)n_cam + i];
            }
        }
    }
    double radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
    int reuse_diagonal = 0, n_invalid = 0;
    sum->initial_cost = x_cost;
    esfm_ba_iteration *it = &sum->iterations[0];
    it->iteration = 0; it->step_is_valid = 1; it->step_is_successful = 1; it->cost = x_cost;
    it->gradient_max_norm = gmax; it->trust_region_radius = radius;
    sum->num_iterations = 0; sum->num_successful_steps = 1;
    if (opt.verbose) printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d % .6e  % .2e  % .2e  % .2e  % .2e  % .2e\n", 0, x_cost, 0.0, gmax, 0.0, 0.0, radius);
    int terminated = 0;
    if (gmax <= opt.gradient_tolerance) { sum->termination = ESFM_BA_CONVERGENCE; terminated = 1; }

    /* ---- main loop (TrustRegionMinimizer::Minimize) ---- */
    int iter = 0;
    double last_gmax = gmax;
    while (!terminated) {
        if (iter >= opt.max_num_iterations) { sum->termination = ESFM_BA_NO_CONVERGENCE; break; }
        if (radius <= opt.min_trust_region_radius) { sum->termination = ESFM_BA_CONVERGENCE; break; }
        ++iter;
        esfm_ba_iteration cur; memset(&cur, 0, sizeof(cur));
        cur.iteration = iter; cur.gradient_max_norm = last_gmax;
        /* LevenbergMarquardtStrategy::ComputeStep */
        if (!reuse_diagonal) {
            column_sqnorms(&S, nc, np);
            for (int i = 0; i < n; ++i) S.diag_c[i] = fmin(fmax(nc[i], opt.min_lm_diagonal), opt.max_lm_diagonal);
            for (size_t i = 0; i < 3 * (size_t)n_pt; ++i) S.diag_p[i] = fmin(fmax(np[i], opt.min_lm_diagonal), opt.max_lm_diagonal);
        }
        int lin_ok = build_reduced(&S, radius, 1, Sm, rhs, EtEinv, Etr);
        /* inactive cameras: identity rows so the factorisation is defined; their step is 0 */
        for (int i = 0; i < n; ++i) if (!act_f[i]) rhs[i] = 0.0;
        if (lin_ok) lin_ok = chol_solve(Sm, rhs, n);
        if (lin_ok) {
            for (int i = 0; i < n; ++i) { step_c[i] = -rhs[i]; lin_ok &= isfinite(step_c[i]); }
            /* back-substitution: y_p = (E'E+D^2)^-1 (E'r - sum E'F y_f) */
            for (int p = 0; p < n_pt; ++p) {
                int b = S.pt_start[p], e = S.pt_start[p + 1];
                if (b == e) { step_p[3 * (size_t)p] = step_p[3 * (size_t)p + 1] = step_p[3 * (size_t)p + 2] = 0.0; continue; }
                double g[3] = { Etr[3 * (size_t)p], Etr[3 * (size_t)p + 1], Etr[3 * (size_t)p + 2] };
                for (int t = b; t < e; ++t) {
                    int k = S.order[t];
                    const double *Jp = S.Jp + 6 * (size_t)k;
                    int ci[10]; double h0[10], h1[10];
                    const int m = obs_fcols(&S, k, ci, h0, h1);
                    double f0 = 0.0, f1 = 0.0; /* F y_f */
                    for (int a = 0; a < m; ++a) { f0 += h0[a] * rhs[ci[a]]; f1 += h1[a] * rhs[ci[a]]; }
                    for (int i = 0; i < 3; ++i) g[i] -= Jp[i] * f0 + Jp[3 + i] * f1;
                }
                const double *Ai = EtEinv + 9 * (size_t)p;
                for (int i = 0; i < 3; ++i) {
                    double y = Ai[3 * i] * g[0] + Ai[3 * i + 1] * g[1] + Ai[3 * i + 2] * g[2];
                    step_p[3 * (size_t)p + i] = -y; lin_ok &= isfinite(y);
                }
            }
        }
        reuse_diagonal = 1;
        double model_cost_change = 0.0;
        if (lin_ok) {
            for (int k = 0; k < n_obs; ++k) {
                int p = pt_idx[k];
                const double *Jp = S.Jp + 6 * (size_t)k, *r = S.r + 2 * (size_t)k;
                int ci[10]; double h0[10], h1[10];
                const int m = obs_fcols(&S, k, ci, h0, h1);
                double m0 = 0.0, m1 = 0.0;
                for (int a = 0; a < m; ++a) { m0 += h0[a] * step_c[ci[a]]; m1 += h1[a] * step_c[ci[a]]; }
                for (int a = 0; a < 3; ++a) { m0 += Jp[a] * step_p[3 * (size_t)p + a]; m1 += Jp[3 + a] * step_p[3 * (size_t)p + a]; }
                model_cost_change -= m0 * (r[0] + m0 / 2.0) + m1 * (r[1] + m1 / 2.0);
            }
        }
        cur.model_cost_change = model_cost_change;
        cur.step_is_valid = lin_ok && (model_cost_change > 0.0);
        if (!cur.step_is_valid) {
            /* HandleInvalidStep */
            if (++n_invalid >= opt.max_num_consecutive_invalid_steps) { sum->termination = ESFM_BA_FAILURE; terminated = 1; }
            radius *= 0.5; reuse_diagonal = 1;
            cur.cost = x_cost; cur.trust_region_radius = radius;
            sum->num_unsuccessful_steps++;
            if (iter < ESFM_BA_MAX_LOG) sum->iterations[iter] = cur;
            sum->num_iterations = iter;
            continue;
        }
        n_invalid = 0;
        /* delta = step .* scaling (zero on blocks that are not part of the problem) */
        for (int i = 0; i < n; ++i) delta_c[i] = act_f[i] ? step_c[i] * S.scale_c[i] : 0.0;
        for (int p = 0; p < n_pt; ++p) {
            int active = S.pt_start[p + 1] > S.pt_start[p];
            for (int i = 0; i < 3; ++i) { size_t j = 3 * (size_t)p + i; delta_p[j] = active ? step_p[j] * S.scale_p[j] : 0.0; }
        }
        if (constrained) {
            /* TrustRegionMinimizer::DoLineSearch: Armijo search from step size 1 along delta */
            double g0 = 0.0, dmax = 0.0;
            for (int i = 0; i < n; ++i) { g0 += S.grad_c[i] * delta_c[i]; dmax = fmax(dmax, fabs(delta_c[i])); }
            for (size_t i = 0; i < 3 * (size_t)n_pt; ++i) { g0 += S.grad_p[i] * delta_p[i]; dmax = fmax(dmax, fabs(delta_p[i])); }
            ls_sample initial = { 0.0, x_cost, g0, 1, 1 }, previous = { 0.0, 0.0, 0.0, 0, 0 }, current = { 1.0, 0.0, 0.0, 0, 0 };
            current.value_valid = current.gradient_valid =
                ls_evaluate(&S, x_c, x_p, delta_c, delta_p, lo, up, current.x, opt.cauchy_a, cand_c, cand_p, &current.value, &current.gradient);
            int ls_it = 0, ls_ok = 1;
            while (!current.value_valid || current.value > x_cost + 1e-4 * g0 * current.x) {
                if (++ls_it >= 20) { ls_ok = 0; break; }
                double t = ls_next_step(&initial, &previous, &current, 1e-3 * current.x, 0.6 * current.x);
                if (t * dmax < 1e-9) { ls_ok = 0; break; }
                previous = current;
                current.x = t;
                current.value_valid = current.gradient_valid =
                    ls_evaluate(&S, x_c, x_p, delta_c, delta_p, lo, up, current.x, opt.cauchy_a, cand_c, cand_p, &current.value, &current.gradient);
            }
            cur.line_search_steps = ls_it;
            if (ls_ok && current.x != 1.0) {
                for (int i = 0; i < n; ++i) delta_c[i] *= current.x;
                for (size_t i = 0; i < 3 * (size_t)n_pt; ++i) delta_p[i] *= current.x;
            }
        }
        /* candidate = Plus(x, delta) */
        double step_norm = 0.0;
        for (int i = 0; i < n; ++i) {
            cand_c[i] = act_f[i] ? clamp_box(x_c[i] + delta_c[i], lo, up, (size_t)i) : x_c[i];
            double d = x_c[i] - cand_c[i]; step_norm += d * d;
        }
        for (size_t j = 0; j < 3 * (size_t)n_pt; ++j) {
            cand_p[j] = x_p[j] + delta_p[j];
            double d = x_p[j] - cand_p[j]; step_norm += d * d;
        }
        step_norm = sqrt(step_norm);
        double cand_cost = cost_any(n_obs, cam_idx, pt_idx, obs_uv, S.K4, calib ? cand_c + 6 * (size_t)n_cam : NULL, cand_c, cand_p, opt.cauchy_a);
        cur.step_norm = step_norm;
        cur.cost_change = x_cost - cand_cost;
        /* ParameterToleranceReached */
        if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) {
            sum->termination = ESFM_BA_CONVERGENCE; terminated = 1;
            cur.cost = x_cost; cur.trust_region_radius = radius;
            if (iter < ESFM_BA_MAX_LOG) sum->iterations[iter] = cur;
            sum->num_iterations = iter;
            break;
        }
        /* FunctionToleranceReached */
        if (fabs(cur.cost_change) <= opt.function_tolerance * x_cost) {
            sum->termination = ESFM_BA_CONVERGENCE; terminated = 1;
            cur.cost = x_cost; cur.trust_region_radius = radius;
            if (iter < ESFM_BA_MAX_LOG) sum->iterations[iter] = cur;
            sum->num_iterations = iter;
            break;
        }
        cur.relative_decrease = (x_cost - cand_cost) / model_cost_change;
        if (cur.relative_decrease > opt.min_relative_decrease) {
            /* HandleSuccessfulStep */
            memcpy(x_c, cand_c, sizeof(double) * (size_t)n);
            memcpy(x_p, cand_p, sizeof(double) * 3 * (size_t)n_pt);
            x_norm = 0.0;
            for (int i = 0; i < n; ++i) if (act_f[i]) x_norm += x_c[i] * x_c[i];
            for (int p = 0; p < n_pt; ++p) if (S.pt_start[p + 1] > S.pt_start[p]) for (int i = 0; i < 3; ++i) x_norm += x_p[3 * (size_t)p + i] * x_p[3 * (size_t)p + i];
            x_norm = sqrt(x_norm);
            x_cost = linearize(&S, x_c, x_p, opt.cauchy_a, opt.jacobi_scaling, lo, up, &gmax, &ok);
            if (!ok) { sum->termination = ESFM_BA_FAILURE; rc = -6; terminated = 1; }
            last_gmax = gmax;
            cur.step_is_successful = 1; cur.cost = x_cost; cur.gradient_max_norm = gmax;
            /* LevenbergMarquardtStrategy::StepAccepted */
            double q = 2.0 * cur.relative_decrease - 1.0;
            radius = radius / fmax(1.0 / 3.0, 1.0 - q * q * q);
            radius = fmin(opt.max_trust_region_radius, radius);
            decrease_factor = 2.0; reuse_diagonal = 0;
            sum->num_successful_steps++;
            if (gmax <= opt.gradient_tolerance) { sum->termination = ESFM_BA_CONVERGENCE; terminated = 1; }
        } else {
            /* HandleUnsuccessfulStep + StepRejected */
            cur.step_is_successful = 0; cur.cost = cand_cost;
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
            sum->num_unsuccessful_steps++;
        }
        cur.trust_region_radius = radius;
        if (iter < ESFM_BA_MAX_LOG) sum->iterations[iter] = cur;
        sum->num_iterations = iter;
        if (opt.verbose) printf("%4d % .6e  % .2e  % .2e  % .2e  % .2e  % .2e\n", iter, cur.cost, cur.cost_change, cur.gradient_max_norm, cur.step_norm, cur.relative_decrease, radius);
    }
    sum->final_cost = x_cost;
done:
    memcpy(cams, x_c, sizeof(double) * 6 * (size_t)n_cam);
    if (calib) memcpy(calib, x_c + 6 * (size_t)n_cam, sizeof(double) * 4);
    sum->solve_seconds = now_sec() - t0;
    free(x_c); free(act_f); free(lo); free(up);
    free(cand_c); free(cand_p); free(Sm); free(rhs); free(EtEinv); free(Etr); free(step_c); free(step_p);
    free(delta_c); free(delta_p); free(nc); free(np);
    state_free(&S);
    return rc;
}

/* ceres::Solve for the problem ba.cpp:140-151 builds: fixed intrinsics, no reference camera. */
int esfm_ref_ba_solve(int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx,
                      const float *obs_uv, const float *K4, double *cams, double *pts,
                      const esfm_ba_options *opt_in, esfm_ba_summary *sum)
{
    return esfm_ref_ba_solve_ex(n_cam, n_pt, n_obs, cam_idx, pt_idx, obs_uv, K4, cams, pts, NULL, 0.0, -1, 0.0, opt_in, sum);
}

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
 *     one residual block + CauchyLoss(0.5) per observation, no bounds when
 *     ba_calib_change_tolerance == 0 and no reference frame;
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
/* forward-mode dual numbers with 9 partials (6 camera + 3 point), ceres::Jet */
#define NJ 9
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

/* ReprojectErrorTerm_fixcalib::operator() (ba.h:113-153) on jets.
 * cam[6] = angle-axis, translation; K4 = fx, cx, fy, cy (float, ba.h:142-143);
 * uv = observed pixel (float, ba.h:145-146).  r[2], Jc[2][6], Jp[2][3]. */
void esfm_ref_ba_residual_jac(const double *cam, const double *pt, const float *K4, const float *uv,
                              double *r, double *Jc, double *Jp)
{
    jet aa[3], tr[3], X[3], p[3];
    for (int k = 0; k < 3; ++k) { aa[k] = jvar(cam[k], k); tr[k] = jvar(cam[3 + k], 3 + k); X[k] = jvar(pt[k], 6 + k); }
    jet_angle_axis_rotate(aa, X, p);
    for (int k = 0; k < 3; ++k) p[k] = jadd(p[k], tr[k]);
    jet x = jdiv(p[0], p[2]);
    jet y = jdiv(p[1], p[2]);
    jet u = jadd(jmul(x, jconst((double)K4[0])), jconst((double)K4[1]));
    jet v = jadd(jmul(y, jconst((double)K4[2])), jconst((double)K4[3]));
    jet r0 = jsub(jconst((double)uv[0]), u);
    jet r1 = jsub(jconst((double)uv[1]), v);
    r[0] = r0.a; r[1] = r1.a;
    if (Jc) for (int k = 0; k < 6; ++k) { Jc[k] = r0.v[k]; Jc[6 + k] = r1.v[k]; }
    if (Jp) for (int k = 0; k < 3; ++k) { Jp[k] = r0.v[6 + k]; Jp[3 + k] = r1.v[6 + k]; }
}

/* Residual only (candidate-cost evaluation), plain doubles, same formulas. */
static void residual_only(const double *cam, const double *pt, const float *K4, const float *uv, double *r)
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
    r[0] = (double)uv[0] - (x * (double)K4[0] + (double)K4[1]);
    r[1] = (double)uv[1] - (y * (double)K4[2] + (double)K4[3]);
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
double esfm_ref_ba_cost(int n_obs, const int32_t *cam_idx, const int32_t *pt_idx, const float *obs_uv,
                        const float *K4, const double *cams, const double *pts, double cauchy_a)
{
    double cost = 0.0; int bad = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : cost) reduction(| : bad) schedule(static)
#endif
    for (int k = 0; k < n_obs; ++k) {
        double r[2], rho[3];
        residual_only(cams + 6 * (size_t)cam_idx[k], pts + 3 * (size_t)pt_idx[k], K4 + 4 * (size_t)cam_idx[k], obs_uv + 2 * (size_t)k, r);
        double s = r[0] * r[0] + r[1] * r[1];
        if (!isfinite(s)) { bad |= 1; continue; }
        loss_eval(cauchy_a, s, rho);
        cost += 0.5 * rho[0];
    }
    return bad ? DBL_MAX : cost;
}

/* ------------------------------------------------------------------------- */
typedef struct {
    int n_cam, n_pt, n_obs;
    const int32_t *cam_idx, *pt_idx; const float *uv, *K4;
    /* observations grouped by point (CSR) */
    int32_t *pt_start; /* n_pt+1 */
    int32_t *order;    /* n_obs: original observation index, grouped by point */
    int32_t *cam_nobs;
    /* linearisation at x (corrected by the loss, columns scaled) */
    double *Jc;  /* 12 per obs, [2][6] */
    double *Jp;  /* 6 per obs, [2][3]  */
    double *r;   /* 2 per obs */
    double *scale_c, *scale_p; /* Jacobi scaling, 6*n_cam, 3*n_pt */
    double *diag_c, *diag_p;   /* LM diagonal (squared column norms, clamped) */
} ba_state;

/* Evaluate residuals + Jacobians at (cams, pts), apply the corrector
 * (corrector.cc: rho''<=0 -> scale both by sqrt(rho')), return cost; gradient
 * max-norm of the UNSCALED problem in *gmax (trust_region_minimizer.cc
 * EvaluateGradientAndJacobian: gradient before column scaling). */
static double linearize(ba_state *S, const double *cams, const double *pts, double cauchy_a,
                        int apply_scaling, double *gmax_out, int *ok)
{
    double cost = 0.0; int bad = 0;
    double *gc = (double *)calloc((size_t)6 * S->n_cam, sizeof(double));
    double *gp = (double *)calloc((size_t)3 * S->n_pt, sizeof(double));
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : cost) reduction(| : bad) schedule(static)
#endif
    for (int k = 0; k < S->n_obs; ++k) {
        double r[2], Jc[12], Jp[6], rho[3];
        int c = S->cam_idx[k], p = S->pt_idx[k];
        esfm_ref_ba_residual_jac(cams + 6 * (size_t)c, pts + 3 * (size_t)p, S->K4 + 4 * (size_t)c, S->uv + 2 * (size_t)k, r, Jc, Jp);
        double s = r[0] * r[0] + r[1] * r[1];
        int fin = isfinite(s);
        for (int i = 0; i < 12; ++i) fin &= isfinite(Jc[i]);
        for (int i = 0; i < 6; ++i) fin &= isfinite(Jp[i]);
        if (!fin) { bad |= 1; continue; }
        loss_eval(cauchy_a, s, rho);
        cost += 0.5 * rho[0];
        double sq = sqrt(rho[1]);
        for (int i = 0; i < 12; ++i) S->Jc[12 * (size_t)k + i] = Jc[i] * sq;
        for (int i = 0; i < 6; ++i) S->Jp[6 * (size_t)k + i] = Jp[i] * sq;
        S->r[2 * (size_t)k] = r[0] * sq; S->r[2 * (size_t)k + 1] = r[1] * sq;
    }
    *ok = !bad;
    if (bad) { free(gc); free(gp); return DBL_MAX; }
    /* gradient g = J' r (serial: deterministic order) */
    for (int k = 0; k < S->n_obs; ++k) {
        const double *Jc = S->Jc + 12 * (size_t)k, *Jp = S->Jp + 6 * (size_t)k, *r = S->r + 2 * (size_t)k;
        int c = S->cam_idx[k], p = S->pt_idx[k];
        for (int i = 0; i < 6; ++i) gc[6 * (size_t)c + i] += Jc[i] * r[0] + Jc[6 + i] * r[1];
        for (int i = 0; i < 3; ++i) gp[3 * (size_t)p + i] += Jp[i] * r[0] + Jp[3 + i] * r[1];
    }
    double gmax = 0.0;
    for (size_t i = 0; i < (size_t)6 * S->n_cam; ++i) if (fabs(gc[i]) > gmax) gmax = fabs(gc[i]);
    for (size_t i = 0; i < (size_t)3 * S->n_pt; ++i) if (fabs(gp[i]) > gmax) gmax = fabs(gp[i]);
    *gmax_out = gmax;
    free(gc); free(gp);
    if (apply_scaling) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
        for (int k = 0; k < S->n_obs; ++k) {
            int c = S->cam_idx[k], p = S->pt_idx[k];
            for (int row = 0; row < 2; ++row) {
                for (int i = 0; i < 6; ++i) S->Jc[12 * (size_t)k + 6 * row + i] *= S->scale_c[6 * (size_t)c + i];
                for (int i = 0; i < 3; ++i) S->Jp[6 * (size_t)k + 3 * row + i] *= S->scale_p[3 * (size_t)p + i];
            }
        }
    }
    return cost;
}

/* Squared column norms of the stored Jacobian. */
static void column_sqnorms(const ba_state *S, double *nc, double *np)
{
    memset(nc, 0, sizeof(double) * 6 * (size_t)S->n_cam);
    memset(np, 0, sizeof(double) * 3 * (size_t)S->n_pt);
    for (int k = 0; k < S->n_obs; ++k) {
        const double *Jc = S->Jc + 12 * (size_t)k, *Jp = S->Jp + 6 * (size_t)k;
        int c = S->cam_idx[k], p = S->pt_idx[k];
        for (int i = 0; i < 6; ++i) nc[6 * (size_t)c + i] += Jc[i] * Jc[i] + Jc[6 + i] * Jc[6 + i];
        for (int i = 0; i < 3; ++i) np[3 * (size_t)p + i] += Jp[i] * Jp[i] + Jp[3 + i] * Jp[3 + i];
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

/*
 * Reduced camera system of the damped normal equations (schur_eliminator_impl.h):
 *   S   = sum F'F + D_c^2 - sum_p W_p (E'E + D_p^2)^-1 W_p',  W_p = sum F_i'E_i
 *   rhs = sum F'r - sum_p W_p (E'E + D_p^2)^-1 E'r
 * over the observations of this state only (a shard builds a partial system;
 * D_c^2 is added iff add_cam_diag).  S is (6 n_cam)^2 row-major, full symmetric.
 * EtEinv (9/pt) and Etr (3/pt) are kept for the back-substitution.
 */
static void reduced_point(const ba_state *S, int p, double radius, double *Sm, double *rhs,
                          double *EtEinv, double *Etr, int *ok)
{
    const int n = 6 * S->n_cam;
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
        int k = S->order[t], ci = S->cam_idx[k];
        const double *Jc = S->Jc + 12 * (size_t)k, *Jp = S->Jp + 6 * (size_t)k, *r = S->r + 2 * (size_t)k;
        /* W_i = F_i' E_i (6x3); F'F and F'r */
        double W[18];
        for (int a = 0; a < 6; ++a)
            for (int j = 0; j < 3; ++j) W[3 * a + j] = Jc[a] * Jp[j] + Jc[6 + a] * Jp[3 + j];
        for (int a = 0; a < 6; ++a) {
            for (int c2 = 0; c2 < 6; ++c2)
                Sm[(size_t)(6 * ci + a) * n + 6 * ci + c2] += Jc[a] * Jc[c2] + Jc[6 + a] * Jc[6 + c2];
            rhs[6 * ci + a] += Jc[a] * r[0] + Jc[6 + a] * r[1];
            rhs[6 * ci + a] -= W[3 * a] * Aig[0] + W[3 * a + 1] * Aig[1] + W[3 * a + 2] * Aig[2];
        }
        /* Y = W_i Ai (6x3) */
        double Y[18];
        for (int a = 0; a < 6; ++a)
            for (int j = 0; j < 3; ++j)
                Y[3 * a + j] = W[3 * a] * Ai[j] + W[3 * a + 1] * Ai[3 + j] + W[3 * a + 2] * Ai[6 + j];
        for (int t2 = b; t2 < e; ++t2) {
            int k2 = S->order[t2], cj = S->cam_idx[k2];
            const double *Jc2 = S->Jc + 12 * (size_t)k2, *Jp2 = S->Jp + 6 * (size_t)k2;
            double W2[18]; /* W_j[c2][m] */
            for (int c2 = 0; c2 < 6; ++c2)
                for (int m = 0; m < 3; ++m) W2[3 * c2 + m] = Jc2[c2] * Jp2[m] + Jc2[6 + c2] * Jp2[3 + m];
            for (int a = 0; a < 6; ++a)
                for (int c2 = 0; c2 < 6; ++c2)
                    Sm[(size_t)(6 * ci + a) * n + 6 * cj + c2] -=
                        Y[3 * a] * W2[3 * c2] + Y[3 * a + 1] * W2[3 * c2 + 1] + Y[3 * a + 2] * W2[3 * c2 + 2];
        }
    }
}

static int build_reduced(const ba_state *S, double radius, int add_cam_diag,
                         double *Sm, double *rhs, double *EtEinv, double *Etr)
{
    const int n = 6 * S->n_cam;
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

/* Exported for the sharding test (tests/test_ba_sharding_gloo.py): partial
 * reduced system + LM diagonal pieces of an observation subset, linearised at
 * (cams, pts) with NO Jacobi scaling and a caller-supplied LM diagonal
 * (diag_c, diag_p = squared column norms, already clamped). */
int esfm_ref_ba_partial_reduced(int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx,
                                const float *obs_uv, const float *K4, const double *cams, const double *pts,
                                double cauchy_a, double radius, const double *diag_c, const double *diag_p,
                                int add_cam_diag, double *Sm /*(6nc)^2*/, double *rhs /*6nc*/);

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
    S->Jc = (double *)malloc(sizeof(double) * 12 * no);
    S->Jp = (double *)malloc(sizeof(double) * 6 * no);
    S->r = (double *)malloc(sizeof(double) * 2 * no);
    S->scale_c = (double *)malloc(sizeof(double) * 6 * (size_t)S->n_cam);
    S->scale_p = (double *)malloc(sizeof(double) * 3 * (size_t)S->n_pt);
    S->diag_c = (double *)calloc(6 * (size_t)S->n_cam, sizeof(double));
    S->diag_p = (double *)calloc(3 * (size_t)S->n_pt, sizeof(double));
    for (size_t i = 0; i < 6 * (size_t)S->n_cam; ++i) S->scale_c[i] = 1.0;
    for (size_t i = 0; i < 3 * (size_t)S->n_pt; ++i) S->scale_p[i] = 1.0;
}

static void state_free(ba_state *S)
{
    free(S->pt_start); free(S->order); free(S->cam_nobs);
    free(S->Jc); free(S->Jp); free(S->r); free(S->scale_c); free(S->scale_p); free(S->diag_c); free(S->diag_p);
}

int esfm_ref_ba_partial_reduced(int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx,
                                const float *obs_uv, const float *K4, const double *cams, const double *pts,
                                double cauchy_a, double radius, const double *diag_c, const double *diag_p,
                                int add_cam_diag, double *Sm, double *rhs)
{
    ba_state S; memset(&S, 0, sizeof(S));
    S.n_cam = n_cam; S.n_pt = n_pt; S.n_obs = n_obs; S.cam_idx = cam_idx; S.pt_idx = pt_idx; S.uv = obs_uv; S.K4 = K4;
    group_by_point(&S); state_alloc(&S);
    double gmax; int ok;
    linearize(&S, cams, pts, cauchy_a, 0, &gmax, &ok);
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
    linearize(&S, cams, pts, cauchy_a, 0, &gmax, &ok);
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

/* ceres::Solve for the problem ba.cpp:140-151 builds (see header). */
int esfm_ref_ba_solve(int n_cam, int n_pt, int n_obs, const int32_t *cam_idx, const int32_t *pt_idx,
                      const float *obs_uv, const float *K4, double *cams, double *pts,
                      const esfm_ba_options *opt_in, esfm_ba_summary *sum)
{
    esfm_ba_options opt;
    if (opt_in) opt = *opt_in; else esfm_ref_ba_options_default(&opt);
    esfm_ba_summary local; if (!sum) sum = &local;
    memset(sum, 0, sizeof(*sum));
    for (int k = 0; k < n_obs; ++k)
        if (cam_idx[k] < 0 || cam_idx[k] >= n_cam || pt_idx[k] < 0 || pt_idx[k] >= n_pt) return -1;

    ba_state S; memset(&S, 0, sizeof(S));
    S.n_cam = n_cam; S.n_pt = n_pt; S.n_obs = n_obs; S.cam_idx = cam_idx; S.pt_idx = pt_idx; S.uv = obs_uv; S.K4 = K4;
    group_by_point(&S); state_alloc(&S);
    const int n = 6 * n_cam;
    int nact_c = 0, nact_p = 0;
    for (int c = 0; c < n_cam; ++c) nact_c += S.cam_nobs[c] > 0;
    for (int p = 0; p < n_pt; ++p) nact_p += S.pt_start[p + 1] > S.pt_start[p];
    sum->num_active_cameras = nact_c; sum->num_active_points = nact_p;

    double *x_c = cams, *x_p = pts; /* current point lives in the caller's arrays */
    double *cand_c = (double *)malloc(sizeof(double) * 6 * (size_t)n_cam);
    double *cand_p = (double *)malloc(sizeof(double) * 3 * (size_t)n_pt);
    double *Sm = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * (n > 0 ? n : 1));
    double *rhs = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *EtEinv = (double *)malloc(sizeof(double) * 9 * (size_t)(n_pt > 0 ? n_pt : 1));
    double *Etr = (double *)malloc(sizeof(double) * 3 * (size_t)(n_pt > 0 ? n_pt : 1));
    double *step_c = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *step_p = (double *)malloc(sizeof(double) * 3 * (size_t)(n_pt > 0 ? n_pt : 1));
    double *nc = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *np = (double *)malloc(sizeof(double) * 3 * (size_t)(n_pt > 0 ? n_pt : 1));

    double t0 = now_sec();
    int rc = 0;
    /* ---- iteration 0 (TrustRegionMinimizer::IterationZero) ---- */
    double x_norm = 0.0;
    for (int c = 0; c < n_cam; ++c) if (S.cam_nobs[c] > 0) for (int i = 0; i < 6; ++i) x_norm += x_c[6 * c + i] * x_c[6 * c + i];
    for (int p = 0; p < n_pt; ++p) if (S.pt_start[p + 1] > S.pt_start[p]) for (int i = 0; i < 3; ++i) x_norm += x_p[3 * (size_t)p + i] * x_p[3 * (size_t)p + i];
    x_norm = sqrt(x_norm);
    double gmax = 0.0; int ok = 1;
    double x_cost = linearize(&S, x_c, x_p, opt.cauchy_a, 0, &gmax, &ok);
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
        for (int c = 0; c < n_cam; ++c) if (S.cam_nobs[c] == 0) for (int i = 0; i < 6; ++i) { rhs[6 * c + i] = 0.0; }
        if (lin_ok) lin_ok = chol_solve(Sm, rhs, n);
        if (lin_ok) {
            for (int i = 0; i < n; ++i) { step_c[i] = -rhs[i]; lin_ok &= isfinite(step_c[i]); }
            /* back-substitution: y_p = (E'E+D^2)^-1 (E'r - sum E'F y_c) */
            for (int p = 0; p < n_pt; ++p) {
                int b = S.pt_start[p], e = S.pt_start[p + 1];
                if (b == e) { step_p[3 * (size_t)p] = step_p[3 * (size_t)p + 1] = step_p[3 * (size_t)p + 2] = 0.0; continue; }
                double g[3] = { Etr[3 * (size_t)p], Etr[3 * (size_t)p + 1], Etr[3 * (size_t)p + 2] };
                for (int t = b; t < e; ++t) {
                    int k = S.order[t], c = cam_idx[k];
                    const double *Jc = S.Jc + 12 * (size_t)k, *Jp = S.Jp + 6 * (size_t)k;
                    double f0 = 0.0, f1 = 0.0; /* F y_c */
                    for (int a = 0; a < 6; ++a) { f0 += Jc[a] * rhs[6 * c + a]; f1 += Jc[6 + a] * rhs[6 * c + a]; }
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
                int c = cam_idx[k], p = pt_idx[k];
                const double *Jc = S.Jc + 12 * (size_t)k, *Jp = S.Jp + 6 * (size_t)k, *r = S.r + 2 * (size_t)k;
                double m0 = 0.0, m1 = 0.0;
                for (int a = 0; a < 6; ++a) { m0 += Jc[a] * step_c[6 * c + a]; m1 += Jc[6 + a] * step_c[6 * c + a]; }
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
        /* candidate = x + step .* scaling */
        double step_norm = 0.0;
        for (int i = 0; i < n; ++i) {
            cand_c[i] = x_c[i] + step_c[i] * S.scale_c[i];
            if (S.cam_nobs[i / 6] > 0) { double d = x_c[i] - cand_c[i]; step_norm += d * d; } else cand_c[i] = x_c[i];
        }
        for (int p = 0; p < n_pt; ++p) {
            int active = S.pt_start[p + 1] > S.pt_start[p];
            for (int i = 0; i < 3; ++i) {
                size_t j = 3 * (size_t)p + i;
                if (active) { cand_p[j] = x_p[j] + step_p[j] * S.scale_p[j]; double d = x_p[j] - cand_p[j]; step_norm += d * d; }
                else cand_p[j] = x_p[j];
            }
        }
        step_norm = sqrt(step_norm);
        double cand_cost = esfm_ref_ba_cost(n_obs, cam_idx, pt_idx, obs_uv, K4, cand_c, cand_p, opt.cauchy_a);
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
            for (int c = 0; c < n_cam; ++c) if (S.cam_nobs[c] > 0) for (int i = 0; i < 6; ++i) x_norm += x_c[6 * c + i] * x_c[6 * c + i];
            for (int p = 0; p < n_pt; ++p) if (S.pt_start[p + 1] > S.pt_start[p]) for (int i = 0; i < 3; ++i) x_norm += x_p[3 * (size_t)p + i] * x_p[3 * (size_t)p + i];
            x_norm = sqrt(x_norm);
            x_cost = linearize(&S, x_c, x_p, opt.cauchy_a, opt.jacobi_scaling, &gmax, &ok);
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
    sum->solve_seconds = now_sec() - t0;
    free(cand_c); free(cand_p); free(Sm); free(rhs); free(EtEinv); free(Etr); free(step_c); free(step_p); free(nc); free(np);
    state_free(&S);
    return rc;
}

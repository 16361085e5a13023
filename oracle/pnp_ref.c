/*
 * oracle/pnp_ref.c -- CPU restatement of the 3-D/2-D registration the reference runs for every new frame:
 * cv::solvePnPRansac(pts3d, pts2d, K, dist = 0, rvec, tvec, false, iterationsCount, reprojectionError, confidence, inliers,
 * cv::SOLVEPNP_EPNP) (reference cpp_code/src/estimate_motion.cpp:161-162, called at cpp_code/test/sfm.cpp:288).
 * SURVEY.md section 8 row f-1.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include, link, call or execute this file
 * (see oracle/match_ref.c header).
 *
 * PARITY UNPINNED: OpenCV (>= 3, unpinned; author ran 3.4.2) is absent and the reference holds no fixture.  Restated from
 * memory of OpenCV 3.4 [upstream modules/calib3d/src/solvepnp.cpp, epnp.cpp, ptsetreg.cpp]:
 *   solvePnPRansac   points to float; RANSAC registrator with 5 model points (EPnP kernel), threshold = reprojectionError,
 *                    confidence, at most iterationsCount iterations, sample stream of cv::RNG((uint64)-1); one model per
 *                    sample (rvec | tvec); error = squared pixel distance between the observation and the projection, as
 *                    float, cut at (float)threshold^2; then EPnP once more on all inliers of the best model (doubles).
 *   EPnP             control points = centroid + principal axes scaled by sqrt(eigenvalue / n); barycentric coordinates;
 *                    M (2n x 12), eigenvectors of M'M for the four smallest eigenvalues; L (6 x 10), rho; betas from the
 *                    three approximations (N = 4 unknowns B11 B12 B13 B14; N = 3: B11 B12 B22; N = 5: B11 B12 B22 B13 B23),
 *                    each refined by 5 Gauss-Newton steps; for each, camera-frame control points, sign fix, absolute
 *                    orientation (Horn / Arun via a 3 x 3 SVD, last row of R negated when det R < 0) and the mean
 *                    reprojection error; the smallest error wins (1, then 2 if strictly smaller, then 3 if strictly smaller).
 *   rvec             cv::Rodrigues(R) (matrix -> vector).
 * Linear-algebra substitutes: symmetric eigen-decompositions by cyclic Jacobi where OpenCV calls cvSVD on a symmetric
 * matrix (same subspaces; a singular vector's sign is arbitrary in both and cancels in solve_for_sign), least squares by
 * normal equations + Jacobi where OpenCV uses cvSolve(CV_SVD) / a Householder QR (same minimiser up to rounding).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t state; } cv_rng;
static inline unsigned rng_next(cv_rng *r) { r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32); return (unsigned)r->state; }
static inline int rng_uniform(cv_rng *r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

static void jacobi_sym(double *A, int n, double *V)
{
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    /* at most 30 sweeps: cvSVD's Jacobi routine stops at max(m, 30) (JacobiSVDImpl_: `max_iter = std::max(m, 30)`, m <= 12 here).  A matrix
     * that has not reached the threshold below by then never will -- its off-diagonal norm has stalled at rounding level (one M'M in
     * forty) -- and until round 5 this restatement let such a matrix run to 100 sweeps, which decides nothing but the noise in the
     * null-space vectors EPnP reads. */
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i) { diag += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j]; }
        /* converged when the off-diagonal part is below 10 DBL_EPSILON of the diagonal part (in norm): the scale of cvSVD's own test
         * (`eps = DBL_EPSILON * 10`, a pair is left alone once |p| <= eps sqrt(a b)).  Until round 5 this read 1e-40, which double
         * arithmetic cannot deliver for one 12 x 12 M'M in forty: those ran into the sweep cap, deciding nothing but noise. */
        if (off <= 4.9303806576313238e-30 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double th = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(1.0 + th * th));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int r = 0; r < n; ++r) { const double x = A[r * n + p], y = A[r * n + q]; A[r * n + p] = c * x - s * y; A[r * n + q] = s * x + c * y; }
                for (int r = 0; r < n; ++r) { const double x = A[p * n + r], y = A[q * n + r]; A[p * n + r] = c * x - s * y; A[q * n + r] = s * x + c * y; }
                for (int r = 0; r < n; ++r) { const double x = V[r * n + p], y = V[r * n + q]; V[r * n + p] = c * x - s * y; V[r * n + q] = s * x + c * y; }
            }
    }
}

/* eigenvectors sorted by DESCENDING eigenvalue (cvSVD order): out rows = vectors (U^T layout), d = values */
static void sym_eig_desc(const double *A_in, int n, double *ut, double *d)
{
    double A[144], V[144]; int o[12];
    memcpy(A, A_in, sizeof(double) * (size_t)n * n);
    jacobi_sym(A, n, V);
    for (int i = 0; i < n; ++i) o[i] = i;
    for (int i = 1; i < n; ++i) { int v = o[i], j = i - 1; while (j >= 0 && A[o[j] * n + o[j]] < A[v * n + v]) { o[j + 1] = o[j]; --j; } o[j + 1] = v; }
    for (int k = 0; k < n; ++k) { d[k] = A[o[k] * n + o[k]]; for (int a = 0; a < n; ++a) ut[k * n + a] = V[a * n + o[k]]; }
}

/* minimum-norm least squares x = argmin |A x - b| (A m x n, m >= n) through the eigen-decomposition of A'A (= cvSolve CV_SVD) */
static void lstsq(const double *A, const double *b, int m, int n, double *x)
{
    double G[36], V[36], g[6];
    for (int i = 0; i < n; ++i) { g[i] = 0; for (int k = 0; k < m; ++k) g[i] += A[k * n + i] * b[k]; for (int j = 0; j < n; ++j) { G[i * n + j] = 0; for (int k = 0; k < m; ++k) G[i * n + j] += A[k * n + i] * A[k * n + j]; } }
    jacobi_sym(G, n, V);
    double mx = 0; for (int i = 0; i < n; ++i) mx = fmax(mx, G[i * n + i]);
    for (int i = 0; i < n; ++i) x[i] = 0;
    for (int k = 0; k < n; ++k) {
        const double ev = G[k * n + k];
        if (!(ev > mx * 1e-28)) continue;                      /* singular direction: minimum-norm solution drops it */
        double c = 0; for (int i = 0; i < n; ++i) c += V[i * n + k] * g[i];
        c /= ev;
        for (int i = 0; i < n; ++i) x[i] += c * V[i * n + k];
    }
}

static double det3(const double *M) { return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]); }

/* absolute orientation (epnp::estimate_R_and_t): pcs -> R pws + t */
static void estimate_R_and_t(const double *pws, const double *pcs, int n, double R[9], double t[3])
{
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
    for (int j = 0; j < 3; ++j) { pc0[j] /= n; pw0[j] /= n; }
    double ABt[9] = {0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) ABt[3 * j + k] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + k] - pw0[k]);
    /* SVD ABt = U D V': V from ABt'ABt, U = ABt V / d (third column by cross product) */
    double G[9], V[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { G[3 * a + b] = 0; for (int k = 0; k < 3; ++k) G[3 * a + b] += ABt[3 * k + a] * ABt[3 * k + b]; }
    jacobi_sym(G, 3, V);
    int o[3] = {0, 1, 2};
    for (int i = 1; i < 3; ++i) { int v = o[i], j = i - 1; while (j >= 0 && G[4 * o[j]] < G[4 * v]) { o[j + 1] = o[j]; --j; } o[j + 1] = v; }
    double v[3][3], u[3][3];
    for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) v[k][a] = V[3 * a + o[k]];
    /* u_k = ABt v_k / sigma_k: the sign of each singular pair is tied by ABt, so U V' does not depend on the eigenvector
     * signs.  A vanishing third singular value (coplanar point set) leaves u_2 undetermined: take the right-handed one. */
    double sig[3];
    for (int k = 0; k < 3; ++k) {
        double nn = 0;
        for (int a = 0; a < 3; ++a) { u[k][a] = ABt[3 * a] * v[k][0] + ABt[3 * a + 1] * v[k][1] + ABt[3 * a + 2] * v[k][2]; nn += u[k][a] * u[k][a]; }
        sig[k] = sqrt(nn);
        if (k < 2 || sig[2] > 1e-12 * sig[0]) for (int a = 0; a < 3; ++a) u[k][a] /= sig[k];
    }
    if (!(sig[2] > 1e-12 * sig[0])) {
        u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1]; u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2]; u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
        const double w0 = v[0][1] * v[1][2] - v[0][2] * v[1][1], w1 = v[0][2] * v[1][0] - v[0][0] * v[1][2], w2 = v[0][0] * v[1][1] - v[0][1] * v[1][0];
        v[2][0] = w0; v[2][1] = w1; v[2][2] = w2;
    }
    /* R = U V' (epnp.cpp: R[i][j] = dot(U row i, V row j)); a reflection gets its last row negated, as OpenCV does */
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[3 * r + c] = u[0][r] * v[0][c] + u[1][r] * v[1][c] + u[2][r] * v[2][c];
    if (det3(R) < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    for (int j = 0; j < 3; ++j) t[j] = pc0[j] - (R[3 * j] * pw0[0] + R[3 * j + 1] * pw0[1] + R[3 * j + 2] * pw0[2]);
}

/* Moore-Penrose inverse of a 3 x 3 through the eigen-decomposition of A'A (cvInvert with CV_SVD) */
static void pinv3(const double *A, double *Ai)
{
    double G[9], V[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { G[3 * a + b] = 0; for (int k = 0; k < 3; ++k) G[3 * a + b] += A[3 * k + a] * A[3 * k + b]; }
    jacobi_sym(G, 3, V);
    double mx = fmax(G[0], fmax(G[4], G[8]));
    for (int i = 0; i < 9; ++i) Ai[i] = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double ev = G[4 * k];
        if (!(ev > mx * 1e-24)) continue;                       /* sigma_k <= 1e-12 sigma_max: dropped */
        double Av[3];
        for (int r = 0; r < 3; ++r) Av[r] = A[3 * r] * V[k] + A[3 * r + 1] * V[3 + k] + A[3 * r + 2] * V[6 + k];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ai[3 * i + j] += V[3 * i + k] * Av[j] / ev;   /* v_k (A v_k)' / sigma_k^2 */
    }
}

typedef struct { double fu, fv, uc, vc; } cam_t;

/* epnp::compute_pose.  pws: n x 3 world points, us: n x 2 pixels.  Returns the mean reprojection error of the chosen pose. */
static double epnp_pose(const cam_t *cam, const double *pws, const double *us, int n, double R[9], double t[3])
{
    /* control points */
    double cws[4][3] = {{0}};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) cws[0][j] += pws[3 * i + j];
    for (int j = 0; j < 3; ++j) cws[0][j] /= n;
    double C[9] = {0};
    for (int i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[3 * a + b] += (pws[3 * i + a] - cws[0][a]) * (pws[3 * i + b] - cws[0][b]);
    double uct[9], dc[3];
    sym_eig_desc(C, 3, uct, dc);
    for (int i = 1; i < 4; ++i) { const double k = sqrt(fmax(dc[i - 1], 0.0) / n); for (int j = 0; j < 3; ++j) cws[i][j] = cws[0][j] + k * uct[3 * (i - 1) + j]; }
    /* barycentric coordinates */
    double CC[9], CCi[9];
    for (int i = 0; i < 3; ++i) for (int j = 1; j < 4; ++j) CC[3 * i + j - 1] = cws[j][i] - cws[0][i];
    pinv3(CC, CCi);   /* cvInvert(&CC, &CC_inv, CV_SVD): the pseudo-inverse when the points are coplanar */
    double *alphas = (double *)malloc(sizeof(double) * 4 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        double *a = alphas + 4 * i;
        for (int j = 0; j < 3; ++j)
            a[1 + j] = CCi[3 * j] * (pws[3 * i] - cws[0][0]) + CCi[3 * j + 1] * (pws[3 * i + 1] - cws[0][1]) + CCi[3 * j + 2] * (pws[3 * i + 2] - cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
    /* M'M */
    double MtM[144]; memset(MtM, 0, sizeof(MtM));
    for (int i = 0; i < n; ++i) {
        const double *as = alphas + 4 * i, u = us[2 * i], v = us[2 * i + 1];
        double m1[12], m2[12];
        for (int k = 0; k < 4; ++k) {
            m1[3 * k] = as[k] * cam->fu; m1[3 * k + 1] = 0.0; m1[3 * k + 2] = as[k] * (cam->uc - u);
            m2[3 * k] = 0.0; m2[3 * k + 1] = as[k] * cam->fv; m2[3 * k + 2] = as[k] * (cam->vc - v);
        }
        for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) MtM[12 * a + b] += m1[a] * m1[b] + m2[a] * m2[b];
    }
    double ut[144], d[12];
    sym_eig_desc(MtM, 12, ut, d);
    const double *v[4] = { ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8 };
    /* L (6 x 10) and rho */
    double dv[4][6][3], L[60], rho[6];
    for (int i = 0; i < 4; ++i) {
        int a = 0, b = 1;
        for (int j = 0; j < 6; ++j) {
            for (int k = 0; k < 3; ++k) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
            ++b; if (b > 3) { ++a; b = a + 1; }
        }
    }
#define DOT(p, q) ((p)[0] * (q)[0] + (p)[1] * (q)[1] + (p)[2] * (q)[2])
    for (int i = 0; i < 6; ++i) {
        double *row = L + 10 * i;
        row[0] = DOT(dv[0][i], dv[0][i]); row[1] = 2.0 * DOT(dv[0][i], dv[1][i]); row[2] = DOT(dv[1][i], dv[1][i]);
        row[3] = 2.0 * DOT(dv[0][i], dv[2][i]); row[4] = 2.0 * DOT(dv[1][i], dv[2][i]); row[5] = DOT(dv[2][i], dv[2][i]);
        row[6] = 2.0 * DOT(dv[0][i], dv[3][i]); row[7] = 2.0 * DOT(dv[1][i], dv[3][i]); row[8] = 2.0 * DOT(dv[2][i], dv[3][i]); row[9] = DOT(dv[3][i], dv[3][i]);
    }
    {
        int a = 0, b = 1;
        for (int j = 0; j < 6; ++j) {
            double s = 0; for (int k = 0; k < 3; ++k) { const double e = cws[a][k] - cws[b][k]; s += e * e; }
            rho[j] = s; ++b; if (b > 3) { ++a; b = a + 1; }
        }
    }
    double betas[4][4], errs[4], Rs[4][9], ts[4][3];
    /* approximation 1: B11 B12 B13 B14 */
    {
        double A[24], x[4];
        for (int i = 0; i < 6; ++i) { A[4 * i] = L[10 * i]; A[4 * i + 1] = L[10 * i + 1]; A[4 * i + 2] = L[10 * i + 3]; A[4 * i + 3] = L[10 * i + 6]; }
        lstsq(A, rho, 6, 4, x);
        double *b = betas[1];
        if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = -x[1] / b[0]; b[2] = -x[2] / b[0]; b[3] = -x[3] / b[0]; }
        else { b[0] = sqrt(x[0]); b[1] = x[1] / b[0]; b[2] = x[2] / b[0]; b[3] = x[3] / b[0]; }
    }
    /* approximation 2: B11 B12 B22 */
    {
        double A[18], x[3];
        for (int i = 0; i < 6; ++i) { A[3 * i] = L[10 * i]; A[3 * i + 1] = L[10 * i + 1]; A[3 * i + 2] = L[10 * i + 2]; }
        lstsq(A, rho, 6, 3, x);
        double *b = betas[2];
        if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
        else { b[0] = sqrt(x[0]); b[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
        if (x[1] < 0) b[0] = -b[0];
        b[2] = 0.0; b[3] = 0.0;
    }
    /* approximation 3: B11 B12 B22 B13 B23 */
    {
        double A[30], x[5];
        for (int i = 0; i < 6; ++i) for (int k = 0; k < 5; ++k) A[5 * i + k] = L[10 * i + k];
        lstsq(A, rho, 6, 5, x);
        double *b = betas[3];
        if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
        else { b[0] = sqrt(x[0]); b[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
        if (x[1] < 0) b[0] = -b[0];
        b[2] = x[3] / b[0]; b[3] = 0.0;
    }
    double *pcs = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    for (int N = 1; N <= 3; ++N) {
        double *b = betas[N];
        for (int it = 0; it < 5; ++it) {   /* gauss_newton */
            double A[24], rhs[6], x[4];
            for (int i = 0; i < 6; ++i) {
                const double *r = L + 10 * i;
                A[4 * i] = 2 * r[0] * b[0] + r[1] * b[1] + r[3] * b[2] + r[6] * b[3];
                A[4 * i + 1] = r[1] * b[0] + 2 * r[2] * b[1] + r[4] * b[2] + r[7] * b[3];
                A[4 * i + 2] = r[3] * b[0] + r[4] * b[1] + 2 * r[5] * b[2] + r[8] * b[3];
                A[4 * i + 3] = r[6] * b[0] + r[7] * b[1] + r[8] * b[2] + 2 * r[9] * b[3];
                rhs[i] = rho[i] - (r[0] * b[0] * b[0] + r[1] * b[0] * b[1] + r[2] * b[1] * b[1] + r[3] * b[0] * b[2] + r[4] * b[1] * b[2] +
                                   r[5] * b[2] * b[2] + r[6] * b[0] * b[3] + r[7] * b[1] * b[3] + r[8] * b[2] * b[3] + r[9] * b[3] * b[3]);
            }
            lstsq(A, rhs, 6, 4, x);
            for (int k = 0; k < 4; ++k) b[k] += x[k];
        }
        /* compute_ccs, compute_pcs, solve_for_sign */
        double ccs[4][3] = {{0}};
        for (int k = 0; k < 4; ++k) for (int c = 0; c < 4; ++c) for (int j = 0; j < 3; ++j) ccs[c][j] += b[k] * v[k][3 * c + j];
        for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) { const double *a = alphas + 4 * i; pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j]; }
        if (pcs[2] < 0.0) for (int i = 0; i < 3 * n; ++i) pcs[i] = -pcs[i];
        estimate_R_and_t(pws, pcs, n, Rs[N], ts[N]);
        double sum = 0;
        for (int i = 0; i < n; ++i) {
            const double *pw = pws + 3 * i, *Rn = Rs[N];
            const double Xc = Rn[0] * pw[0] + Rn[1] * pw[1] + Rn[2] * pw[2] + ts[N][0], Yc = Rn[3] * pw[0] + Rn[4] * pw[1] + Rn[5] * pw[2] + ts[N][1];
            const double inv = 1.0 / (Rn[6] * pw[0] + Rn[7] * pw[1] + Rn[8] * pw[2] + ts[N][2]);
            const double ue = cam->uc + cam->fu * Xc * inv, ve = cam->vc + cam->fv * Yc * inv;
            sum += sqrt((us[2 * i] - ue) * (us[2 * i] - ue) + (us[2 * i + 1] - ve) * (us[2 * i + 1] - ve));
        }
        errs[N] = sum / n;
    }
    int N = 1;
    if (errs[2] < errs[1]) N = 2;
    if (errs[3] < errs[N]) N = 3;
    memcpy(R, Rs[N], sizeof(double) * 9); memcpy(t, ts[N], sizeof(double) * 3);
    free(alphas); free(pcs);
    return errs[N];
}

/* exported: EPnP alone.  pts3d: n x 3 doubles, pix: n x 2 doubles (pixels), K4 = fx, cx, fy, cy. */
double esfm_ref_epnp(const double *pts3d, const double *pix, int n, const double *K4, double *R, double *t)
{
    cam_t cam = { K4[0], K4[2], K4[1], K4[3] };
    return epnp_pose(&cam, pts3d, pix, n, R, t);
}

/* cv::Rodrigues(R -> rvec) [upstream calib3d.cpp, matrix branch] */
void esfm_ref_rodrigues_to_vec(const double *R, double *rvec)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { rvec[0] = rvec[1] = rvec[2] = 0.0; return; }
        double t0 = (R[0] + 1) * 0.5, t1 = (R[4] + 1) * 0.5, t2 = (R[8] + 1) * 0.5;
        rx = sqrt(fmax(t0, 0.0)); ry = sqrt(fmax(t1, 0.0)) * (R[1] < 0 ? -1.0 : 1.0); rz = sqrt(fmax(t2, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        const double k = theta / sqrt(rx * rx + ry * ry + rz * rz);
        rvec[0] = rx * k; rvec[1] = ry * k; rvec[2] = rz * k;
        return;
    }
    const double vth = 1.0 / (2.0 * s) * theta;
    rvec[0] = rx * vth; rvec[1] = ry * vth; rvec[2] = rz * vth;
}

static int pnp_inliers(const cam_t *cam, const float *p3, const float *p2, int n, const double *R, const double *t, double thresh, uint8_t *mask)
{
    const float tt = (float)(thresh * thresh);
    int nz = 0;
    for (int i = 0; i < n; ++i) {
        const double X = (double)p3[3 * i], Y = (double)p3[3 * i + 1], Z = (double)p3[3 * i + 2];
        const double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0], yc = R[3] * X + R[4] * Y + R[5] * Z + t[1], zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        const double iz = zc != 0.0 ? 1.0 / zc : 1.0;                       /* cvProjectPoints2: z = z ? 1/z : 1 */
        /* projectPoints on float points returns Point2f; the difference and its squared norm are float arithmetic */
        const float u = (float)(cam->fu * (xc * iz) + cam->uc), v = (float)(cam->fv * (yc * iz) + cam->vc);
        const float dx = p2[2 * i] - u, dy = p2[2 * i + 1] - v;
        const float err = dx * dx + dy * dy;
        const int f = err <= tt;
        mask[i] = (uint8_t)f; nz += f;
    }
    return nz;
}

static int update_iters(double p, double ep, int model_points, int max_iters)
{
    p = fmax(p, 0.0); p = fmin(p, 1.0); ep = fmax(ep, 0.0); ep = fmin(ep, 1.0);
    double num = fmax(1.0 - p, DBL_MIN), denom = 1.0 - pow(1.0 - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* cv::solvePnPRansac(..., SOLVEPNP_EPNP).  p3: n x 3 float, p2: n x 2 float pixels, K4 = fx, cx, fy, cy (float).
 * R[9], t[3], rvec[3] out; mask[n] = inliers of the best RANSAC model.  Returns 1 on success (OpenCV's `true`). */
int esfm_ref_solve_pnp_ransac(const float *p3, const float *p2, int n, const float *K4, int iterations_count, double reproj_error,
                              double confidence, double *R, double *t, double *rvec, uint8_t *mask, int32_t *iters_run, int32_t *n_inliers)
{
    if (iters_run) *iters_run = 0;
    if (n_inliers) *n_inliers = 0;
    if (n < 5) return 0;     /* OpenCV switches to P3P for n == 4 and asserts below; not restated */
    cam_t cam = { (double)K4[0], (double)K4[2], (double)K4[1], (double)K4[3] };
    uint8_t *cur = (uint8_t *)malloc((size_t)n);
    int niters = iterations_count > 1 ? iterations_count : 1, max_good = 0, iter = 0;
    cv_rng rng = { 0xFFFFFFFFFFFFFFFFull };
    double bestR[9], bestt[3];
    if (n == 5) {
        double w[15], px[10];
        for (int i = 0; i < 5; ++i) { for (int k = 0; k < 3; ++k) w[3 * i + k] = (double)p3[3 * i + k]; px[2 * i] = (double)p2[2 * i]; px[2 * i + 1] = (double)p2[2 * i + 1]; }
        epnp_pose(&cam, w, px, 5, bestR, bestt);
        memset(mask, 1, (size_t)n); max_good = 5;
    } else {
        for (iter = 0; iter < niters; ++iter) {
            int id[5];
            for (int i = 0; i < 5;) {
                int v;
                for (;;) { v = id[i] = rng_uniform(&rng, 0, n); int j = 0; for (; j < i; ++j) if (v == id[j]) break; if (j == i) break; }
                ++i;
            }
            double w[15], px[10], Rm[9], tm[3];
            for (int i = 0; i < 5; ++i) { for (int k = 0; k < 3; ++k) w[3 * i + k] = (double)p3[3 * id[i] + k]; px[2 * i] = (double)p2[2 * id[i]]; px[2 * i + 1] = (double)p2[2 * id[i] + 1]; }
            epnp_pose(&cam, w, px, 5, Rm, tm);
            int finite = 1;
            for (int k = 0; k < 9; ++k) finite &= isfinite(Rm[k]);
            for (int k = 0; k < 3; ++k) finite &= isfinite(tm[k]);
            if (!finite) continue;
            const int good = pnp_inliers(&cam, p3, p2, n, Rm, tm, reproj_error, cur);
            if (good > (max_good > 4 ? max_good : 4)) {
                memcpy(mask, cur, (size_t)n); memcpy(bestR, Rm, sizeof(bestR)); memcpy(bestt, tm, sizeof(bestt));
                max_good = good;
                niters = update_iters(confidence, (double)(n - good) / n, 5, niters);
            }
        }
    }
    free(cur);
    if (iters_run) *iters_run = iter;
    if (max_good <= 0) return 0;
    /* EPnP on all inliers of the best model */
    int m = 0;
    for (int i = 0; i < n; ++i) m += mask[i];
    double *w = (double *)malloc(sizeof(double) * 3 * (size_t)m), *px = (double *)malloc(sizeof(double) * 2 * (size_t)m);
    for (int i = 0, k = 0; i < n; ++i) if (mask[i]) { for (int c = 0; c < 3; ++c) w[3 * k + c] = (double)p3[3 * i + c]; px[2 * k] = (double)p2[2 * i]; px[2 * k + 1] = (double)p2[2 * i + 1]; ++k; }
    epnp_pose(&cam, w, px, m, R, t);
    free(w); free(px);
    esfm_ref_rodrigues_to_vec(R, rvec);
    if (n_inliers) *n_inliers = m;
    return 1;
}

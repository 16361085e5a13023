// EPnP (Lepetit, Moreno-Noguer, Fua) as cv::solvePnP(..., SOLVEPNP_EPNP) runs it [upstream opencv/modules/calib3d/src/epnp.cpp],
// restated as small fixed-size routines that compile for the host and for gfx950: the per-hypothesis solver of
// pnp_solve_kernel calls them on 5 points inside one thread; the refit on all RANSAC inliers calls the same routines on the
// host between the kernels that reduce over the correspondences (pnp_kernels.hip).  Reference call site:
// cpp_code/src/estimate_motion.cpp:161-162 (cv::solvePnPRansac, SOLVEPNP_EPNP).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#define ESFM_HD __host__ __device__ inline

#ifndef EPNP_MARK
#define EPNP_MARK(k) do { } while (0)      // (pnp_kernels.hip -DESFM_PNP_TRACE: stage stamps of the device chain)
#endif
namespace esfm {
namespace epnp {

struct Cam { double fu, fv, uc, vc; };

// the stopping rule of the CPU restatement's jacobi_sym (oracle/pnp_ref.c), to the letter: EPnP reads the FOUR SMALLEST eigenpairs of M'M,
// which sit at the rounding level of the largest, and a sweep more or less changes them in the fifth digit.  Until round 5 this side
// stopped at 60 sweeps / 1e-36 and the oracle at 100 sweeps / 1e-40: a random sweep (tests/stress_pnp.py) found 3.7 % of the RANSAC
// problems choosing differently, and both thresholds were beyond what double arithmetic delivers for 1 - 2.5 % of the matrices, which
// then ran into the cap.  Both sides now stop where cvSVD's Jacobi routine does: 10 DBL_EPSILON, at most max(m, 30) sweeps.
constexpr int kJacobiSweeps = 30;       // (cvSVD's Jacobi routine: max(m, 30) sweeps)
constexpr double kJacobiOff = 4.9303806576313238e-30;      // (10 DBL_EPSILON)^2: the scale of cvSVD's own convergence test
// cos / sin of a rotation from th = (a_qq - a_pp) / (2 a_pq):  t = sign(th) / (|th| + sqrt(1 + th^2)),  c = 1 / sqrt(1 + t^2),  s = t c.
// For 2^27 <= |th| <= 2^500 the chain collapses EXACTLY, in IEEE double arithmetic: th^2 >= 2^54, so 1 + th^2 rounds to th^2;
// sqrt of a correctly rounded square is |th| (radix 2, no overflow below 2^511); |th| + |th| is exact; so t is the same division
// sign / (2 |th|) either way, |t| <= 2^-28, 1 + t^2 rounds to 1, c = 1 and s = t.  Two long-latency operations instead of five --
// what the sweeps of a nearly diagonal matrix (the last sweeps of every diagonalisation, all 30 of a stalled one) consist of.
// Checked against the CPU restatement's full chain on 20 000 five-point samples, bit for bit through every stage of the solve.
ESFM_HD void jacobi_cs(double th, double &c, double &s)
{
    const double ath = fabs(th);
    if (ath >= 134217728.0 && ath <= 0x1p500) { c = 1.0; s = (th >= 0.0 ? 1.0 : -1.0) / (ath + ath); return; }
    const double t = (th >= 0.0 ? 1.0 : -1.0) / (ath + sqrt(1.0 + th * th));
    c = 1.0 / sqrt(1.0 + t * t); s = t * c;
}
template <int N> ESFM_HD void jacobi_sym(double *A, double *V)
{
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) V[i * N + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < kJacobiSweeps; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < N; ++i) { diag += A[i * N + i] * A[i * N + i]; for (int j = i + 1; j < N; ++j) off += A[i * N + j] * A[i * N + j]; }
        if (off <= kJacobiOff * diag || off == 0.0) break;
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                const double apq = A[p * N + q];
                if (apq == 0.0) continue;
                const double th = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
                double c, s;
                jacobi_cs(th, c, s);
                for (int r = 0; r < N; ++r) { const double x = A[r * N + p], y = A[r * N + q]; A[r * N + p] = c * x - s * y; A[r * N + q] = s * x + c * y; }
                for (int r = 0; r < N; ++r) { const double x = A[p * N + r], y = A[q * N + r]; A[p * N + r] = c * x - s * y; A[q * N + r] = s * x + c * y; }
                for (int r = 0; r < N; ++r) { const double x = V[r * N + p], y = V[r * N + q]; V[r * N + p] = c * x - s * y; V[r * N + q] = s * x + c * y; }
            }
    }
}

// eigenvectors as ROWS of ut, eigenvalues descending (the layout cvSVD(..., CV_SVD_U_T) returns for a symmetric matrix)
// (the `_ws` forms take their large work arrays from the caller: pnp_solve_kernel keeps them in LDS -- as thread-private arrays indexed
// inside loops they live in scratch memory, one memory round trip per access: 17 ms per launch for 1024 hypotheses)
template <int N> struct JacobiSerial { ESFM_HD void operator()(double *A, double *V) const { jacobi_sym<N>(A, V); } };
// JAC: what diagonalises A in place and leaves the rotations in V -- jacobi_sym<N>, or a form that shares the same rotations' row /
// column updates among the lanes of a wave (pnp_kernels.hip): the same operations on the same operands in the same order
template <int N, class JAC> ESFM_HD void sym_eig_desc_ws(const double *A_in, double *ut, double *d, double *A /* N x N */, double *V /* N x N */, JAC jac)
{
    int o[N];
    for (int i = 0; i < N * N; ++i) A[i] = A_in[i];
    jac(A, V);
    for (int i = 0; i < N; ++i) o[i] = i;
    for (int i = 1; i < N; ++i) { const int v = o[i]; int j = i - 1; while (j >= 0 && A[o[j] * N + o[j]] < A[v * N + v]) { o[j + 1] = o[j]; --j; } o[j + 1] = v; }
    for (int k = 0; k < N; ++k) { d[k] = A[o[k] * N + o[k]]; for (int a = 0; a < N; ++a) ut[k * N + a] = V[a * N + o[k]]; }
}
template <int N> ESFM_HD void sym_eig_desc(const double *A_in, double *ut, double *d)
{
    double A[N * N], V[N * N];
    sym_eig_desc_ws<N>(A_in, ut, d, A, V, JacobiSerial<N>());
}

// minimum-norm least squares through the eigen-decomposition of A'A (what cvSolve(CV_SVD) / the QR of epnp.cpp minimise)
template <int M, int N> ESFM_HD void lstsq(const double *A, const double *b, double *x)
{
    double G[N * N], V[N * N], g[N];
    for (int i = 0; i < N; ++i) {
        g[i] = 0.0;
        for (int k = 0; k < M; ++k) g[i] += A[k * N + i] * b[k];
        for (int j = 0; j < N; ++j) { double s = 0.0; for (int k = 0; k < M; ++k) s += A[k * N + i] * A[k * N + j]; G[i * N + j] = s; }
    }
    jacobi_sym<N>(G, V);
    double mx = 0.0;
    for (int i = 0; i < N; ++i) mx = fmax(mx, G[i * N + i]);
    for (int i = 0; i < N; ++i) x[i] = 0.0;
    for (int k = 0; k < N; ++k) {
        const double ev = G[k * N + k];
        if (!(ev > mx * 1e-28)) continue;
        double c = 0.0;
        for (int i = 0; i < N; ++i) c += V[i * N + k] * g[i];
        c /= ev;
        for (int i = 0; i < N; ++i) x[i] += c * V[i * N + k];
    }
}

ESFM_HD double det3(const double *M) { return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]); }

// choose_control_points + the inverse used by compute_barycentric_coordinates.  sum_pw = sum of the points, sum_pwpw = sum of
// pw pw' (raw second moments, row-major 3 x 3).
// (the tail both forms share: cws[0] = the centroid and C = the scatter matrix about it are in)
ESFM_HD void control_points_from_scatter(const double C[9], int n, double cws[4][3], double CCi[9]);
ESFM_HD void control_points(const double sum_pw[3], const double sum_pwpw[9], int n, double cws[4][3], double CCi[9])
{
    for (int j = 0; j < 3; ++j) cws[0][j] = sum_pw[j] / n;
    double C[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[3 * a + b] = sum_pwpw[3 * a + b] - n * cws[0][a] * cws[0][b];
    control_points_from_scatter(C, n, cws, CCi);
}
// The same from the points themselves, the scatter matrix summed about the centroid in the order the CPU restatement sums it
// (oracle/pnp_ref.c epnp_pose): what the RANSAC hypotheses use.  The moment form above is for the re-fit, whose sums over thousands of
// correspondences are device reductions and agree with a serial loop to rounding only; a 5-point hypothesis has no such excuse, and
// EPnP amplifies a last-bit difference of C into pose differences that move threshold-borderline correspondences in and out of the
// inlier set (a random sweep against the oracle: 3.7 % of the problems picked another of two nearly equal hypotheses).
ESFM_HD void control_points_centred_n(const double *pws, int n, double cws[4][3], double CCi[9])
{
    for (int j = 0; j < 3; ++j) cws[0][j] = 0.0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) cws[0][j] += pws[3 * i + j];
    for (int j = 0; j < 3; ++j) cws[0][j] /= n;
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[3 * a + b] += (pws[3 * i + a] - cws[0][a]) * (pws[3 * i + b] - cws[0][b]);
    control_points_from_scatter(C, n, cws, CCi);
}
template <int K> ESFM_HD void control_points_centred(const double *pws, double cws[4][3], double CCi[9]) { control_points_centred_n(pws, K, cws, CCi); }
ESFM_HD void control_points_from_scatter(const double C[9], int n, double cws[4][3], double CCi[9])
{
    double uct[9], dc[3];
    sym_eig_desc<3>(C, uct, dc);
    for (int i = 1; i < 4; ++i) { const double k = sqrt(fmax(dc[i - 1], 0.0) / n); for (int j = 0; j < 3; ++j) cws[i][j] = cws[0][j] + k * uct[3 * (i - 1) + j]; }
    double CC[9];
    for (int i = 0; i < 3; ++i) for (int j = 1; j < 4; ++j) CC[3 * i + j - 1] = cws[j][i] - cws[0][i];
    // cvInvert(&CC, &CC_inv, CV_SVD): the Moore-Penrose inverse, so that a coplanar point set (third axis of length 0) still
    // gets finite barycentric coordinates
    double G[9], V[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { G[3 * a + b] = 0.0; for (int k = 0; k < 3; ++k) G[3 * a + b] += CC[3 * k + a] * CC[3 * k + b]; }
    jacobi_sym<3>(G, V);
    const double mx = fmax(G[0], fmax(G[4], G[8]));
    for (int i = 0; i < 9; ++i) CCi[i] = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double ev = G[4 * k];
        if (!(ev > mx * 1e-24)) continue;
        double Av[3];
        for (int r = 0; r < 3; ++r) Av[r] = CC[3 * r] * V[k] + CC[3 * r + 1] * V[3 + k] + CC[3 * r + 2] * V[6 + k];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) CCi[3 * i + j] += V[3 * i + k] * Av[j] / ev;
    }
}

ESFM_HD void alphas_of(const double c0[3], const double CCi[9], const double pw[3], double a[4])
{
    const double e0 = pw[0] - c0[0], e1 = pw[1] - c0[1], e2 = pw[2] - c0[2];
    for (int j = 0; j < 3; ++j) a[1 + j] = CCi[3 * j] * e0 + CCi[3 * j + 1] * e1 + CCi[3 * j + 2] * e2;
    a[0] = 1.0 - a[1] - a[2] - a[3];
}

// the two rows of M this correspondence contributes (epnp::fill_M)
ESFM_HD void m_rows(const Cam &cam, const double as[4], double u, double v, double m1[12], double m2[12])
{
    for (int k = 0; k < 4; ++k) {
        m1[3 * k] = as[k] * cam.fu; m1[3 * k + 1] = 0.0; m1[3 * k + 2] = as[k] * (cam.uc - u);
        m2[3 * k] = 0.0; m2[3 * k + 1] = as[k] * cam.fv; m2[3 * k + 2] = as[k] * (cam.vc - v);
    }
}

// From M'M and the control points: the four null-space vectors v[k][12] and the three beta candidates (approximations 1-3, each
// after 5 Gauss-Newton steps): compute_L_6x10, compute_rho, find_betas_approx_{1,2,3}, gauss_newton of epnp.cpp.
template <class JAC>
ESFM_HD void betas_from_mtm_ws(const double MtM[144], const double cws[4][3], double v[4][12], double betas[3][4], double *ws /* 3 x 144: ut, A, V */, JAC jac)
{
    double *ut = ws, d[12];
    sym_eig_desc_ws<12>(MtM, ut, d, ws + 144, ws + 288, jac);
    EPNP_MARK(2);
    for (int k = 0; k < 4; ++k) for (int a = 0; a < 12; ++a) v[k][a] = ut[12 * (11 - k) + a];
    double dv[4][6][3], L[60], rho[6];
    for (int i = 0; i < 4; ++i) {
        int a = 0, b = 1;
        for (int j = 0; j < 6; ++j) {
            for (int k = 0; k < 3; ++k) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
            ++b; if (b > 3) { ++a; b = a + 1; }
        }
    }
    auto dot = [](const double *p, const double *q) { return p[0] * q[0] + p[1] * q[1] + p[2] * q[2]; };
    for (int i = 0; i < 6; ++i) {
        double *row = L + 10 * i;
        row[0] = dot(dv[0][i], dv[0][i]); row[1] = 2.0 * dot(dv[0][i], dv[1][i]); row[2] = dot(dv[1][i], dv[1][i]);
        row[3] = 2.0 * dot(dv[0][i], dv[2][i]); row[4] = 2.0 * dot(dv[1][i], dv[2][i]); row[5] = dot(dv[2][i], dv[2][i]);
        row[6] = 2.0 * dot(dv[0][i], dv[3][i]); row[7] = 2.0 * dot(dv[1][i], dv[3][i]); row[8] = 2.0 * dot(dv[2][i], dv[3][i]);
        row[9] = dot(dv[3][i], dv[3][i]);
    }
    {
        int a = 0, b = 1;
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) { const double e = cws[a][k] - cws[b][k]; s += e * e; }
            rho[j] = s; ++b; if (b > 3) { ++a; b = a + 1; }
        }
    }
    EPNP_MARK(3);
    {   // approximation 1: B11 B12 B13 B14
        double A[24], x[4];
        for (int i = 0; i < 6; ++i) { A[4 * i] = L[10 * i]; A[4 * i + 1] = L[10 * i + 1]; A[4 * i + 2] = L[10 * i + 3]; A[4 * i + 3] = L[10 * i + 6]; }
        lstsq<6, 4>(A, rho, x);
        double *b = betas[0];
        if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = -x[1] / b[0]; b[2] = -x[2] / b[0]; b[3] = -x[3] / b[0]; }
        else { b[0] = sqrt(x[0]); b[1] = x[1] / b[0]; b[2] = x[2] / b[0]; b[3] = x[3] / b[0]; }
    }
    {   // approximation 2: B11 B12 B22
        double A[18], x[3];
        for (int i = 0; i < 6; ++i) { A[3 * i] = L[10 * i]; A[3 * i + 1] = L[10 * i + 1]; A[3 * i + 2] = L[10 * i + 2]; }
        lstsq<6, 3>(A, rho, x);
        double *b = betas[1];
        if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
        else { b[0] = sqrt(x[0]); b[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
        if (x[1] < 0) b[0] = -b[0];
        b[2] = 0.0; b[3] = 0.0;
    }
    {   // approximation 3: B11 B12 B22 B13 B23
        double A[30], x[5];
        for (int i = 0; i < 6; ++i) for (int k = 0; k < 5; ++k) A[5 * i + k] = L[10 * i + k];
        lstsq<6, 5>(A, rho, x);
        double *b = betas[2];
        if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
        else { b[0] = sqrt(x[0]); b[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
        if (x[1] < 0) b[0] = -b[0];
        b[2] = x[3] / b[0]; b[3] = 0.0;
    }
    EPNP_MARK(4);
    for (int N = 0; N < 3; ++N) {
        double *b = betas[N];
        for (int it = 0; it < 5; ++it) {
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
            lstsq<6, 4>(A, rhs, x);
            for (int k = 0; k < 4; ++k) b[k] += x[k];
        }
    }
}

ESFM_HD void betas_from_mtm(const double MtM[144], const double cws[4][3], double v[4][12], double betas[3][4])
{
    double ws[3 * 144];
    betas_from_mtm_ws(MtM, cws, v, betas, ws, JacobiSerial<12>());
}

// camera-frame control points for one beta vector (compute_ccs)
ESFM_HD void ccs_of(const double b[4], const double v[4][12], double ccs[4][3])
{
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 3; ++j) ccs[c][j] = b[0] * v[0][3 * c + j] + b[1] * v[1][3 * c + j] + b[2] * v[2][3 * c + j] + b[3] * v[3][3 * c + j];
}

// estimate_R_and_t from the raw sums over the correspondences: sum_pc, sum_pw (3), sum_pcpw = sum pc pw' (3 x 3)
ESFM_HD void rt_from_cross_covariance(const double pc0[3], const double pw0[3], const double ABt[9], double R[9], double t[3]);
ESFM_HD void rt_from_sums(int n, const double sum_pc[3], const double sum_pw[3], const double sum_pcpw[9], double R[9], double t[3])
{
    double pc0[3], pw0[3], ABt[9];
    for (int j = 0; j < 3; ++j) { pc0[j] = sum_pc[j] / n; pw0[j] = sum_pw[j] / n; }
    for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) ABt[3 * j + k] = sum_pcpw[3 * j + k] - n * pc0[j] * pw0[k];
    rt_from_cross_covariance(pc0, pw0, ABt, R, t);
}
// estimate_R_and_t from the points themselves, centred sums in the CPU restatement's order (the hypotheses; see control_points_centred)
ESFM_HD void rt_centred_n(const double *pcs /* n x 3 */, const double *pws /* n x 3 */, int n, double R[9], double t[3])
{
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0}, ABt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
    for (int j = 0; j < 3; ++j) { pc0[j] /= n; pw0[j] /= n; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) ABt[3 * j + k] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + k] - pw0[k]);
    rt_from_cross_covariance(pc0, pw0, ABt, R, t);
}
template <int K> ESFM_HD void rt_centred(const double *pcs /* K x 3 */, const double *pws /* K x 3 */, double R[9], double t[3]) { rt_centred_n(pcs, pws, K, R, t); }
ESFM_HD void rt_from_cross_covariance(const double pc0[3], const double pw0[3], const double ABt[9], double R[9], double t[3])
{
    double G[9], V[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { G[3 * a + b] = 0.0; for (int k = 0; k < 3; ++k) G[3 * a + b] += ABt[3 * k + a] * ABt[3 * k + b]; }
    jacobi_sym<3>(G, V);
    int o[3] = {0, 1, 2};
    for (int i = 1; i < 3; ++i) { const int vv = o[i]; int j = i - 1; while (j >= 0 && G[4 * o[j]] < G[4 * vv]) { o[j + 1] = o[j]; --j; } o[j + 1] = vv; }
    double v[3][3], u[3][3], sig[3];
    for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) v[k][a] = V[3 * a + o[k]];
    for (int k = 0; k < 3; ++k) {
        double nn = 0.0;
        for (int a = 0; a < 3; ++a) { u[k][a] = ABt[3 * a] * v[k][0] + ABt[3 * a + 1] * v[k][1] + ABt[3 * a + 2] * v[k][2]; nn += u[k][a] * u[k][a]; }
        sig[k] = sqrt(nn);
        if (k < 2 || sig[2] > 1e-12 * sig[0]) for (int a = 0; a < 3; ++a) u[k][a] /= sig[k];
    }
    if (!(sig[2] > 1e-12 * sig[0])) {   // coplanar set: the third pair is the right-handed completion
        u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1]; u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2]; u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
        const double w0 = v[0][1] * v[1][2] - v[0][2] * v[1][1], w1 = v[0][2] * v[1][0] - v[0][0] * v[1][2], w2 = v[0][0] * v[1][1] - v[0][1] * v[1][0];
        v[2][0] = w0; v[2][1] = w1; v[2][2] = w2;
    }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[3 * r + c] = u[0][r] * v[0][c] + u[1][r] * v[1][c] + u[2][r] * v[2][c];
    if (det3(R) < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }   // epnp.cpp negates the last row of a reflection
    for (int j = 0; j < 3; ++j) t[j] = pc0[j] - (R[3 * j] * pw0[0] + R[3 * j + 1] * pw0[1] + R[3 * j + 2] * pw0[2]);
}

ESFM_HD double reproj_dist(const Cam &cam, const double R[9], const double t[3], const double pw[3], double u, double v)
{
    const double Xc = R[0] * pw[0] + R[1] * pw[1] + R[2] * pw[2] + t[0], Yc = R[3] * pw[0] + R[4] * pw[1] + R[5] * pw[2] + t[1];
    const double inv = 1.0 / (R[6] * pw[0] + R[7] * pw[1] + R[8] * pw[2] + t[2]);
    const double ue = cam.uc + cam.fu * Xc * inv, ve = cam.vc + cam.fv * Yc * inv;
    return sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
}

// epnp::compute_pose for K points held by one thread (the RANSAC kernel: K = 5)
template <int K, class JAC> ESFM_HD double solve_small_ws(const Cam &cam, const double *pws, const double *us, double R[9], double t[3], double *ws /* 4 x 144 doubles */, JAC jac)
{
    double cws[4][3], CCi[9];
    control_points_centred<K>(pws, cws, CCi);
    EPNP_MARK(0);
    double alphas[K][4];
    double *MtM = ws;
    for (int a = 0; a < 144; ++a) MtM[a] = 0.0;
    for (int i = 0; i < K; ++i) {
        alphas_of(cws[0], CCi, pws + 3 * i, alphas[i]);
        double m1[12], m2[12];
        m_rows(cam, alphas[i], us[2 * i], us[2 * i + 1], m1, m2);
        for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) MtM[12 * a + b] += m1[a] * m1[b] + m2[a] * m2[b];
    }
    double v[4][12], betas[3][4];
    EPNP_MARK(1);
    betas_from_mtm_ws(MtM, cws, v, betas, ws + 144, jac);
    EPNP_MARK(5);
    double best_err = 0.0;
    for (int N = 0; N < 3; ++N) {
        double ccs[4][3];
        ccs_of(betas[N], v, ccs);
        double pcs[K][3];
        for (int i = 0; i < K; ++i) for (int j = 0; j < 3; ++j) pcs[i][j] = alphas[i][0] * ccs[0][j] + alphas[i][1] * ccs[1][j] + alphas[i][2] * ccs[2][j] + alphas[i][3] * ccs[3][j];
        if (pcs[0][2] < 0.0) for (int i = 0; i < K; ++i) for (int j = 0; j < 3; ++j) pcs[i][j] = -pcs[i][j];   // solve_for_sign
        double Rn[9], tn[3];
        rt_centred<K>(&pcs[0][0], pws, Rn, tn);
        double e = 0.0;
        for (int i = 0; i < K; ++i) e += reproj_dist(cam, Rn, tn, pws + 3 * i, us[2 * i], us[2 * i + 1]);
        e /= K;
        if (N == 0 || e < best_err) { best_err = e; for (int a = 0; a < 9; ++a) R[a] = Rn[a]; for (int a = 0; a < 3; ++a) t[a] = tn[a]; }
    }
    EPNP_MARK(6);
    return best_err;
}
// epnp::compute_pose on n points from the points themselves, every sum serial and in index order: the CPU restatement's epnp_pose to the
// letter (the re-fit of a SMALL inlier set on the host, pnp_api.cpp: such sets are the ill-conditioned ones, where the device
// reductions' rounding was amplified into visibly different poses).  alphas: 4 n doubles of scratch, pcs: 3 n.
inline double solve_n(const Cam &cam, const double *pws, const double *us, int n, double *alphas, double *pcs, double R[9], double t[3])
{
    double cws[4][3], CCi[9], MtM[144];
    control_points_centred_n(pws, n, cws, CCi);
    for (int a = 0; a < 144; ++a) MtM[a] = 0.0;
    for (int i = 0; i < n; ++i) {
        alphas_of(cws[0], CCi, pws + 3 * i, alphas + 4 * i);
        double m1[12], m2[12];
        m_rows(cam, alphas + 4 * i, us[2 * i], us[2 * i + 1], m1, m2);
        for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) MtM[12 * a + b] += m1[a] * m1[b] + m2[a] * m2[b];
    }
    double v[4][12], betas[3][4];
    betas_from_mtm(MtM, cws, v, betas);
    double best_err = 0.0;
    for (int N = 0; N < 3; ++N) {
        double ccs[4][3];
        ccs_of(betas[N], v, ccs);
        for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) { const double *a = alphas + 4 * i; pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j]; }
        if (pcs[2] < 0.0) for (int i = 0; i < 3 * n; ++i) pcs[i] = -pcs[i];   // solve_for_sign
        double Rn[9], tn[3];
        rt_centred_n(pcs, pws, n, Rn, tn);
        double e = 0.0;
        for (int i = 0; i < n; ++i) e += reproj_dist(cam, Rn, tn, pws + 3 * i, us[2 * i], us[2 * i + 1]);
        e /= n;
        if (N == 0 || e < best_err) { best_err = e; for (int a = 0; a < 9; ++a) R[a] = Rn[a]; for (int a = 0; a < 3; ++a) t[a] = tn[a]; }
    }
    return best_err;
}
template <int K> ESFM_HD double solve_small(const Cam &cam, const double *pws, const double *us, double R[9], double t[3])
{
    double ws[4 * 144];
    return solve_small_ws<K>(cam, pws, us, R, t, ws, JacobiSerial<12>());
}

}  // namespace epnp
}  // namespace esfm

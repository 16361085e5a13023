/*
 * oracle/ransac_ref.c -- CPU restatement of the two-view geometric verification the reference runs on every matched pair:
 * cv::findEssentialMat(..., CV_RANSAC, prob, threshold, mask) followed by cv::recoverPose (reference
 * cpp_code/src/estimate_motion.cpp:49-67, called at cpp_code/test/sfm.cpp:165).  SURVEY.md section 8 row f-1.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include, link, call or execute this file
 * (see oracle/match_ref.c header).
 *
 * PARITY UNPINNED: OpenCV (>= 3, unpinned; author ran 3.4.2) is absent here and the reference holds no fixture.  Restated
 * from memory of OpenCV 3.4 [upstream modules/calib3d/src/five-point.cpp, ptsetreg.cpp, modules/core/src/rand.cpp]:
 *   findEssentialMat   points and camera matrix to double; x = (u - cx) / fx, y = (v - cy) / fy; threshold /= (fx + fy) / 2;
 *                      RANSAC point-set registrator with 5 model points, confidence `prob`, at most 1000 iterations
 *   RANSAC             RNG rng((uint64)-1): state = (unsigned)state * 4164903690 + (state >> 32), uniform(0, n) = next % n;
 *                      getSubset: 5 indices, a repeated index is redrawn; every model of the sample is scored with
 *                      findInliers (error <= (float)(threshold^2) on float errors); a model replaces the best iff
 *                      count > max(best, 4), then niters = RANSACUpdateNumIters(prob, outlier ratio, 5, niters)
 *   error              Sampson distance (x2'E x1)^2 / (|E x1|_xy^2 + |E' x2|_xy^2), double, stored as float
 *   5-point kernel     null space of the 5 x 9 epipolar system, the ten cubic constraints det E = 0 and
 *                      2 E E'E - tr(E E')E = 0 in the monomial order x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3
 *                      z2 z 1, Gauss-Jordan on the first ten columns, rows (4,5) (6,7) (8,9) combined to a 3 x 3 polynomial
 *                      matrix in z whose determinant is the degree-10 polynomial; its real roots (|imag| <= 1e-10) give z,
 *                      the null vector of B(z) gives x, y; E = x E1 + y E2 + z E3 + E4, scaled to unit Frobenius norm.
 *                      Deviations: OpenCV tries the models of one sample in the order cv::solvePoly emits the roots, with
 *                      the sign its SVD null-space basis happens to give -- artefacts of its iteration that only decide
 *                      ties between models of the same sample.  Here every model is given a canonical sign (its
 *                      largest-magnitude entry positive) and the models of a sample are ordered by ascending E[0][0];
 *                      the polynomial roots come from a Durand-Kerner iteration written here, not cv::solvePoly itself.
 *   recoverPose        decomposeEssentialMat (SVD, det U, det V' forced positive, W = [0 1 0; -1 0 0; 0 0 1]); the four
 *                      (R, t) candidates in OpenCV's order (R1,t) (R2,t) (R1,-t) (R2,-t); per candidate every point is
 *                      triangulated in double against [I|0] and kept iff Z W > 0, Z / W < 50 in the first camera and
 *                      0 < Z < 50 in the second; masks are AND-ed with the RANSAC mask; the candidate with most points wins
 *                      (first in order on ties).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ----------------------------------------------------------------------------------------------- cv::RNG */
typedef struct { uint64_t state; } cv_rng;
static inline unsigned rng_next(cv_rng *r)
{
    r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
    return (unsigned)r->state;
}
static inline int rng_uniform(cv_rng *r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

/* exported: the index stream of getSubset, for the product-side test of the same generator */
void esfm_ref_ransac_samples(int count, int n_samples, int32_t *idx /* 5 per sample */)
{
    cv_rng rng = { 0xFFFFFFFFFFFFFFFFull };
    for (int s = 0; s < n_samples; ++s) {
        int id[5];
        for (int i = 0; i < 5;) {
            int v;
            for (;;) {
                v = id[i] = rng_uniform(&rng, 0, count);
                int j = 0;
                for (; j < i; ++j) if (v == id[j]) break;
                if (j == i) break;
            }
            ++i;
        }
        for (int i = 0; i < 5; ++i) idx[5 * s + i] = id[i];
    }
}

/* ----------------------------------------------------------------------------------------------- small linear algebra */
/* eigen-decomposition of a symmetric n x n matrix (n <= 9) by cyclic Jacobi; V columns = eigenvectors */
static void jacobi_eig(double *A, int n, double *V)
{
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i) { diag += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j]; }
        if (off <= 1e-40 * diag || off == 0.0) break;
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

/* ----------------------------------------------------------------------------------------------- trivariate cubics */
/* monomial index in the solver's column order */
static int mono_index(int a, int b, int c)   /* x^a y^b z^c */
{
    static const int tab[20][3] = { {3,0,0},{0,3,0},{2,1,0},{1,2,0},{2,0,1},{2,0,0},{0,2,1},{0,2,0},{1,1,1},{1,1,0},
                                    {1,0,2},{1,0,1},{1,0,0},{0,1,2},{0,1,1},{0,1,0},{0,0,3},{0,0,2},{0,0,1},{0,0,0} };
    for (int i = 0; i < 20; ++i) if (tab[i][0] == a && tab[i][1] == b && tab[i][2] == c) return i;
    return -1;
}
static const int kMono[20][3] = { {3,0,0},{0,3,0},{2,1,0},{1,2,0},{2,0,1},{2,0,0},{0,2,1},{0,2,0},{1,1,1},{1,1,0},
                                  {1,0,2},{1,0,1},{1,0,0},{0,1,2},{0,1,1},{0,1,0},{0,0,3},{0,0,2},{0,0,1},{0,0,0} };
typedef struct { double c[20]; } poly3;
static poly3 p_zero(void) { poly3 p; memset(&p, 0, sizeof(p)); return p; }
static poly3 p_lin(double x, double y, double z, double w)
{
    poly3 p = p_zero();
    p.c[mono_index(1, 0, 0)] = x; p.c[mono_index(0, 1, 0)] = y; p.c[mono_index(0, 0, 1)] = z; p.c[mono_index(0, 0, 0)] = w;
    return p;
}
static poly3 p_add(poly3 a, poly3 b) { for (int i = 0; i < 20; ++i) a.c[i] += b.c[i]; return a; }
static poly3 p_sub(poly3 a, poly3 b) { for (int i = 0; i < 20; ++i) a.c[i] -= b.c[i]; return a; }
static poly3 p_scale(poly3 a, double s) { for (int i = 0; i < 20; ++i) a.c[i] *= s; return a; }
static poly3 p_mul(poly3 a, poly3 b)
{
    poly3 r = p_zero();
    for (int i = 0; i < 20; ++i) {
        if (a.c[i] == 0.0) continue;
        for (int j = 0; j < 20; ++j) {
            if (b.c[j] == 0.0) continue;
            const int e0 = kMono[i][0] + kMono[j][0], e1 = kMono[i][1] + kMono[j][1], e2 = kMono[i][2] + kMono[j][2];
            if (e0 + e1 + e2 > 3) continue;   /* never happens for the products formed below */
            r.c[mono_index(e0, e1, e2)] += a.c[i] * b.c[j];
        }
    }
    return r;
}

/* ----------------------------------------------------------------------------------------------- polynomial roots */
/* all complex roots of c[0] + c[1] z + ... + c[n] z^n by Durand-Kerner; returns the real ones (|imag| <= 1e-10), ascending */
static int real_roots(const double *c, int n, double *out)
{
    while (n > 0 && c[n] == 0.0) --n;
    if (n <= 0) return 0;
    double re[16], im[16];
    /* start on a circle of the Cauchy bound radius */
    double bound = 0.0;
    for (int i = 0; i < n; ++i) bound = fmax(bound, fabs(c[i] / c[n]));
    bound = 1.0 + bound;
    for (int i = 0; i < n; ++i) { const double a = 2.0 * 3.14159265358979323846 * i / n + 0.4; re[i] = 0.5 * bound * cos(a) * pow(0.9, i); im[i] = 0.5 * bound * sin(a) * pow(0.9, i); }
    for (int it = 0; it < 2000; ++it) {
        double move = 0.0;
        for (int i = 0; i < n; ++i) {
            double pr = c[n], pi = 0.0;                               /* p(z_i) / c[n] by Horner */
            for (int k = n - 1; k >= 0; --k) { const double t = pr * re[i] - pi * im[i] + c[k]; pi = pr * im[i] + pi * re[i]; pr = t; }
            double dr = c[n], di = 0.0;                               /* c[n] * prod (z_i - z_j) */
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                const double ar = re[i] - re[j], ai = im[i] - im[j];
                const double t = dr * ar - di * ai; di = dr * ai + di * ar; dr = t;
            }
            const double den = dr * dr + di * di;
            if (den == 0.0) continue;
            const double qr = (pr * dr + pi * di) / den, qi = (pi * dr - pr * di) / den;
            re[i] -= qr; im[i] -= qi;
            move = fmax(move, fabs(qr) + fabs(qi));
        }
        if (move <= 1e-15 * bound) break;
    }
    /* polish the near-real ones with Newton on the real axis */
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (fabs(im[i]) > 1e-10 * fmax(1.0, fabs(re[i]))) continue;
        double z = re[i];
        for (int it = 0; it < 3; ++it) {
            double p = c[n], d = 0.0;
            for (int k = n - 1; k >= 0; --k) { d = d * z + p; p = p * z + c[k]; }
            if (d == 0.0) break;
            z -= p / d;
        }
        out[m++] = z;
    }
    for (int i = 1; i < m; ++i) { double v = out[i]; int j = i - 1; while (j >= 0 && out[j] > v) { out[j + 1] = out[j]; --j; } out[j + 1] = v; }
    return m;
}

/* ----------------------------------------------------------------------------------------------- the 5-point kernel */
/* q1, q2: 5 normalised correspondences (x2' E x1 = 0).  E_out: up to 10 matrices, row-major, unit Frobenius norm. */
int esfm_ref_five_point(const double *q1, const double *q2, double *E_out)
{
    /* null space of the 5 x 9 system: eigenvectors of Q'Q with the 4 smallest eigenvalues */
    double QtQ[81]; memset(QtQ, 0, sizeof(QtQ));
    for (int i = 0; i < 5; ++i) {
        const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        const double row[9] = { x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0 };
        for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) QtQ[a * 9 + b] += row[a] * row[b];
    }
    double V[81];
    jacobi_eig(QtQ, 9, V);
    int order[9];
    for (int i = 0; i < 9; ++i) order[i] = i;
    for (int i = 1; i < 9; ++i) { int v = order[i], j = i - 1; while (j >= 0 && QtQ[order[j] * 9 + order[j]] > QtQ[v * 9 + v]) { order[j + 1] = order[j]; --j; } order[j + 1] = v; }
    double N[4][9];
    for (int k = 0; k < 4; ++k) for (int a = 0; a < 9; ++a) N[k][a] = V[a * 9 + order[k]];

    /* E(x, y, z) = x N0 + y N1 + z N2 + N3, entries are linear polynomials */
    poly3 E[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) E[r][c] = p_lin(N[0][3 * r + c], N[1][3 * r + c], N[2][3 * r + c], N[3][3 * r + c]);
    poly3 eq[10];
    /* det E */
    eq[0] = p_add(p_sub(p_mul(E[0][0], p_sub(p_mul(E[1][1], E[2][2]), p_mul(E[1][2], E[2][1]))),
                        p_mul(E[0][1], p_sub(p_mul(E[1][0], E[2][2]), p_mul(E[1][2], E[2][0])))),
                  p_mul(E[0][2], p_sub(p_mul(E[1][0], E[2][1]), p_mul(E[1][1], E[2][0]))));
    /* 2 E E'E - tr(E E') E */
    poly3 EEt[3][3], tr = p_zero();
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        EEt[r][c] = p_zero();
        for (int k = 0; k < 3; ++k) EEt[r][c] = p_add(EEt[r][c], p_mul(E[r][k], E[c][k]));
    }
    for (int r = 0; r < 3; ++r) tr = p_add(tr, EEt[r][r]);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        poly3 s = p_zero();
        for (int k = 0; k < 3; ++k) s = p_add(s, p_mul(EEt[r][k], E[k][c]));
        eq[1 + 3 * r + c] = p_sub(p_scale(s, 2.0), p_mul(tr, E[r][c]));
    }
    /* Gauss-Jordan on the first ten columns (partial pivoting) */
    double M[10][20];
    for (int r = 0; r < 10; ++r) for (int c = 0; c < 20; ++c) M[r][c] = eq[r].c[c];
    for (int col = 0; col < 10; ++col) {
        int piv = col; double best = fabs(M[col][col]);
        for (int r = col + 1; r < 10; ++r) if (fabs(M[r][col]) > best) { best = fabs(M[r][col]); piv = r; }
        if (best < 1e-300) return 0;
        if (piv != col) for (int c = 0; c < 20; ++c) { const double t = M[col][c]; M[col][c] = M[piv][c]; M[piv][c] = t; }
        const double inv = 1.0 / M[col][col];
        for (int c = 0; c < 20; ++c) M[col][c] *= inv;
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            const double f = M[r][col];
            if (f == 0.0) continue;
            for (int c = 0; c < 20; ++c) M[r][c] -= f * M[col][c];
        }
    }
    /* B (3 x 13): row(2i+4) - z row(2i+5); layout x: z^3..z^0 (4), y: z^3..z^0 (4), 1: z^4..z^0 (5) */
    double B[3][13];
    for (int i = 0; i < 3; ++i) {
        const double *a = &M[2 * i + 4][10], *b = &M[2 * i + 5][10];
        B[i][0] = -b[0]; B[i][1] = a[0] - b[1]; B[i][2] = a[1] - b[2]; B[i][3] = a[2];
        B[i][4] = -b[3]; B[i][5] = a[3] - b[4]; B[i][6] = a[4] - b[5]; B[i][7] = a[5];
        B[i][8] = -b[6]; B[i][9] = a[6] - b[7]; B[i][10] = a[7] - b[8]; B[i][11] = a[8] - b[9]; B[i][12] = a[9];
    }
    /* determinant polynomial (coefficients lowest degree first); p, q degree 3, r degree 4 */
    double P[3][4], Qp[3][4], R[3][5];
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 4; ++k) { P[i][k] = B[i][3 - k]; Qp[i][k] = B[i][7 - k]; }
        for (int k = 0; k < 5; ++k) R[i][k] = B[i][12 - k];
    }
    double det[11]; memset(det, 0, sizeof(det));
    static const int perm[6][3] = { {0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0} };
    static const int sign[6] = { 1, -1, -1, 1, 1, -1 };
    for (int s = 0; s < 6; ++s) {   /* sum over permutations: P[row a] Q[row b] R[row c] */
        const int a = perm[s][0], b = perm[s][1], c = perm[s][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 5; ++k) det[i + j + k] += sign[s] * P[a][i] * Qp[b][j] * R[c][k];
    }
    double zs[10];
    const int nz = real_roots(det, 10, zs);
    int count = 0;
    for (int t = 0; t < nz && count < 10; ++t) {
        const double z = zs[t], z2 = z * z, z3 = z2 * z, z4 = z3 * z;
        double Bz[9];
        for (int j = 0; j < 3; ++j) {
            Bz[3 * j] = B[j][0] * z3 + B[j][1] * z2 + B[j][2] * z + B[j][3];
            Bz[3 * j + 1] = B[j][4] * z3 + B[j][5] * z2 + B[j][6] * z + B[j][7];
            Bz[3 * j + 2] = B[j][8] * z4 + B[j][9] * z3 + B[j][10] * z2 + B[j][11] * z + B[j][12];
        }
        /* SVD::solveZ: right singular vector of the smallest singular value = eigenvector of Bz'Bz */
        double G[9], W[9];
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { G[3 * a + b] = 0; for (int k = 0; k < 3; ++k) G[3 * a + b] += Bz[3 * k + a] * Bz[3 * k + b]; }
        jacobi_eig(G, 3, W);
        int m = 0;
        for (int k = 1; k < 3; ++k) if (G[4 * k] < G[4 * m]) m = k;
        const double vx = W[m], vy = W[3 + m], vw = W[6 + m];
        const double vn = sqrt(vx * vx + vy * vy + vw * vw);
        if (fabs(vw / vn) < 1e-10) continue;
        const double x = vx / vw, y = vy / vw;
        double Ev[9], nrm = 0.0;
        for (int a = 0; a < 9; ++a) { Ev[a] = x * N[0][a] + y * N[1][a] + z * N[2][a] + N[3][a]; nrm += Ev[a] * Ev[a]; }
        nrm = sqrt(nrm);
        int big = 0;
        for (int a = 1; a < 9; ++a) if (fabs(Ev[a]) > fabs(Ev[big])) big = a;
        if (Ev[big] < 0) nrm = -nrm;                                   /* canonical sign */
        for (int a = 0; a < 9; ++a) E_out[9 * count + a] = Ev[a] / nrm;
        ++count;
    }
    /* canonical order: ascending E[0][0] */
    for (int i = 1; i < count; ++i) {
        double tmp[9]; memcpy(tmp, E_out + 9 * i, sizeof(tmp));
        int j = i - 1;
        while (j >= 0 && E_out[9 * j] > tmp[0]) { memcpy(E_out + 9 * (j + 1), E_out + 9 * j, sizeof(tmp)); --j; }
        memcpy(E_out + 9 * (j + 1), tmp, sizeof(tmp));
    }
    return count;
}

/* ----------------------------------------------------------------------------------------------- Sampson error */
static int find_inliers(const double *p1, const double *p2, int n, const double *E, double thresh, uint8_t *mask)
{
    const float t = (float)(thresh * thresh);
    int nz = 0;
    for (int i = 0; i < n; ++i) {
        const double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        const double Ex0 = E[0] * x1 + E[1] * y1 + E[2], Ex1 = E[3] * x1 + E[4] * y1 + E[5], Ex2 = E[6] * x1 + E[7] * y1 + E[8];
        const double Et0 = E[0] * x2 + E[3] * y2 + E[6], Et1 = E[1] * x2 + E[4] * y2 + E[7];
        const double x2tEx1 = x2 * Ex0 + y2 * Ex1 + Ex2;
        const float err = (float)(x2tEx1 * x2tEx1 / (Ex0 * Ex0 + Ex1 * Ex1 + Et0 * Et0 + Et1 * Et1));
        const int f = err <= t;
        mask[i] = (uint8_t)f; nz += f;
    }
    return nz;
}

static int ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = fmax(p, 0.0); p = fmin(p, 1.0);
    ep = fmax(ep, 0.0); ep = fmin(ep, 1.0);
    double num = fmax(1.0 - p, DBL_MIN);
    double denom = 1.0 - pow(1.0 - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* cv::findEssentialMat(points1, points2, K, RANSAC, prob, threshold, mask).  pts: n x 2 float pixels; K4 = fx, cx, fy, cy (float,
 * the reference passes frame 1's CV_32F K, estimate_motion.cpp:43-44).  E[9] row-major, mask[n].  Returns 1 on success;
 * *iters_run = RANSAC iterations executed. */
int esfm_ref_find_essential_ransac(const float *pts1, const float *pts2, int n, const float *K4, double prob, double threshold,
                                   double *E, uint8_t *mask, int32_t *iters_run, int32_t *best_count)
{
    if (iters_run) *iters_run = 0;
    if (best_count) *best_count = 0;
    if (n < 5) return 0;
    const double fx = (double)K4[0], cx = (double)K4[1], fy = (double)K4[2], cy = (double)K4[3];
    double *p1 = (double *)malloc(sizeof(double) * 2 * (size_t)n), *p2 = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        p1[2 * i] = ((double)pts1[2 * i] - cx) / fx; p1[2 * i + 1] = ((double)pts1[2 * i + 1] - cy) / fy;
        p2[2 * i] = ((double)pts2[2 * i] - cx) / fx; p2[2 * i + 1] = ((double)pts2[2 * i + 1] - cy) / fy;
    }
    threshold /= (fx + fy) / 2.0;
    uint8_t *cur = (uint8_t *)malloc((size_t)n);
    int niters = 1000, max_good = 0, ok = 0;
    cv_rng rng = { 0xFFFFFFFFFFFFFFFFull };
    double models[90];
    int iter = 0;
    if (n == 5) {
        const int nm = esfm_ref_five_point(p1, p2, models);
        if (nm > 0) { memcpy(E, models, sizeof(double) * 9); memset(mask, 1, (size_t)n); ok = 1; max_good = 5; }
    } else {
        for (iter = 0; iter < niters; ++iter) {
            int id[5];
            for (int i = 0; i < 5;) {
                int v;
                for (;;) {
                    v = id[i] = rng_uniform(&rng, 0, n);
                    int j = 0;
                    for (; j < i; ++j) if (v == id[j]) break;
                    if (j == i) break;
                }
                ++i;
            }
            double s1[10], s2[10];
            for (int i = 0; i < 5; ++i) { s1[2 * i] = p1[2 * id[i]]; s1[2 * i + 1] = p1[2 * id[i] + 1]; s2[2 * i] = p2[2 * id[i]]; s2[2 * i + 1] = p2[2 * id[i] + 1]; }
            const int nm = esfm_ref_five_point(s1, s2, models);
            for (int m = 0; m < nm; ++m) {
                const int good = find_inliers(p1, p2, n, models + 9 * m, threshold, cur);
                if (good > (max_good > 4 ? max_good : 4)) {
                    memcpy(mask, cur, (size_t)n);
                    memcpy(E, models + 9 * m, sizeof(double) * 9);
                    max_good = good;
                    niters = ransac_update_num_iters(prob, (double)(n - good) / n, 5, niters);
                }
            }
        }
        ok = max_good > 0;
    }
    if (iters_run) *iters_run = iter;
    if (best_count) *best_count = max_good;
    free(p1); free(p2); free(cur);
    return ok;
}

/* ----------------------------------------------------------------------------------------------- recoverPose */
static double det3(const double *M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* SVD of a 3x3 via the eigen-decomposition of E'E: E = U diag(s) V', singular values descending */
static void svd3(const double *E, double *U, double *s, double *Vt)
{
    double G[9], V[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { G[3 * a + b] = 0; for (int k = 0; k < 3; ++k) G[3 * a + b] += E[3 * k + a] * E[3 * k + b]; }
    jacobi_eig(G, 3, V);
    int o[3] = { 0, 1, 2 };
    for (int i = 1; i < 3; ++i) { int v = o[i], j = i - 1; while (j >= 0 && G[4 * o[j]] < G[4 * v]) { o[j + 1] = o[j]; --j; } o[j + 1] = v; }
    double v[3][3];
    for (int k = 0; k < 3; ++k) { s[k] = sqrt(fmax(G[4 * o[k]], 0.0)); for (int a = 0; a < 3; ++a) v[k][a] = V[3 * a + o[k]]; }
    double u[3][3];
    for (int k = 0; k < 2; ++k) {
        double nn = 0;
        for (int a = 0; a < 3; ++a) { u[k][a] = E[3 * a] * v[k][0] + E[3 * a + 1] * v[k][1] + E[3 * a + 2] * v[k][2]; nn += u[k][a] * u[k][a]; }
        nn = sqrt(nn);
        for (int a = 0; a < 3; ++a) u[k][a] /= nn;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1]; u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2]; u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    for (int a = 0; a < 3; ++a) for (int k = 0; k < 3; ++k) { U[3 * a + k] = u[k][a]; Vt[3 * k + a] = v[k][a]; }
}

static void mat3mul(const double *A, const double *B, double *C)
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += A[3 * r + k] * B[3 * k + c]; C[3 * r + c] = s; }
}

void esfm_ref_decompose_essential(const double *E, double *R1, double *R2, double *t)
{
    double U[9], s[3], Vt[9];
    svd3(E, U, s, Vt);
    if (det3(U) < 0) for (int i = 0; i < 9; ++i) U[i] = -U[i];
    if (det3(Vt) < 0) for (int i = 0; i < 9; ++i) Vt[i] = -Vt[i];
    const double W[9] = { 0, 1, 0, -1, 0, 0, 0, 0, 1 }, Wt[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 };
    double T[9];
    mat3mul(U, W, T); mat3mul(T, Vt, R1);
    mat3mul(U, Wt, T); mat3mul(T, Vt, R2);
    t[0] = U[2]; t[1] = U[5]; t[2] = U[8];
}

/* smallest right singular vector of the 4 x 4 DLT system in double */
static void triangulate_d(const double *P0, const double *P1, double x0, double y0, double x1, double y1, double *X)
{
    double A[4][4];
    for (int k = 0; k < 4; ++k) {
        A[0][k] = x0 * P0[8 + k] - P0[k]; A[1][k] = y0 * P0[8 + k] - P0[4 + k];
        A[2][k] = x1 * P1[8 + k] - P1[k]; A[3][k] = y1 * P1[8 + k] - P1[4 + k];
    }
    double G[16], V[16];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) { G[4 * a + b] = 0; for (int k = 0; k < 4; ++k) G[4 * a + b] += A[k][a] * A[k][b]; }
    jacobi_eig(G, 4, V);
    int m = 0;
    for (int k = 1; k < 4; ++k) if (G[5 * k] < G[5 * m]) m = k;
    for (int a = 0; a < 4; ++a) X[a] = V[4 * a + m];
}

/* cv::recoverPose(E, points1, points2, K, R, t, mask) with distanceThresh = 50.  mask[n] in/out.  Returns the number of
 * points that pass the cheirality check with the chosen pose. */
int esfm_ref_recover_pose(const double *E, const float *pts1, const float *pts2, int n, const float *K4, double *R, double *t, uint8_t *mask)
{
    const double fx = (double)K4[0], cx = (double)K4[1], fy = (double)K4[2], cy = (double)K4[3];
    double R1[9], R2[9], tt[3];
    esfm_ref_decompose_essential(E, R1, R2, tt);
    const double *Rs[4] = { R1, R2, R1, R2 };
    const double sg[4] = { 1, 1, -1, -1 };
    const double P0[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    uint8_t *masks = (uint8_t *)malloc(4 * (size_t)(n > 0 ? n : 1));
    int good[4] = { 0, 0, 0, 0 };
    for (int c = 0; c < 4; ++c) {
        double P[12];
        for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) P[4 * r + k] = Rs[c][3 * r + k]; P[4 * r + 3] = sg[c] * tt[r]; }
        for (int i = 0; i < n; ++i) {
            const double x0 = ((double)pts1[2 * i] - cx) / fx, y0 = ((double)pts1[2 * i + 1] - cy) / fy;
            const double x1 = ((double)pts2[2 * i] - cx) / fx, y1 = ((double)pts2[2 * i + 1] - cy) / fy;
            double Q[4];
            triangulate_d(P0, P, x0, y0, x1, y1, Q);
            int m = Q[2] * Q[3] > 0;
            const double X = Q[0] / Q[3], Y = Q[1] / Q[3], Z = Q[2] / Q[3];
            m = m && (Z < 50.0);
            const double Z2 = P[8] * X + P[9] * Y + P[10] * Z + P[11];
            m = m && (Z2 > 0) && (Z2 < 50.0);
            if (mask) m = m && mask[i];
            masks[(size_t)c * n + i] = (uint8_t)m; good[c] += m;
        }
    }
    int best;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) best = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) best = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) best = 2;
    else best = 3;
    memcpy(R, Rs[best], sizeof(double) * 9);
    for (int r = 0; r < 3; ++r) t[r] = sg[best] * tt[r];
    if (mask) memcpy(mask, masks + (size_t)best * n, (size_t)n);
    const int g = good[best];
    free(masks);
    return g;
}

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
 *   5-point kernel     EMEstimatorCallback::runKernel: see "the 5-point kernel" below -- the steps, which of them follow OpenCV and
 *                      which are this build's own rule (model order and sign within a sample, the root iteration), and why the
 *                      product's kernels and this file share one arithmetic
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

/* ----------------------------------------------------------------------------------------------- the 5-point kernel */
/* EMEstimatorCallback::runKernel [upstream five-point.cpp], ONE arithmetic for both sides since round 6: the product's kernels
 * (easysfm_amd/csrc/five_point_core.hpp, run by essential_setup_kernel / essential_roots_kernel) and this file evaluate the same
 * expressions in the same order, so a hypothesis' models agree to the bit and so do inlier counts, iteration counts and masks
 * (until round 5 the two sides took different routes to the same models -- null space by elimination against eigenvectors of Q'Q, two
 * different root iterations -- and 2.2 % of random RANSAC problems were decided differently by threshold-borderline correspondences).
 * The steps, and whose rule each one is:
 *   1  null space of the 5 x 9 epipolar system: OpenCV takes the last four right singular vectors (an orthonormal basis).  Here: the
 *      last four columns of the orthogonal factor of a Householder QR of the system's transpose -- also orthonormal, and no
 *      squaring of the condition number as with eigenvectors of Q'Q.  The models do not depend on the basis in exact arithmetic.
 *   2  the ten cubic constraints det E = 0, 2 E E'E - tr(E E')E = 0 in the monomial order
 *      x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1 (OpenCV's), products accumulated in the order written below
 *   3  Gauss-Jordan with partial pivoting on the first ten columns (OpenCV's), rows (4,5) (6,7) (8,9) -> B(z), det B(z) (degree 10)
 *   4  roots: cv::solvePoly is a Durand-Kerner iteration; so is this (own start points, simultaneous update, own stopping rule --
 *      the build's rule: OpenCV's root order and accuracy are artefacts of its iteration)
 *   5  per real root: x, y from the null vector of B(z) (cross product of two rows; OpenCV: SVD::solveZ), E = x E1 + y E2 + z E3 + E4,
 *      unit Frobenius norm, canonical sign (largest-magnitude entry positive), a sample's models in ascending (E[0][0], z) order
 *      (the build's rule: OpenCV tries them in solvePoly's order with its SVD's sign, which only decides ties between models of the
 *      same sample).
 * Only + - * / sqrt, frexp / ldexp and comparisons: every operation is correctly rounded (or exact) on both sides; no pow, no
 * sin / cos (the start points come from a table), no fused multiply-add (both builds use -ffp-contract=off). */

/* linear monomials x y z 1; quadratic x2 y2 z2 xy xz yz x y z 1; cubic: the 20 columns above */
static const signed char kLinLin[4][4] = { {0, 3, 4, 6}, {3, 1, 5, 7}, {4, 5, 2, 8}, {6, 7, 8, 9} };          /* linear x linear -> quadratic */
static const signed char kQuadLin[10][4] = {                                                                  /* quadratic x linear -> cubic column */
    /* x2 */ {0, 2, 4, 5},   /* y2 */ {3, 1, 6, 7},   /* z2 */ {10, 13, 16, 17}, /* xy */ {2, 3, 8, 9}, /* xz */ {4, 8, 10, 11},
    /* yz */ {8, 6, 13, 14}, /* x  */ {5, 9, 11, 12}, /* y  */ {9, 7, 14, 15},   /* z  */ {11, 14, 17, 18}, /* 1 */ {12, 15, 18, 19} };

static void quad_mul_acc(const double *a, const double *b, double s, double *q)     /* q += s a b   (linear x linear) */
{
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) q[kLinLin[i][j]] += s * a[i] * b[j];
}
static void cubic_mul_acc(const double *q, const double *l, double s, double *c)    /* c += s q l   (quadratic x linear) */
{
    for (int i = 0; i < 10; ++i) { const double qi = s * q[i]; for (int j = 0; j < 4; ++j) c[kQuadLin[i][j]] += qi * l[j]; }
}

/* step 1: N[k][9], k < 4: an orthonormal basis of the null space of the 5 x 9 system whose row i is
 * (x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1) of correspondence i */
static void null_space(const double *q1, const double *q2, double N[4][9])
{
    double a[5][9], beta[5];                          /* a[c] = row c of the system = column c of its transpose */
    for (int i = 0; i < 5; ++i) {
        const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        a[i][0] = x2 * x1; a[i][1] = x2 * y1; a[i][2] = x2; a[i][3] = y2 * x1; a[i][4] = y2 * y1; a[i][5] = y2; a[i][6] = x1; a[i][7] = y1; a[i][8] = 1.0;
    }
    for (int j = 0; j < 5; ++j) {                     /* reflector j: v = x - alpha e_j on rows j..8, kept in a[j][j..8]; H = I - beta v v' */
        double s = 0.0;
        for (int r = j; r < 9; ++r) s += a[j][r] * a[j][r];
        const double nrm = sqrt(s), x0 = a[j][j];
        const double v0 = x0 - (x0 >= 0.0 ? -nrm : nrm);
        a[j][j] = v0;
        double vtv = v0 * v0;
        for (int r = j + 1; r < 9; ++r) vtv += a[j][r] * a[j][r];
        beta[j] = vtv > 0.0 ? 2.0 / vtv : 0.0;
        for (int c = j + 1; c < 5; ++c) {
            double d = 0.0;
            for (int r = j; r < 9; ++r) d += a[j][r] * a[c][r];
            const double f = beta[j] * d;
            for (int r = j; r < 9; ++r) a[c][r] -= f * a[j][r];
        }
    }
    for (int k = 0; k < 4; ++k) {                     /* column 5 + k of H0 H1 H2 H3 H4 */
        double n[9];
        for (int r = 0; r < 9; ++r) n[r] = r == 5 + k ? 1.0 : 0.0;
        for (int j = 4; j >= 0; --j) {
            double d = 0.0;
            for (int r = j; r < 9; ++r) d += a[j][r] * n[r];
            const double f = beta[j] * d;
            for (int r = j; r < 9; ++r) n[r] -= f * a[j][r];
        }
        for (int r = 0; r < 9; ++r) N[k][r] = n[r];
    }
}

/* steps 2 + 3: det[11] (lowest degree first) and the rows P[3][4], Qp[3][4], R[3][5] of B(z) = [P(z) Qp(z) R(z)]; 0 = degenerate */
static int determinant_polynomial(double N[4][9], double *det, double P[3][4], double Qp[3][4], double R[3][5])
{
    double L[3][3][4];                                /* E(x, y, z) = x N0 + y N1 + z N2 + N3: entry (r, c) as a linear polynomial */
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) for (int k = 0; k < 4; ++k) L[r][c][k] = N[k][3 * r + c];
    double G[3][3][10], tr[10];                       /* G = E E' (entry (r, c) = sum_k L[min][k] L[max][k]), tr = its trace */
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        const int lo = r < c ? r : c, hi = r < c ? c : r;
        for (int k = 0; k < 10; ++k) G[r][c][k] = 0.0;
        for (int k = 0; k < 3; ++k) quad_mul_acc(L[lo][k], L[hi][k], 1.0, G[r][c]);
    }
    for (int k = 0; k < 10; ++k) tr[k] = G[0][0][k] + G[1][1][k] + G[2][2][k];
    double M[10][20];
    memset(M, 0, sizeof(M));
    for (int c = 0; c < 3; ++c) {                     /* row 0: det E = sum_c E[0][c] cofactor(0, c) */
        const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
        double cof[10];
        for (int k = 0; k < 10; ++k) cof[k] = 0.0;
        quad_mul_acc(L[1][c1], L[2][c2], 1.0, cof);
        quad_mul_acc(L[1][c2], L[2][c1], -1.0, cof);
        cubic_mul_acc(cof, L[0][c], 1.0, M[0]);
    }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {      /* rows 1..9: 2 (E E') E - tr(E E') E */
        double *row = M[1 + 3 * r + c];
        for (int k = 0; k < 3; ++k) cubic_mul_acc(G[r][k], L[k][c], 2.0, row);
        cubic_mul_acc(tr, L[r][c], -1.0, row);
    }
    for (int col = 0; col < 10; ++col) {              /* Gauss-Jordan on the first ten columns, partial pivoting */
        int piv = col; double best = fabs(M[col][col]);
        for (int r = col + 1; r < 10; ++r) { const double v = fabs(M[r][col]); if (v > best) { best = v; piv = r; } }
        if (!(best > 1e-300)) return 0;
        if (piv != col) for (int c = 0; c < 20; ++c) { const double t = M[col][c]; M[col][c] = M[piv][c]; M[piv][c] = t; }
        const double inv = 1.0 / M[col][col];
        for (int c = col; c < 20; ++c) M[col][c] *= inv;
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            const double f = M[r][col];
            if (f == 0.0) continue;
            for (int c = col; c < 20; ++c) M[r][c] -= f * M[col][c];
        }
    }
    for (int i = 0; i < 3; ++i) {                     /* B row i = row(2i + 4) - z row(2i + 5) of the reduced system; lowest degree first */
        const double *a = &M[2 * i + 4][10], *b = &M[2 * i + 5][10];
        P[i][3] = -b[0]; P[i][2] = a[0] - b[1]; P[i][1] = a[1] - b[2]; P[i][0] = a[2];
        Qp[i][3] = -b[3]; Qp[i][2] = a[3] - b[4]; Qp[i][1] = a[4] - b[5]; Qp[i][0] = a[5];
        R[i][4] = -b[6]; R[i][3] = a[6] - b[7]; R[i][2] = a[7] - b[8]; R[i][1] = a[8] - b[9]; R[i][0] = a[9];
    }
    static const int perm[6][3] = { {0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0} };
    static const double sgn[6] = { 1, -1, -1, 1, 1, -1 };
    for (int k = 0; k < 11; ++k) det[k] = 0.0;
    for (int s = 0; s < 6; ++s) {                     /* sum over permutations: P[row a] Qp[row b] R[row c] */
        const int a = perm[s][0], b = perm[s][1], c = perm[s][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            const double pq = sgn[s] * P[a][i] * Qp[b][j];
            for (int k = 0; k < 5; ++k) det[i + j + k] += pq * R[c][k];
        }
    }
    return 1;
}

/* step 4.  An upper estimate of a^(1/k) from exact operations only: a itself, its square root, or the power of two (in quarter
 * steps) above the k-th root of the power of two above a */
static double root_upper(double a, int k)
{
    static const double quarter[4] = { 1.0, 0x1.306fe0a31b715p+0, 0x1.6a09e667f3bcdp+0, 0x1.ae89f995ad3adp+0 };      /* 2^(j/4) */
    if (!(a > 1e-300)) return 0.0;
    if (k == 1) return a;
    if (k == 2) return sqrt(a);
    int e;
    (void)frexp(a, &e);                               /* a = m 2^e, 1/2 <= m < 1 */
    const int num = 4 * e;
    int t = num / k;                                  /* ceil(4 e / k) */
    if (t * k < num) ++t;
    int q = t / 4;                                    /* floor(t / 4) */
    if (4 * q > t) --q;
    return ldexp(quarter[t - 4 * q], q);
}
/* start points: estimate i on the circle of radius (half of Fujiwara's bound) x 0.9^i at the angle 2 pi i / 10 + 0.4 (a spiral: no
 * symmetry of the polynomial can be a symmetry of the start) */
static const double kStartCos[10] = { 0x1.d7954e7dba2f8p-1, 0x1.08532eee8b103p-1, -0x1.5f2c08503a8c7p-4, -0x1.4f59d8cac4b95p-1, -0x1.f2b6774fec871p-1,
                                      -0x1.d7954e7dba2f9p-1, -0x1.08532eee8b101p-1, 0x1.5f2c08503a8dep-4, 0x1.4f59d8cac4b97p-1, 0x1.f2b6774fec871p-1 };
static const double kStartSin[10] = { 0x1.8ec3ae92b676bp-2, 0x1.b67e458544eb3p-1, 0x1.fe1d62c483ff6p-1, 0x1.82e3cb1245546p-1, 0x1.cf8b5a26ac140p-3,
                                      -0x1.8ec3ae92b6767p-2, -0x1.b67e458544eb5p-1, -0x1.fe1d62c483ff6p-1, -0x1.82e3cb1245544p-1, -0x1.cf8b5a26ac134p-3 };

/* the ten complex roots of z^10 + cc[9] z^9 + ... + cc[0] by the simultaneous (Jacobi-style) Durand-Kerner iteration: every sweep
 * moves all ten estimates from the previous sweep's values.  An estimate is at rest when its step is <= 1e-13 of its magnitude, or is
 * already < 1e-7 and has not halved for twelve sweeps (the evaluation noise of an ill-conditioned root); the iteration stops when all
 * ten are at rest, or after 300 sweeps.  Returns the number of sweeps. */
static int durand_kerner10(const double *cc, double *re, double *im)
{
    double rad = 0.0;
    for (int k = 1; k <= 10; ++k) rad = fmax(rad, root_upper(fabs(cc[10 - k]) * (k == 10 ? 0.5 : 1.0), k));
    if (!(rad > 1e-300)) rad = 1.0;
    for (int i = 0; i < 10; ++i) {
        double r = rad;
        for (int k = 0; k < i; ++k) r = r * 0.9;
        re[i] = r * kStartCos[i]; im[i] = r * kStartSin[i];
    }
    double best_mv[10]; int stale[10];
    for (int i = 0; i < 10; ++i) { best_mv[i] = 1e300; stale[i] = 0; }
    int it = 0;
    for (; it < 300; ++it) {
        double nre[10], nim[10];
        int moving = 0;
        for (int i = 0; i < 10; ++i) {
            double pr = 1.0, pim = 0.0;                               /* p(z_i) by Horner */
            for (int k = 9; k >= 0; --k) { const double tt = pr * re[i] - pim * im[i] + cc[k]; pim = pr * im[i] + pim * re[i]; pr = tt; }
            double dr = 1.0, di = 0.0;                                /* prod_{j != i} (z_i - z_j), j ascending */
            for (int j = 0; j < 10; ++j) {
                if (j == i) continue;
                const double ar = re[i] - re[j], ai = im[i] - im[j];
                const double tt = dr * ar - di * ai, ti = dr * ai + di * ar;
                dr = tt; di = ti;
            }
            const double den = dr * dr + di * di;
            const double inv = den > 0.0 ? 1.0 / den : 0.0;
            const double qr = (pr * dr + pim * di) * inv, qi = (pim * dr - pr * di) * inv;
            nre[i] = re[i] - qr; nim[i] = im[i] - qi;
            const double mv = (fabs(qr) + fabs(qi)) / (fabs(nre[i]) + fabs(nim[i]) + 1e-300);
            const int better = mv < 0.5 * best_mv[i];
            if (better) best_mv[i] = mv;
            stale[i] = better ? 0 : stale[i] + 1;
            if (!(mv <= 1e-13) && !(stale[i] >= 12 && best_mv[i] < 1e-7)) moving = 1;
        }
        for (int i = 0; i < 10; ++i) { re[i] = nre[i]; im[i] = nim[i]; }
        if (!moving) { ++it; break; }
    }
    return it;
}

/* step 5 for root estimate (re, im): 1 and the unit-norm, sign-canonical model in Ev, with its polished root in *z_out; 0 = no model */
static int model_from_root(const double *cc, double P[3][4], double Qp[3][4], double R[3][5], double N[4][9], double re, double im, double *Ev, double *z_out)
{
    const int is_real = !(fabs(im) > 1e-8 * fmax(1.0, fabs(re)));
    double z = re;
    for (int nit = 0; nit < 4; ++nit) {               /* Newton on the real axis */
        double pz = 1.0, dz = 0.0;
        for (int k = 9; k >= 0; --k) { dz = dz * z + pz; pz = pz * z + cc[k]; }
        if (dz == 0.0) break;
        z -= pz / dz;
    }
    double Bz[3][3];
    for (int j = 0; j < 3; ++j) {
        Bz[j][0] = ((P[j][3] * z + P[j][2]) * z + P[j][1]) * z + P[j][0];
        Bz[j][1] = ((Qp[j][3] * z + Qp[j][2]) * z + Qp[j][1]) * z + Qp[j][0];
        Bz[j][2] = (((R[j][4] * z + R[j][3]) * z + R[j][2]) * z + R[j][1]) * z + R[j][0];
    }
    double bx = 0, by = 0, bw = 0, bn = -1.0;         /* null vector of B(z): the largest of the cross products of rows (0,1) (1,2) (2,0) */
    for (int a = 0; a < 3; ++a) {
        const int b = (a + 1) % 3;
        const double cx = Bz[a][1] * Bz[b][2] - Bz[a][2] * Bz[b][1], cy = Bz[a][2] * Bz[b][0] - Bz[a][0] * Bz[b][2],
                     cw = Bz[a][0] * Bz[b][1] - Bz[a][1] * Bz[b][0];
        const double nn = cx * cx + cy * cy + cw * cw;
        if (nn > bn) { bn = nn; bx = cx; by = cy; bw = cw; }
    }
    if (!(is_real && bn > 0.0 && !(fabs(bw) < 1e-10 * sqrt(bn)))) return 0;
    const double x = bx / bw, y = by / bw;
    double nrm = 0.0;
    for (int a = 0; a < 9; ++a) { Ev[a] = x * N[0][a] + y * N[1][a] + z * N[2][a] + N[3][a]; nrm += Ev[a] * Ev[a]; }
    nrm = 1.0 / sqrt(nrm);
    int big = 0;
    for (int a = 1; a < 9; ++a) if (fabs(Ev[a]) > fabs(Ev[big])) big = a;
    if (Ev[big] < 0.0) nrm = -nrm;                    /* canonical sign */
    for (int a = 0; a < 9; ++a) Ev[a] *= nrm;
    *z_out = z;
    return 1;
}

/* q1, q2: 5 normalised correspondences (x2' E x1 = 0).  E_out: up to 10 matrices, row-major, unit Frobenius norm. */
int esfm_ref_five_point(const double *q1, const double *q2, double *E_out)
{
    double N[4][9], det[11], P[3][4], Qp[3][4], R[3][5];
    null_space(q1, q2, N);
    if (!determinant_polynomial(N, det, P, Qp, R)) return 0;
    if (!(fabs(det[10]) > 0.0)) return 0;
    double cc[10], re[10], im[10];
    for (int k = 0; k < 10; ++k) cc[k] = det[k] / det[10];
    durand_kerner10(cc, re, im);
    double Ev[10][9], z[10];
    int valid[10], count = 0;
    for (int i = 0; i < 10; ++i) { valid[i] = model_from_root(cc, P, Qp, R, N, re[i], im[i], Ev[i], &z[i]); count += valid[i]; }
    for (int i = 0; i < 10; ++i) {                    /* canonical order: rank of (E[0][0], z, i) among the models */
        if (!valid[i]) continue;
        int rank = 0;
        for (int j = 0; j < 10; ++j)
            if (valid[j] && (Ev[j][0] < Ev[i][0] || (Ev[j][0] == Ev[i][0] && (z[j] < z[i] || (z[j] == z[i] && j < i))))) ++rank;
        memcpy(E_out + 9 * rank, Ev[i], sizeof(double) * 9);
    }
    return count;
}

/* the intermediate stages of one sample, for stage-by-stage comparison with the product's host and device builds
 * (tests/test_five_point_stages.py): stages[0..35] = N, [36..46] = det, [47..58] = P, [59..70] = Qp, [71..85] = R, [86..95] = cc,
 * [96..105] = re, [106..115] = im.  Returns the sweeps of the root iteration, -1 = no polynomial. */
int esfm_ref_five_point_stages(const double *q1, const double *q2, double *stages)
{
    double N[4][9], det[11], P[3][4], Qp[3][4], R[3][5], cc[10], re[10], im[10];
    memset(stages, 0, sizeof(double) * 116);
    null_space(q1, q2, N);
    memcpy(stages, N, sizeof(N));
    if (!determinant_polynomial(N, det, P, Qp, R) || !(fabs(det[10]) > 0.0)) return -1;
    memcpy(stages + 36, det, sizeof(det)); memcpy(stages + 47, P, sizeof(P)); memcpy(stages + 59, Qp, sizeof(Qp)); memcpy(stages + 71, R, sizeof(R));
    for (int k = 0; k < 10; ++k) cc[k] = det[k] / det[10];
    const int sweeps = durand_kerner10(cc, re, im);
    memcpy(stages + 86, cc, sizeof(cc)); memcpy(stages + 96, re, sizeof(re)); memcpy(stages + 106, im, sizeof(im));
    return sweeps;
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

// The 5-point essential-matrix kernel (EMEstimatorCallback::runKernel [upstream opencv/modules/calib3d/src/five-point.cpp], called by
// cv::findEssentialMat at reference cpp_code/src/estimate_motion.cpp:49-51) as small routines that compile for the host and for gfx950,
// the way epnp_core.hpp serves the PnP row.  SURVEY.md section 8 row f-1.
//
// ONE arithmetic for both sides (round 6): oracle/ransac_ref.c restates these expressions to the letter, in the same order, so that a
// hypothesis' models agree with the CPU restatement to the BIT -- and with them inlier counts, iteration counts and masks.  (Rounds 1-5
// reached the same models along two routes -- null space by elimination here, eigenvectors of Q'Q there; two root iterations -- to
// ~1e-8, and tests/stress_essential.py found 2.2 % of random RANSAC problems decided differently by threshold-borderline
// correspondences.)  What that asks of this file:
//   * only + - * / sqrt, frexp / ldexp and comparisons: correctly rounded or exact on both sides.  No pow (the start radius used it),
//     no sin / cos (the start points come from a table), no fused multiply-add (the library is built -ffp-contract=off);
//   * a fixed order of every sum;
//   * results that do not depend on the launch shape: the sixteen lanes of essential_roots_kernel and the sequential host build
//     (five_point_models_host: what tests/test_five_point_stages.py checks against the oracle WITHOUT a GPU) run the same per-estimate
//     routines below.
// The steps (whose rule each one is: the header of oracle/ransac_ref.c's section "the 5-point kernel"):
//   null_space             orthonormal basis of the null space of the 5 x 9 epipolar system (Householder QR of its transpose)
//   determinant_polynomial the ten cubic constraints -> 10 x 20 system -> Gauss-Jordan -> B(z) -> det B(z), degree 10
//   dk_*                   its ten roots by the simultaneous Durand-Kerner iteration
//   model_from_root        z -> (x, y) from the null vector of B(z) -> E, unit norm, canonical sign
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

// (forced inline: the small routines take register arrays by pointer, and a call would put them in scratch memory)
#define ESFM_FP_HD __host__ __device__ __forceinline__

namespace esfm {
namespace fivept {

// ---- trivariate polynomial bookkeeping -------------------------------------------------------------------------------
// cubic monomials in the solver's column order: x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1
// quadratic order: x2 y2 z2 xy xz yz x y z 1        linear order: x y z 1
// (constexpr, not __constant__: with the loops unrolled every index below is a compile-time number, so the small arrays live in registers)
constexpr signed char kLinLin[4][4] = {      // product of two linear monomials -> quadratic index
    {0, 3, 4, 6}, {3, 1, 5, 7}, {4, 5, 2, 8}, {6, 7, 8, 9}};
constexpr signed char kQuadLin[10][4] = {    // quadratic monomial x linear monomial -> cubic column
    /* x2 */ {0, 2, 4, 5},   /* y2 */ {3, 1, 6, 7},   /* z2 */ {10, 13, 16, 17}, /* xy */ {2, 3, 8, 9}, /* xz */ {4, 8, 10, 11},
    /* yz */ {8, 6, 13, 14}, /* x  */ {5, 9, 11, 12}, /* y  */ {9, 7, 14, 15},   /* z  */ {11, 14, 17, 18}, /* 1 */ {12, 15, 18, 19}};

ESFM_FP_HD void quad_mul_acc(const double *a, const double *b, double s, double *q)   // q += s * a * b (linear x linear)
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) q[kLinLin[i][j]] += s * a[i] * b[j];
}

template <typename Row>
ESFM_FP_HD void cubic_mul_acc(const double *q, const double *l, double s, Row c)  // c += s * q * l (quadratic x linear)
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const double qi = s * q[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) c(kQuadLin[i][j]) += qi * l[j];
    }
}

// Where the solver's big arrays live: entry e of this hypothesis.  On the device one LDS column of `Stride` lanes per entry (the rows
// of the 10 x 20 system are swapped by data -- pivots -- which in registers means scratch memory: 1 400 scratch loads and 1 500 stores
// in the one-lane-per-hypothesis kernel of rounds 1-2, each a dependent memory round trip); on the host a plain array (Stride 1).
template <int Stride>
struct Store {
    double *base;
    ESFM_FP_HD double &operator()(int e) const { return base[e * Stride]; }
    ESFM_FP_HD Store at(int e) const { return Store{base + e * Stride}; }
};

// what determinant_polynomial leaves for the per-root stage: det[11], P[3][4], Qp[3][4], R[3][5], N[4][9] (86 values: they fit the 90
// doubles a hypothesis owns in the RANSAC's `models` array)
constexpr int kSetupDet = 0, kSetupP = 11, kSetupQ = 23, kSetupR = 35, kSetupN = 50, kSetupValues = 86;

// Orthonormal basis N[k][9], k < 4, of the null space of the 5 x 9 epipolar system (row i = (x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1)
// of correspondence i): the last four columns of the orthogonal factor of a Householder QR of its transpose.  `lds` entries 0..44 hold
// the system (a(c, r) = component r of row c; the reflectors overwrite it), 48..83 receive N.  No pivoting, no data-dependent index.
// (OpenCV: the last four right singular vectors -- also orthonormal; the models do not depend on the basis in exact arithmetic.
// Until round 5: Gauss-Jordan with complete pivoting here, a non-orthogonal basis.)
template <int Stride>
ESFM_FP_HD void null_space(const double *q1, const double *q2, Store<Stride> lds)
{
    // (everything indexed by a loop counter lives in `lds`, reflector scales in entries 84..88: rolled loops, no register arrays)
    auto a = [&](int c, int r) -> double & { return lds(9 * c + r); };
    auto beta = [&](int j) -> double & { return lds(84 + j); };
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        a(i, 0) = x2 * x1; a(i, 1) = x2 * y1; a(i, 2) = x2; a(i, 3) = y2 * x1; a(i, 4) = y2 * y1; a(i, 5) = y2; a(i, 6) = x1; a(i, 7) = y1; a(i, 8) = 1.0;
    }
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {                          // reflector j: v = x - alpha e_j on rows j..8, kept in a(j, j..8); H = I - beta v v'
        double s = 0.0;
        for (int r = j; r < 9; ++r) s += a(j, r) * a(j, r);
        const double nrm = sqrt(s), x0 = a(j, j);
        const double v0 = x0 - (x0 >= 0.0 ? -nrm : nrm);
        a(j, j) = v0;
        double vtv = v0 * v0;
        for (int r = j + 1; r < 9; ++r) vtv += a(j, r) * a(j, r);
        const double bj = vtv > 0.0 ? 2.0 / vtv : 0.0;
        beta(j) = bj;
#pragma unroll 1
        for (int c = j + 1; c < 5; ++c) {
            double d = 0.0;
            for (int r = j; r < 9; ++r) d += a(j, r) * a(c, r);
            const double f = bj * d;
            for (int r = j; r < 9; ++r) a(c, r) -= f * a(j, r);
        }
    }
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {                          // column 5 + k of H0 H1 H2 H3 H4, in place
        auto n = [&](int r) -> double & { return lds(48 + 9 * k + r); };
        for (int r = 0; r < 9; ++r) n(r) = r == 5 + k ? 1.0 : 0.0;
#pragma unroll 1
        for (int j = 4; j >= 0; --j) {
            double d = 0.0;
            for (int r = j; r < 9; ++r) d += a(j, r) * n(r);
            const double f = beta(j) * d;
            for (int r = j; r < 9; ++r) n(r) -= f * a(j, r);
        }
    }
}

// N (lds entries 48..83) -> w[0 .. 85] (see kSetup*); false = degenerate sample (a zero pivot).  `lds` needs 200 entries; N is read
// before the 10 x 20 system takes its place.
template <int Stride>
ESFM_FP_HD bool determinant_polynomial(double *w, Store<Stride> lds)
{
    // E(x, y, z) = x N0 + y N1 + z N2 + N3: entry (r, c) as the linear polynomial L[r][c][4]; N itself goes out to the per-root stage
    double L[3][3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) { L[r][c][k] = lds(48 + 9 * k + 3 * r + c); w[kSetupN + 9 * k + 3 * r + c] = L[r][c][k]; }
    // G = E E' (6 unique quadratic entries; (r, c) with r <= c is formed as sum_k L[r][k] L[c][k], in that operand order) and its trace
    auto gram = [&](int r, int c, double (&g)[10]) {
        const int lo = r < c ? r : c, hi = r < c ? c : r;
#pragma unroll
        for (int k = 0; k < 10; ++k) g[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) quad_mul_acc(L[lo][k], L[hi][k], 1.0, g);
    };
    double tr[10];
    {
        double g0[10], g1[10], g2[10];
        gram(0, 0, g0); gram(1, 1, g1); gram(2, 2, g2);
#pragma unroll
        for (int k = 0; k < 10; ++k) tr[k] = g0[k] + g1[k] + g2[k];
    }
    // the 10 x 20 system M (entry 20 r + c; the epipolar system and N are dead)
    for (int e = 0; e < 200; ++e) lds(e) = 0.0;
    // row 0: det E = sum_c E[0][c] * cofactor(0, c)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
        double cof[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) cof[k] = 0.0;
        quad_mul_acc(L[1][c1], L[2][c2], 1.0, cof);
        quad_mul_acc(L[1][c2], L[2][c1], -1.0, cof);
        cubic_mul_acc(cof, L[0][c], 1.0, lds.at(0));
    }
    // rows 1..9: 2 (E E') E - tr(E E') E   (row r of E E' formed when its three rows of M are)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double G[3][10];
#pragma unroll
        for (int k = 0; k < 3; ++k) gram(r, k, G[k]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const Store<Stride> row = lds.at(20 * (1 + 3 * r + c));
#pragma unroll
            for (int k = 0; k < 3; ++k) cubic_mul_acc(G[k], L[k][c], 2.0, row);
            cubic_mul_acc(tr, L[r][c], -1.0, row);
        }
    }
    auto M = [&](int r, int c) -> double & { return lds(20 * r + c); };
    // Gauss-Jordan on the first ten columns, partial pivoting (columns left of the pivot are never read again: not touched)
    for (int col = 0; col < 10; ++col) {
        int piv = col; double best = fabs(M(col, col));
        for (int r = col + 1; r < 10; ++r) { const double v = fabs(M(r, col)); if (v > best) { best = v; piv = r; } }
        if (!(best > 1e-300)) return false;
        if (piv != col) for (int c = 0; c < 20; ++c) { const double t = M(col, c); M(col, c) = M(piv, c); M(piv, c) = t; }
        const double inv = 1.0 / M(col, col);
        double prow[20];                                   // the pivot row in registers for the eliminations (c < col: not used)
#pragma unroll
        for (int c = 0; c < 20; ++c) { prow[c] = c >= col ? M(col, c) * inv : 0.0; if (c >= col) M(col, c) = prow[c]; }
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            const double f = M(r, col);
            if (f == 0.0) continue;
#pragma unroll
            for (int c = 0; c < 20; ++c) if (c >= col) M(r, c) -= f * prow[c];
        }
    }
    // B(z): rows (4,5), (6,7), (8,9); P, Qp degree 3 and R degree 4, lowest degree first
    double P[3][4], Qp[3][4], R[3][5];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double a[10], b[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) { a[k] = M(2 * i + 4, 10 + k); b[k] = M(2 * i + 5, 10 + k); }
        P[i][3] = -b[0]; P[i][2] = a[0] - b[1]; P[i][1] = a[1] - b[2]; P[i][0] = a[2];
        Qp[i][3] = -b[3]; Qp[i][2] = a[3] - b[4]; Qp[i][1] = a[4] - b[5]; Qp[i][0] = a[5];
        R[i][4] = -b[6]; R[i][3] = a[6] - b[7]; R[i][2] = a[7] - b[8]; R[i][1] = a[8] - b[9]; R[i][0] = a[9];
    }
    double det[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) det[k] = 0.0;
    constexpr int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    constexpr double sgn[6] = {1, -1, -1, 1, 1, -1};
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int a = perm[s][0], b = perm[s][1], c = perm[s][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double pq = sgn[s] * P[a][i] * Qp[b][j];
#pragma unroll
                for (int k = 0; k < 5; ++k) det[i + j + k] += pq * R[c][k];
            }
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) w[kSetupDet + k] = det[k];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { w[kSetupP + 4 * i + k] = P[i][k]; w[kSetupQ + 4 * i + k] = Qp[i][k]; }
#pragma unroll
        for (int k = 0; k < 5; ++k) w[kSetupR + 5 * i + k] = R[i][k];
    }
    return true;
}

// ---- the roots of z^10 + cc[9] z^9 + ... + cc[0]: per-estimate pieces of the simultaneous Durand-Kerner iteration -----------------
// An upper estimate of a^(1/k) from exact operations only: a itself, its square root, or the power of two (in quarter steps) above the
// k-th root of the power of two above a.  (pow(a, 1.0 / k) until round 5: device and host libraries round it differently.)
ESFM_FP_HD double root_upper(double a, int k)
{
    if (!(a > 1e-300)) return 0.0;
    if (k == 1) return a;
    if (k == 2) return sqrt(a);
    int e;
    (void)frexp(a, &e);                                    // a = m 2^e, 1/2 <= m < 1
    const int num = 4 * e;
    int t = num / k;                                       // ceil(4 e / k)
    if (t * k < num) ++t;
    int q = t / 4;                                         // floor(t / 4)
    if (4 * q > t) --q;
    const int rem = t - 4 * q;
    const double quarter = rem == 0 ? 1.0 : rem == 1 ? 0x1.306fe0a31b715p+0 : rem == 2 ? 0x1.6a09e667f3bcdp+0 : 0x1.ae89f995ad3adp+0;   // 2^(rem/4)
    return ldexp(quarter, q);
}
// start radius: half of Fujiwara's bound 2 max_k |c_{n-k}|^(1/k) (every root lies within the bound), the k-th roots estimated from
// above.  Cauchy's 1 + max |c_k|, used until round 3, is looser by orders of magnitude here, and the estimates approach from outside by
// a factor ~ 9/10 per sweep.
ESFM_FP_HD double start_radius(const double *cc)
{
    double rad = 0.0;
#pragma unroll
    for (int k = 1; k <= 10; ++k) rad = fmax(rad, root_upper(fabs(cc[10 - k]) * (k == 10 ? 0.5 : 1.0), k));
    return rad > 1e-300 ? rad : 1.0;
}
// estimate i starts on the circle of radius rad x 0.9^i at the angle 2 pi i / 10 + 0.4: a spiral around the origin (no symmetry of the
// polynomial can be a symmetry of the start).  cos / sin of the ten angles as constants.
ESFM_FP_HD void start_point(double rad, int i, double &re, double &im)
{
    constexpr double cs[10] = {0x1.d7954e7dba2f8p-1, 0x1.08532eee8b103p-1, -0x1.5f2c08503a8c7p-4, -0x1.4f59d8cac4b95p-1, -0x1.f2b6774fec871p-1,
                               -0x1.d7954e7dba2f9p-1, -0x1.08532eee8b101p-1, 0x1.5f2c08503a8dep-4, 0x1.4f59d8cac4b97p-1, 0x1.f2b6774fec871p-1};
    constexpr double sn[10] = {0x1.8ec3ae92b676bp-2, 0x1.b67e458544eb3p-1, 0x1.fe1d62c483ff6p-1, 0x1.82e3cb1245546p-1, 0x1.cf8b5a26ac140p-3,
                               -0x1.8ec3ae92b6767p-2, -0x1.b67e458544eb5p-1, -0x1.fe1d62c483ff6p-1, -0x1.82e3cb1245544p-1, -0x1.cf8b5a26ac134p-3};
    double r = rad, c = cs[0], s = sn[0];
#pragma unroll
    for (int k = 0; k < 9; ++k) { r = k < i ? r * 0.9 : r; c = k + 1 == i ? cs[k + 1] : c; s = k + 1 == i ? sn[k + 1] : s; }
    re = r * c; im = r * s;
}
ESFM_FP_HD void poly_eval(const double *cc, double re, double im, double &pr, double &pim)   // p(z), monic, by Horner
{
    pr = 1.0; pim = 0.0;
#pragma unroll
    for (int k = 9; k >= 0; --k) { const double tt = pr * re - pim * im + cc[k]; pim = pr * im + pim * re; pr = tt; }
}
ESFM_FP_HD void dk_times(double &dr, double &di, double ar, double ai)                        // (dr, di) *= (ar, ai)
{
    const double tt = dr * ar - di * ai, ti = dr * ai + di * ar;
    dr = tt; di = ti;
}
ESFM_FP_HD void dk_step(double pr, double pim, double dr, double di, double &qr, double &qi)  // p(z_i) / prod_{j != i} (z_i - z_j)
{
    const double den = dr * dr + di * di;
    const double inv = den > 0.0 ? 1.0 / den : 0.0;
    qr = (pr * dr + pim * di) * inv; qi = (pim * dr - pr * di) * inv;
}
// An estimate is at rest when its step is below 1e-13 of its magnitude -- or when it has reached the noise of its own evaluation: an
// ill-conditioned root of this degree-10 polynomial never gets its step under 1e-13 (5.4 % of the hypotheses used to run into the cap of
// 300 sweeps for that), it jitters at 1e-12 .. 1e-9 instead.  So: a step that is already small (< 1e-7) and has not halved for twelve
// sweeps is noise.  Linear convergence at a cluster (ratio (m - 1) / m <= 0.9 per sweep) halves within seven and goes on; the approach
// from the start circle has steps of ~0.1 and is not affected.  (re, im) = the estimate AFTER the step (qr, qi).
struct Rest { double best_mv = 1e300; int stale = 0; };
ESFM_FP_HD bool dk_moving(Rest &st, double qr, double qi, double re, double im)
{
    const double mv = (fabs(qr) + fabs(qi)) / (fabs(re) + fabs(im) + 1e-300);
    const bool better = mv < 0.5 * st.best_mv;
    st.best_mv = better ? mv : st.best_mv;
    st.stale = better ? 0 : st.stale + 1;
    return !(mv <= 1e-13) && !(st.stale >= 12 && st.best_mv < 1e-7);
}
constexpr int kMaxSweeps = 300;

// Root estimate (re, im) -> model: Newton on the real axis if it is real to 1e-8, x, y from the null vector of B(z) (the largest of the
// cross products of two of its rows; OpenCV: SVD::solveZ), E = x N0 + y N1 + z N2 + N3 scaled to unit Frobenius norm with its
// largest-magnitude entry positive.  Ev and z are written in every case (the device ranks lanes by them under `valid`).
ESFM_FP_HD bool model_from_root(const double *cc, const double *w, double re, double im, double *Ev, double &z)
{
    const bool is_real = !(fabs(im) > 1e-8 * fmax(1.0, fabs(re)));
    z = re;
    for (int nit = 0; nit < 4; ++nit) {
        double pz = 1.0, dz = 0.0;
#pragma unroll
        for (int k = 9; k >= 0; --k) { dz = dz * z + pz; pz = pz * z + cc[k]; }
        if (dz == 0.0) break;
        z -= pz / dz;
    }
    double Bz[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double *P = w + kSetupP + 4 * j, *Q = w + kSetupQ + 4 * j, *R = w + kSetupR + 5 * j;
        Bz[j][0] = ((P[3] * z + P[2]) * z + P[1]) * z + P[0];
        Bz[j][1] = ((Q[3] * z + Q[2]) * z + Q[1]) * z + Q[0];
        Bz[j][2] = (((R[4] * z + R[3]) * z + R[2]) * z + R[1]) * z + R[0];
    }
    double bx = 0, by = 0, bw = 0, bn = -1.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int b = (a + 1) % 3;
        const double cx = Bz[a][1] * Bz[b][2] - Bz[a][2] * Bz[b][1], cy = Bz[a][2] * Bz[b][0] - Bz[a][0] * Bz[b][2],
                     cw = Bz[a][0] * Bz[b][1] - Bz[a][1] * Bz[b][0];
        const double nn = cx * cx + cy * cy + cw * cw;
        if (nn > bn) { bn = nn; bx = cx; by = cy; bw = cw; }
    }
    const bool valid = is_real && bn > 0.0 && !(fabs(bw) < 1e-10 * sqrt(bn));
    const double x = bx / bw, y = by / bw;
    double nrm = 0.0;
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        Ev[a] = x * w[kSetupN + a] + y * w[kSetupN + 9 + a] + z * w[kSetupN + 18 + a] + w[kSetupN + 27 + a];
        nrm += Ev[a] * Ev[a];
    }
    nrm = 1.0 / sqrt(nrm);
    int big = 0;
#pragma unroll
    for (int a = 1; a < 9; ++a) if (fabs(Ev[a]) > fabs(Ev[big])) big = a;
    double ebig = Ev[0];
#pragma unroll
    for (int a = 1; a < 9; ++a) ebig = a == big ? Ev[a] : ebig;
    if (ebig < 0.0) nrm = -nrm;
#pragma unroll
    for (int a = 0; a < 9; ++a) Ev[a] *= nrm;
    return valid;
}
// a sample's models in ascending (E[0][0], z, estimate index) order -- a basis-independent rule (OpenCV's order is whatever cv::solvePoly
// and its SVD basis produce, and only decides ties between models of the same sample): does model j come before model i?
ESFM_FP_HD bool model_precedes(double e00_j, double z_j, int j, double e00_i, double z_i, int i)
{
    return e00_j < e00_i || (e00_j == e00_i && (z_j < z_i || (z_j == z_i && j < i)));
}

// The whole kernel for one sample on the host, estimate after estimate: what the sixteen lanes of essential_roots_kernel do side by
// side.  E_out[90]; returns the number of models (0: degenerate).  stages (or NULL): 116 doubles, the layout of
// esfm_ref_five_point_stages (oracle/ransac_ref.c); *sweeps (or NULL) = sweeps of the root iteration.
inline int five_point_models_host(const double *q1, const double *q2, double *E_out, double *stages, int *sweeps)
{
    double lds[200], w[kSetupValues];
    if (stages) for (int k = 0; k < 116; ++k) stages[k] = 0.0;
    if (sweeps) *sweeps = -1;
    null_space(q1, q2, Store<1>{lds});
    if (stages) for (int k = 0; k < 36; ++k) stages[k] = lds[48 + k];
    if (!determinant_polynomial(w, Store<1>{lds})) return 0;
    const double c10 = w[kSetupDet + 10];
    if (!(fabs(c10) > 0.0)) return 0;
    if (stages) for (int k = 0; k < 50; ++k) stages[36 + k] = w[k];
    double cc[10], re[10], im[10];
    for (int k = 0; k < 10; ++k) cc[k] = w[kSetupDet + k] / c10;
    const double rad = start_radius(cc);
    for (int i = 0; i < 10; ++i) start_point(rad, i, re[i], im[i]);
    Rest rest[10];
    int it = 0;
    for (bool active = true; it < kMaxSweeps && active; ++it) {
        double nre[10], nim[10];
        active = false;
        for (int i = 0; i < 10; ++i) {
            double pr, pim, dr = 1.0, di = 0.0, qr, qi;
            poly_eval(cc, re[i], im[i], pr, pim);
            for (int j = 0; j < 10; ++j) if (j != i) dk_times(dr, di, re[i] - re[j], im[i] - im[j]);
            dk_step(pr, pim, dr, di, qr, qi);
            nre[i] = re[i] - qr; nim[i] = im[i] - qi;
            if (dk_moving(rest[i], qr, qi, nre[i], nim[i])) active = true;
        }
        for (int i = 0; i < 10; ++i) { re[i] = nre[i]; im[i] = nim[i]; }
    }
    if (sweeps) *sweeps = it;
    if (stages) for (int k = 0; k < 10; ++k) { stages[86 + k] = cc[k]; stages[96 + k] = re[k]; stages[106 + k] = im[k]; }
    double Ev[10][9], z[10];
    bool valid[10];
    int count = 0;
    for (int i = 0; i < 10; ++i) { valid[i] = model_from_root(cc, w, re[i], im[i], Ev[i], z[i]); count += valid[i] ? 1 : 0; }
    for (int i = 0; i < 10; ++i) {
        if (!valid[i]) continue;
        int rank = 0;
        for (int j = 0; j < 10; ++j) rank += (valid[j] && model_precedes(Ev[j][0], z[j], j, Ev[i][0], z[i], i)) ? 1 : 0;
        for (int a = 0; a < 9; ++a) E_out[9 * rank + a] = Ev[i][a];
    }
    return count;
}

}  // namespace fivept
}  // namespace esfm

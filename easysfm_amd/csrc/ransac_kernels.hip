// Two-view geometric verification for gfx950 (MI355X): the device side of the replacement for
// cv::findEssentialMat(..., CV_RANSAC, ...) + cv::recoverPose as MotionEstimator::estimate2D2D_E5P_RANSAC calls them
// (reference cpp_code/src/estimate_motion.cpp:49-67; once per matched image pair, cpp_code/test/sfm.cpp:165).
// SURVEY.md section 8 row f-1.
//
// RANSAC hypotheses are data-parallel: the sample stream of cv::RNG is replayed on the host (it is a 64-bit
// multiply-with-carry recurrence, sequential by nature and a few microseconds long), every hypothesis of a chunk of
// iterations of every pair is then solved and scored concurrently, and the host replays OpenCV's sequential
// best-model / adaptive-iteration-count bookkeeping on the inlier counts -- the result is the one the sequential loop
// reaches, independent of the chunk size.
//
//   essential_setup_kernel   one thread per (pair, iteration): 5-point kernel up to the degree-10 polynomial
//   essential_roots_kernel   sixteen lanes per (pair, iteration): its roots -> up to 10 essential matrices
//   essential_score_kernel   one workgroup per (pair, iteration): Sampson inlier count of each of its models
//   essential_mask_kernel    one workgroup per pair: inlier mask of the winning model
//   pose_cheirality_kernel   one workgroup per (pair, candidate pose): f64 DLT triangulation + cheirality test per point
//
// All arithmetic in f64 like OpenCV; the error is rounded to float before the threshold test like
// RANSACPointSetRegistrator::findInliers.
#include "ransac_kernels.hpp"

#include <float.h>
#include <math.h>

namespace esfm {

// ---- trivariate polynomial bookkeeping -------------------------------------------------------------------------------
// cubic monomials in the solver's column order: x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1
// quadratic order: x2 y2 z2 xy xz yz x y z 1        linear order: x y z 1
// (constexpr, not __constant__: with the loops unrolled every index below is a compile-time number, so the small arrays live in registers)
__device__ constexpr signed char kLinLin[4][4] = {      // product of two linear monomials -> quadratic index
    {0, 3, 4, 6}, {3, 1, 5, 7}, {4, 5, 2, 8}, {6, 7, 8, 9}};
__device__ constexpr signed char kQuadLin[10][4] = {    // quadratic monomial x linear monomial -> cubic column
    /* x2 */ {0, 2, 4, 5},   /* y2 */ {3, 1, 6, 7},   /* z2 */ {10, 13, 16, 17}, /* xy */ {2, 3, 8, 9}, /* xz */ {4, 8, 10, 11},
    /* yz */ {8, 6, 13, 14}, /* x  */ {5, 9, 11, 12}, /* y  */ {9, 7, 14, 15},   /* z  */ {11, 14, 17, 18}, /* 1 */ {12, 15, 18, 19}};

__device__ __forceinline__ void quad_mul_acc(const double *a, const double *b, double s, double *q)   // q += s * a * b (linear x linear)
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) q[kLinLin[i][j]] += s * a[i] * b[j];
}

// The solver's big arrays live in LDS, one column of kSetupLanes lanes per entry (entry e of this lane: lds[e * kSetupLanes + lane]):
// their rows and columns are picked by data (pivots), which in registers means scratch memory -- 1 400 scratch loads and 1 500 stores
// in the one-lane-per-hypothesis kernel of rounds 1-2, each a dependent memory round trip.
constexpr int kSetupLanes = 32;
struct LdsVec {
    double *base;                                          // &lds[lane]
    __device__ __forceinline__ double &operator()(int e) const { return base[e * kSetupLanes]; }
};

template <typename Row>
__device__ __forceinline__ void cubic_mul_acc(const double *q, const double *l, double s, Row c)  // c += s * q * l (quadratic x linear)
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const double qi = s * q[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) c(kQuadLin[i][j]) += qi * l[j];
    }
}

// EMEstimatorCallback::runKernel [upstream five-point.cpp]: q1, q2 = 5 normalised correspondences -> up to 10 unit-norm essential
// matrices (row-major), each with its largest-magnitude entry positive, in ascending order of E[0][0] (a basis-independent order;
// OpenCV's is whatever cv::solvePoly and its SVD basis produce).
// Split in two since round 3 (see essential_roots_kernel): this part -- null space, the 10 x 20 elimination, B(z) and its determinant
// -- is one hypothesis per lane and leaves det[11], P[3][4], Qp[3][4], R[3][5], N[4][9] (86 values) in `w`; false = no model.
constexpr int kSetupDet = 0, kSetupP = 11, kSetupQ = 23, kSetupR = 35, kSetupN = 50;   // (86 values: they fit the 90 doubles a hypothesis owns in `models`)
__device__ bool five_point_setup(const double *q1, const double *q2, double *w, LdsVec lds)
{
    // null space of the 5 x 9 epipolar system by Gauss-Jordan with complete pivoting: 4 basis vectors N[k][9]
    auto Q = [&](int r, int c) -> double & { return lds(9 * r + c); };              // 45 entries (the 10 x 20 system takes their place later)
    for (int i = 0; i < 5; ++i) {
        const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        Q(i, 0) = x2 * x1; Q(i, 1) = x2 * y1; Q(i, 2) = x2; Q(i, 3) = y2 * x1; Q(i, 4) = y2 * y1; Q(i, 5) = y2; Q(i, 6) = x1; Q(i, 7) = y1; Q(i, 8) = 1.0;
    }
    int colperm = 0x876543210 & 0xffffffff;              // nine 4-bit column numbers packed (column 8 in `cp8`): no indexed int array either
    int cp8 = 8;
    auto cp_get = [&](int k) { return k == 8 ? cp8 : (colperm >> (4 * k)) & 15; };
    auto cp_set = [&](int k, int v) { if (k == 8) cp8 = v; else colperm = (colperm & ~(15 << (4 * k))) | (v << (4 * k)); };
    for (int k = 0; k < 5; ++k) {
        int pr = k, pc = k; double best = -1.0;
        for (int r = k; r < 5; ++r) for (int c = k; c < 9; ++c) { const double v = fabs(Q(r, c)); if (v > best) { best = v; pr = r; pc = c; } }
        if (!(best > 1e-300)) return false;
        for (int c = 0; c < 9; ++c) { const double t = Q(k, c); Q(k, c) = Q(pr, c); Q(pr, c) = t; }
        for (int r = 0; r < 5; ++r) { const double t = Q(r, k); Q(r, k) = Q(r, pc); Q(r, pc) = t; }
        { const int t = cp_get(k); cp_set(k, cp_get(pc)); cp_set(pc, t); }
        const double inv = 1.0 / Q(k, k);
        for (int c = 0; c < 9; ++c) Q(k, c) *= inv;
        for (int r = 0; r < 5; ++r) {
            if (r == k) continue;
            const double f = Q(r, k);
            for (int c = 0; c < 9; ++c) Q(r, c) -= f * Q(k, c);
        }
    }
    auto N = [&](int k, int a) -> double & { return lds(48 + 9 * k + a); };         // 36 entries behind Q
    for (int k = 0; k < 4; ++k) {
        double nn = 1.0;
        for (int a = 0; a < 9; ++a) N(k, a) = 0.0;
        N(k, cp_get(5 + k)) = 1.0;
        for (int r = 0; r < 5; ++r) { const double v = -Q(r, 5 + k); N(k, cp_get(r)) = v; nn += v * v; }
        nn = 1.0 / sqrt(nn);
        for (int a = 0; a < 9; ++a) N(k, a) *= nn;
    }
    // E(x, y, z) = x N0 + y N1 + z N2 + N3: entry (r, c) as the linear polynomial L[r][c][4]; N itself goes out to the roots kernel
    double L[3][3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) { L[r][c][k] = N(k, 3 * r + c); w[kSetupN + 9 * k + 3 * r + c] = L[r][c][k]; }
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
    // the 10 x 20 system M (LDS: entry 20 r + c; Q and N above are dead)
    auto Mrow = [&](int r) { return LdsVec{&lds(20 * r)}; };
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
        cubic_mul_acc(cof, L[0][c], 1.0, Mrow(0));
    }
    // rows 1..9: 2 (E E') E - tr(E E') E   (row r of E E' formed when its three rows of M are)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double G[3][10];
#pragma unroll
        for (int k = 0; k < 3; ++k) gram(r, k, G[k]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const LdsVec row = Mrow(1 + 3 * r + c);
#pragma unroll
            for (int k = 0; k < 3; ++k) cubic_mul_acc(G[k], L[k][c], 2.0, row);
            cubic_mul_acc(tr, L[r][c], -1.0, row);
        }
    }
    auto M = [&](int r, int c) -> double & { return lds(20 * r + c); };
    // Gauss-Jordan on the first ten columns
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

__device__ __forceinline__ void normalise_pt(const RansacPair &pr, const float2 *__restrict__ p1, const float2 *__restrict__ p2, int i,
                                             double &x1, double &y1, double &x2, double &y2)
{
    const float2 a = p1[pr.first + i], b = p2[pr.first + i];
    x1 = ((double)a.x - pr.cx) / pr.fx; y1 = ((double)a.y - pr.cy) / pr.fy;
    x2 = ((double)b.x - pr.cx) / pr.fx; y2 = ((double)b.y - pr.cy) / pr.fy;
}

// value of lane J of the caller's row of 16 lanes: one v_mov_b64_dpp (row_newbcast) instead of the two ds_bpermute_b32 and their LDS
// round trip that __shfl(v, J, 16) costs.  (The s_nop covers the DPP read-after-VALU-write hazard, which the compiler does not track
// through inline asm.)
#ifdef ESFM_DK_HIST
// timing / diagnosis build: histogram of Durand-Kerner sweeps per hypothesis (scratch/dk_hist.py)
__device__ unsigned int g_dk_hist[301];
extern "C" int esfm_debug_dk_hist(unsigned int *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dk_hist), sizeof(g_dk_hist)); }
#endif
template <int J>
__device__ __forceinline__ double row16_bcast(double v)
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
    return r;
}
template <int J>
__device__ __forceinline__ void dk_factor(double re, double im, int i, double &dr, double &di)
{
    const double ar = re - row16_bcast<J>(re), ai = im - row16_bcast<J>(im);
    const double tt = dr * ar - di * ai, ti = dr * ai + di * ar;
    dr = J == i ? dr : tt; di = J == i ? di : ti;
}

// one thread per (pair, iteration of this chunk): the polynomial system of its sample -> models[90 g ..] (86 values);
// n_models[g] = 1 when there is one, -1 when the slot is idle or the sample degenerate (essential_roots_kernel turns both into counts)
__global__ __launch_bounds__(kSetupLanes) void essential_setup_kernel(const RansacPair *__restrict__ pairs, int n_pairs, const float2 *__restrict__ p1,
                                                                      const float2 *__restrict__ p2, const int32_t *__restrict__ samples, int chunk,
                                                                      double *__restrict__ models, int32_t *__restrict__ n_models)
{
    __shared__ double lds[200 * kSetupLanes];             // 51 KB: three workgroups per CU, 19 200 hypotheses resident at once
    const int g = blockIdx.x * kSetupLanes + threadIdx.x;
    if (g >= n_pairs * chunk) return;
    const int pi = g / chunk;
    const RansacPair pr = pairs[pi];
    const int32_t *id = samples + 5 * (size_t)g;
    if (!pr.active || id[0] < 0) { n_models[g] = -1; return; }
    double q1[10], q2[10];
    for (int k = 0; k < 5; ++k) normalise_pt(pr, p1, p2, id[k], q1[2 * k], q1[2 * k + 1], q2[2 * k], q2[2 * k + 1]);
    n_models[g] = five_point_setup(q1, q2, models + 90 * (size_t)g, LdsVec{lds + threadIdx.x}) ? 1 : -1;
}

// Roots of the degree-10 determinant and the models they give: SIXTEEN LANES PER HYPOTHESIS, lane i = root estimate i (ten of them).
// Until round 3 this was the tail of a one-lane-per-hypothesis kernel whose Durand-Kerner loop updated the ten estimates one after the
// other, 7.7 us per sweep, and whose launch took as long as its slowest lane: clusters and multiple roots converge linearly and run
// into the cap of 300 sweeps, so 19 200 hypotheses -- 300 waves, one per CU, 63 lanes of most of them idle -- cost 2.3 ms whatever the
// average (timing-only builds: cap 80: 0.95 ms, 40: 0.80 ms, 20: 0.72 ms; results change below ~100).  Here a sweep updates all ten
// estimates at once from the previous sweep's values (Weierstrass / Durand-Kerner in its simultaneous form: the same fixed points,
// the same quadratic convergence at simple roots): p(z_i) by Horner in every lane, the other estimates through DPP row broadcasts, one
// division per lane.  A group stops when every estimate is at rest (step <= 1e-13 of its magnitude, or down at its evaluation noise;
// or after 300 sweeps) and is frozen from then on, so a hypothesis' result does not depend on which others share its wave.  Then lane i polishes its estimate
// on the real axis if it is real to 1e-8 (Newton), back-substitutes (x, y from the null vector of B(z)), forms E and takes the
// output slot given by its rank in (E[0][0], z, i) among the group's valid lanes -- the order the sequential code produced by
// sorting the roots, then the models.
__global__ __launch_bounds__(256) void essential_roots_kernel(int n, double *__restrict__ models, int32_t *__restrict__ n_models)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int g = t >> 4, i = t & 15, grp = (threadIdx.x & 63) >> 4;
    const bool in = g < n;
    const size_t gc = in ? (size_t)g : (size_t)(n - 1);
    const double *w = models + 90 * gc;
    const bool have_poly = in && n_models[gc] > 0;
    const double c10 = have_poly ? w[kSetupDet + 10] : 0.0;
    const bool have = have_poly && fabs(c10) > 0.0;
    double cc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) cc[k] = have ? w[kSetupDet + k] / c10 : 0.0;
    // start: a spiral around the origin (no symmetry of the polynomial can be a symmetry of the start)
    double re, im;
    {
        // start radius: half of Fujiwara's bound 2 max_k |c_{n-k}|^(1/k) (every root lies within it).  Cauchy's 1 + max |c_k|, used
        // until round 3, is looser by orders of magnitude here, and the estimates approach from outside by a factor ~ 9/10 per sweep:
        // mean sweeps per hypothesis 75 -> 45, median 60 -> 28 (scratch/dk_hist.py), the same RANSAC outcome on every test
        double fuji = 0.0, sn, cs;
#pragma unroll
        for (int k = 1; k <= 10; ++k) fuji = fmax(fuji, pow(fabs(cc[10 - k]) * (k == 10 ? 0.5 : 1.0), 1.0 / k));
        double r = fuji;
        if (!(r > 1e-300)) r = 1.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) r = k < i ? r * 0.9 : r;
        sincos(2.0 * 3.14159265358979323846 * (i < 10 ? i : 0) / 10.0 + 0.4, &sn, &cs);
        re = r * cs; im = r * sn;
    }
    bool active = have;                                   // (uniform over the group)
    double best_mv = 1e300;
    int stale = 0;
#ifdef ESFM_DK_HIST
    int my_sweeps = 0;
#endif
    for (int it = 0; it < 300; ++it) {
        if (!__any(active)) break;
#ifdef ESFM_DK_HIST
        my_sweeps += active ? 1 : 0;
#endif
        double pr = 1.0, pim = 0.0;                       // p(z_i), monic, by Horner
#pragma unroll
        for (int k = 9; k >= 0; --k) { const double tt = pr * re - pim * im + cc[k]; pim = pr * im + pim * re; pr = tt; }
        double dr = 1.0, di = 0.0;                        // prod_{j != i} (z_i - z_j), j ascending
        dk_factor<0>(re, im, i, dr, di); dk_factor<1>(re, im, i, dr, di); dk_factor<2>(re, im, i, dr, di); dk_factor<3>(re, im, i, dr, di);
        dk_factor<4>(re, im, i, dr, di); dk_factor<5>(re, im, i, dr, di); dk_factor<6>(re, im, i, dr, di); dk_factor<7>(re, im, i, dr, di);
        dk_factor<8>(re, im, i, dr, di); dk_factor<9>(re, im, i, dr, di);
        const double den = dr * dr + di * di;
        const double inv = den > 0.0 ? 1.0 / den : 0.0;
        const double qr = (pr * dr + pim * di) * inv, qi = (pim * dr - pr * di) * inv;
        const bool upd = active && i < 10;
        re = upd ? re - qr : re; im = upd ? im - qi : im;
        // An estimate is at rest when its step is below 1e-13 of its magnitude -- or when it has reached the noise of its own
        // evaluation: an ill-conditioned root of this degree-10 polynomial never gets its step under 1e-13 (5.4 % of the hypotheses
        // used to run into the cap of 300 sweeps for that), it jitters at 1e-12 .. 1e-9 instead.  So: a step that is already small
        // (< 1e-7) and has not halved for twelve sweeps is noise.  Linear convergence at a cluster (ratio (m - 1) / m <= 0.9 per
        // sweep) halves within seven and goes on; the approach from the start circle has steps of ~0.1 and is not affected.
        const double mv = (fabs(qr) + fabs(qi)) / (fabs(re) + fabs(im) + 1e-300);
        const bool better = mv < 0.5 * best_mv;
        best_mv = better ? mv : best_mv;
        stale = better ? 0 : stale + 1;
        const bool still = upd && !(mv <= 1e-13) && !(stale >= 12 && best_mv < 1e-7);
        const unsigned long long moving = __ballot(still);
        if (((moving >> (16 * grp)) & 0xffffull) == 0ull) active = false;
    }
#ifdef ESFM_DK_HIST
    if (have && i == 0) atomicAdd(&g_dk_hist[my_sweeps], 1u);
#endif
    // near-real estimates: Newton on the real axis
    const bool is_real = have && i < 10 && !(fabs(im) > 1e-8 * fmax(1.0, fabs(re)));
    double z = re;
    for (int nit = 0; nit < 4; ++nit) {
        double pz = 1.0, dz = 0.0;
#pragma unroll
        for (int k = 9; k >= 0; --k) { dz = dz * z + pz; pz = pz * z + cc[k]; }
        if (dz == 0.0) break;
        z -= pz / dz;
    }
    // B(z), its null vector, E
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
    double Ev[9], nrm = 0.0;
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
    // output slot: rank in (E[0][0], z, i) among the group's valid lanes
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const double k0 = __shfl(Ev[0], j, 16), kz = __shfl(z, j, 16);
        const int vj = __shfl((int)valid, j, 16);
        rank += (vj && (k0 < Ev[0] || (k0 == Ev[0] && (kz < z || (kz == z && j < i))))) ? 1 : 0;
    }
    const unsigned long long vm = __ballot(valid);
    const int count = __popcll((vm >> (16 * grp)) & 0xffffull);
    // (every lane of the group is past its reads of the setup values: the models take their place)
    if (valid) {
        double *dst = models + 90 * gc + 9 * rank;
#pragma unroll
        for (int a = 0; a < 9; ++a) dst[a] = Ev[a];
    }
    if (in && i == 0) n_models[g] = have ? count : 0;
}

__device__ __forceinline__ bool sampson_inlier(const double *E, double x1, double y1, double x2, double y2, float t)
{
    const double Ex0 = E[0] * x1 + E[1] * y1 + E[2], Ex1 = E[3] * x1 + E[4] * y1 + E[5], Ex2 = E[6] * x1 + E[7] * y1 + E[8];
    const double Et0 = E[0] * x2 + E[3] * y2 + E[6], Et1 = E[1] * x2 + E[4] * y2 + E[7];
    const double v = x2 * Ex0 + y2 * Ex1 + Ex2;
    const float err = (float)(v * v / (Ex0 * Ex0 + Ex1 * Ex1 + Et0 * Et0 + Et1 * Et1));
    return err <= t;
}

// the correspondences in normalised coordinates (x1, y1, x2, y2), once per call: one workgroup per pair
__global__ __launch_bounds__(256) void essential_normalise_kernel(const RansacPair *__restrict__ pairs, const float2 *__restrict__ p1,
                                                                  const float2 *__restrict__ p2, double4 *__restrict__ npts)
{
    const RansacPair pr = pairs[blockIdx.x];
    for (int i = threadIdx.x; i < pr.count; i += 256) {
        double x1, y1, x2, y2;
        normalise_pt(pr, p1, p2, i, x1, y1, x2, y2);
        npts[pr.first + i] = make_double4(x1, y1, x2, y2);
    }
}

int launch_essential_normalise(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, double *npts)
{
    if (n_pairs <= 0) return ESFM_OK;
    hipLaunchKernelGGL(essential_normalise_kernel, dim3(n_pairs), dim3(256), 0, st, pairs, reinterpret_cast<const float2 *>(p1),
                       reinterpret_cast<const float2 *>(p2), reinterpret_cast<double4 *>(npts));
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

// one workgroup per (pair, iteration): counts[g][m] = inliers of model m
__global__ __launch_bounds__(256) void essential_score_kernel(const RansacPair *__restrict__ pairs, const double4 *__restrict__ npts, int chunk,
                                                              const double *__restrict__ models, const int32_t *__restrict__ n_models,
                                                              int32_t *__restrict__ counts)
{
    __shared__ int red[10][4];
    __shared__ double sE[90];
    const int g = blockIdx.x;
    const int nm = __builtin_amdgcn_readfirstlane(n_models[g]);
    if (nm <= 0) return;
    const RansacPair pr = pairs[g / chunk];
    for (int k = threadIdx.x; k < 9 * nm; k += 256) sE[k] = models[90 * (size_t)g + k];
    if (threadIdx.x < 40) red[threadIdx.x >> 2][threadIdx.x & 3] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Four correspondences per thread and sweep in registers (normalised once per call by essential_normalise_kernel: the four f64
    // divisions per correspondence were repeated for every hypothesis of every round), then the hypothesis' models one after the other
    // in a REAL loop over the nm of them, each with its nine entries in registers.  (Until round 3: correspondences outside, the ten
    // model slots unrolled inside under `m < nm` -- every workgroup evaluated all ten slots under lane masks, nine LDS reads per
    // slot and correspondence: 131 us per round against 79.)
    for (int i0 = threadIdx.x; i0 < pr.count; i0 += 4 * 256) {
        double4 v[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { ok[u] = i0 + 256 * u < pr.count; v[u] = ok[u] ? npts[pr.first + i0 + 256 * u] : make_double4(0.0, 0.0, 0.0, 0.0); }
#pragma unroll 1
        for (int m = 0; m < nm; ++m) {
            double E[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) E[k] = sE[9 * m + k];
            int c = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) c += (ok[u] && sampson_inlier(E, v[u].x, v[u].y, v[u].z, v[u].w, pr.thresh_sq)) ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
            if (lane == 0) red[m][wave] += c;              // (this wave's own slot)
        }
    }
    __syncthreads();
    if (threadIdx.x < 10) counts[10 * (size_t)g + threadIdx.x] = threadIdx.x < nm ? red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3] : 0;
}

// one workgroup per pair: inlier mask of its best model (best[9 * pair])
__global__ __launch_bounds__(256) void essential_mask_kernel(const RansacPair *__restrict__ pairs, const float2 *__restrict__ p1,
                                                             const float2 *__restrict__ p2, const double *__restrict__ best,
                                                             uint8_t *__restrict__ mask)
{
    const RansacPair pr = pairs[blockIdx.x];
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = best[9 * (size_t)blockIdx.x + k];
    for (int i = threadIdx.x; i < pr.count; i += 256) {
        double x1, y1, x2, y2;
        normalise_pt(pr, p1, p2, i, x1, y1, x2, y2);
        mask[pr.first + i] = sampson_inlier(E, x1, y1, x2, y2, pr.thresh_sq) ? 1 : 0;
    }
}

// smallest right singular vector of the 4 x 4 DLT system of ([I|0], P) in double (cv::triangulatePoints on CV_64F input)
__device__ void triangulate_f64(const double *P, double x0, double y0, double x1, double y1, double X[4])
{
    double A[4][4];
    A[0][0] = -1.0; A[0][1] = 0.0; A[0][2] = x0; A[0][3] = 0.0;
    A[1][0] = 0.0; A[1][1] = -1.0; A[1][2] = y0; A[1][3] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { A[2][k] = x1 * P[8 + k] - P[k]; A[3][k] = y1 * P[8 + k] - P[4 + k]; }
    double M[4][4], V[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) s += A[r][p] * A[r][q];
            M[p][q] = s; V[p][q] = p == q ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            diag += M[p][p] * M[p][p];
#pragma unroll
            for (int q = p + 1; q < 4; ++q) off += M[p][q] * M[p][q];
        }
        if (off <= 1e-34 * diag) break;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double apq = M[p][q];
                if (apq == 0.0) continue;
                const double theta = (M[q][q] - M[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const double mp = M[r][p], mq = M[r][q]; M[r][p] = c * mp - s * mq; M[r][q] = s * mp + c * mq; }
#pragma unroll
                for (int r = 0; r < 4; ++r) { const double mp = M[p][r], mq = M[q][r]; M[p][r] = c * mp - s * mq; M[q][r] = s * mp + c * mq; }
#pragma unroll
                for (int r = 0; r < 4; ++r) { const double vp = V[r][p], vq = V[r][q]; V[r][p] = c * vp - s * vq; V[r][q] = s * vp + c * vq; }
            }
    }
    int best = 0;
    double bv = M[0][0];
#pragma unroll
    for (int k = 1; k < 4; ++k) if (M[k][k] < bv) { bv = M[k][k]; best = k; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = V[r][0];
        v = best == 1 ? V[r][1] : v; v = best == 2 ? V[r][2] : v; v = best == 3 ? V[r][3] : v;
        X[r] = v;
    }
}

// grid (4, n_pairs): candidate pose c of pair p.  poses[p][c] = 3 x 4 [R | t].  cand_mask[c][point], good[p][c].
__global__ __launch_bounds__(256) void pose_cheirality_kernel(const RansacPair *__restrict__ pairs, const float2 *__restrict__ p1,
                                                              const float2 *__restrict__ p2, const double *__restrict__ poses,
                                                              const uint8_t *__restrict__ in_mask, int n_total, uint8_t *__restrict__ cand_mask,
                                                              int32_t *__restrict__ good)
{
    __shared__ int red[4];
    const int c = blockIdx.x, pi = blockIdx.y;
    const RansacPair pr = pairs[pi];
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = poses[12 * (4 * (size_t)pi + c) + k];
    int cnt = 0;
    for (int i = threadIdx.x; i < pr.count; i += 256) {
        double x0, y0, x1, y1;
        normalise_pt(pr, p1, p2, i, x0, y0, x1, y1);
        double Q[4];
        triangulate_f64(P, x0, y0, x1, y1, Q);
        bool m = Q[2] * Q[3] > 0.0;
        const double X = Q[0] / Q[3], Y = Q[1] / Q[3], Z = Q[2] / Q[3];
        m = m && (Z < pr.dist_thresh);
        const double Z2 = P[8] * X + P[9] * Y + P[10] * Z + P[11];
        m = m && (Z2 > 0.0) && (Z2 < pr.dist_thresh);
        if (in_mask) m = m && in_mask[pr.first + i] != 0;
        cand_mask[(size_t)c * n_total + pr.first + i] = m ? 1 : 0;
        cnt += m ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) good[4 * pi + c] = red[0] + red[1] + red[2] + red[3];
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
int launch_essential_chunk(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, const double *npts,
                           const int32_t *samples, int chunk, double *models, int32_t *n_models, int32_t *counts, esfm_ctx *timing_ctx)
{
    const int n = n_pairs * chunk;
    if (n <= 0) return ESFM_OK;
    KernelTimer tm(timing_ctx, ESFM_K_RANSAC);
    hipLaunchKernelGGL(essential_setup_kernel, dim3((n + kSetupLanes - 1) / kSetupLanes), dim3(kSetupLanes), 0, st, pairs, n_pairs, reinterpret_cast<const float2 *>(p1),
                       reinterpret_cast<const float2 *>(p2), samples, chunk, models, n_models);
    ESFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(essential_roots_kernel, dim3((16 * n + 255) / 256), dim3(256), 0, st, n, models, n_models);
    ESFM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(essential_score_kernel, dim3(n), dim3(256), 0, st, pairs, reinterpret_cast<const double4 *>(npts), chunk, models, n_models, counts);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

__global__ __launch_bounds__(256) void essential_take_best_kernel(const int32_t *__restrict__ take, int n_take, const double *__restrict__ models,
                                                                  double *__restrict__ best)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 9 * n_take) return;
    const int e = t / 9, a = t % 9;
    best[9 * (size_t)take[3 * e] + a] = models[90 * (size_t)take[3 * e + 1] + 9 * (size_t)take[3 * e + 2] + a];
}

int launch_essential_take_best(hipStream_t st, const int32_t *take, int n_take, const double *models, double *best)
{
    if (n_take <= 0) return ESFM_OK;
    hipLaunchKernelGGL(essential_take_best_kernel, dim3((9 * n_take + 255) / 256), dim3(256), 0, st, take, n_take, models, best);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_essential_mask(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, const double *best, uint8_t *mask)
{
    if (n_pairs <= 0) return ESFM_OK;
    hipLaunchKernelGGL(essential_mask_kernel, dim3(n_pairs), dim3(256), 0, st, pairs, reinterpret_cast<const float2 *>(p1),
                       reinterpret_cast<const float2 *>(p2), best, mask);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

int launch_pose_cheirality(hipStream_t st, const RansacPair *pairs, int n_pairs, const float *p1, const float *p2, const double *poses,
                           const uint8_t *in_mask, int n_total, uint8_t *cand_mask, int32_t *good)
{
    if (n_pairs <= 0) return ESFM_OK;
    hipLaunchKernelGGL(pose_cheirality_kernel, dim3(4, n_pairs), dim3(256), 0, st, pairs, reinterpret_cast<const float2 *>(p1),
                       reinterpret_cast<const float2 *>(p2), poses, in_mask, n_total, cand_mask, good);
    ESFM_HIP_TRY(hipGetLastError());
    return ESFM_OK;
}

}  // namespace esfm

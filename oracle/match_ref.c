/*
 * oracle/match_ref.c -- CPU restatement of the reference's pairwise matcher.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (easysfm_amd/, the
 * C-ABI library, bin/) may include, link, call or execute this file.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * (SURVEY.md section 4), and it cannot be built here (needs OpenCV 3.x +
 * contrib, Ceres, PCL).  The arithmetic lives in un-vendored OpenCV (author ran
 * 3.4.2, python_code/notebook/test_feature.ipynb cell 0).  This file restates
 *   - the control flow of cpp_code/src/feature_matching.cpp:71-97 (ORB) and
 *     :115-142 (SURF): knnMatch(query, train, nn2, 2), then keep nn2[i][0] iff
 *     nn2[i][0].distance < ratio_thre * nn2[i][1].distance  (:88, :133; the
 *     right-hand side is double*float, so the compare happens in double),
 *     survivors appended in query order (:90, :135);
 *   - the exact brute-force variant python_code/feature_match.py:31-40 uses
 *     for SURF (cv2.BFMatcher(NORM_L2).knnMatch(k=2)); the C++ SURF path uses
 *     the approximate FlannBasedMatcher (feature_matching.cpp:120), whose
 *     output is not a function of the inputs alone (randomised kd-trees), so
 *     "bit-exact" is only definable against exact brute force;
 *   - [upstream, from memory of OpenCV 3.4 modules/core/src/{stat,batch_distance}.cpp,
 *     NOT verifiable in this container]
 *       * L2 distance of one pair = sqrt(normL2Sqr_(a,b,n)) in float, where the
 *         SSE2-baseline normL2Sqr_ keeps two 4-lane accumulators over blocks
 *         of 8 elements (t=a-b; acc += t*t, separate multiply and add), adds
 *         the two accumulators lane-wise, sums the 4 lanes left to right and
 *         then adds the scalar tail sequentially;
 *       * Hamming distance = popcount(a xor b) over nbytes, as int, converted
 *         to float in DMatch.distance;
 *       * K-best selection per query row: scan train rows in ascending index,
 *         insert d iff d < current K-th best (strict), shifting entries that
 *         are > d.  Equal distances therefore keep the LOWER train index
 *         first, and a NaN distance is never inserted.
 *   - defined behaviour where the reference has undefined behaviour: with
 *     fewer than 2 train rows the reference indexes nn2[i][1] out of bounds
 *     (feature_matching.cpp:88,133); here such a query emits no match and
 *     knn2 reports index -1 / distance FLT_MAX for the missing neighbours.
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off; NO -ffast-math: the
 * summation order above is part of the contract).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Canonical squared L2 distance (see header): 8 partial sums over blocks of 8,
 * (acc[c] + acc[c+4]) for c = 0..3, summed left to right, then the tail. */
float esfm_ref_l2sqr(const float *a, const float *b, int n)
{
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j <= n - 8; j += 8) {
        for (int c = 0; c < 8; ++c) {
            float t = a[j + c] - b[j + c];
            acc[c] = acc[c] + t * t;
        }
    }
    float s0 = acc[0] + acc[4];
    float s1 = acc[1] + acc[5];
    float s2 = acc[2] + acc[6];
    float s3 = acc[3] + acc[7];
    float d = s0 + s1;
    d = d + s2;
    d = d + s3;
    for (; j < n; ++j) {
        float t = a[j] - b[j];
        d = d + t * t;
    }
    return d;
}

static inline int popcount8(uint8_t x) { return __builtin_popcount((unsigned)x); }

int esfm_ref_hamming(const uint8_t *a, const uint8_t *b, int nbytes)
{
    int d = 0;
    for (int j = 0; j < nbytes; ++j) d += popcount8((uint8_t)(a[j] ^ b[j]));
    return d;
}

/* K = 2 insertion of (d, j) into (d0,i0),(d1,i1), OpenCV batchDistance rule. */
#define KNN2_INSERT(d, j, d0, i0, d1, i1) \
    do {                                  \
        if ((d) < (d1)) {                 \
            if ((d0) > (d)) {             \
                d1 = d0; i1 = i0;         \
                d0 = (d); i0 = (j);       \
            } else {                      \
                d1 = (d); i1 = (j);       \
            }                             \
        }                                 \
    } while (0)

/* knnMatch(query, train, out, 2) for NORM_L2 on float descriptors
 * (feature_matching.cpp:125 with exact BF; feature_match.py:33-34).
 * idx[2*i+k], dist[2*i+k] = k-th neighbour of query i; -1 / FLT_MAX if absent.
 * The plain loop: one esfm_ref_l2sqr per (query, train) pair.  Kept as the definition; the SIMD body below must (and is tested to)
 * return the same bits. */
void esfm_ref_knn2_l2_f32_scalar(const float *q, int nq, const float *t, int nt, int dim,
                                 int32_t *idx, float *dist)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < nq; ++i) {
        float d0 = FLT_MAX, d1 = FLT_MAX;
        int32_t i0 = -1, i1 = -1;
        const float *a = q + (size_t)i * dim;
        for (int j = 0; j < nt; ++j) {
            float d = sqrtf(esfm_ref_l2sqr(a, t + (size_t)j * dim, dim));
            KNN2_INSERT(d, j, d0, i0, d1, i1);
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        dist[2 * i] = d0; dist[2 * i + 1] = d1;
    }
}

/* The same result at the speed of a SIMD brute-force matcher (what cv::BFMatcher's batchDistance is on the reference's side; round 4:
 * the cpu_baseline of bench.py must not be a scalar strawman).  Vectorised ACROSS train rows -- eight rows per 256-bit register, lane =
 * row -- so that every lane performs exactly the scalar function's operations in exactly its order: eight accumulators acc[c] over
 * blocks of eight elements (t = a - b; acc[c] = acc[c] + t * t: separate IEEE multiply and add, -ffp-contract=off), then
 * (acc[c] + acc[c + 4]) for c = 0..3 summed left to right, sqrtf (IEEE).  The train set is transposed once per call into blocks of
 * eight rows (element-major inside a block); rows past nt are padded with +inf so that their distance is never inserted.  Needs
 * dim % 8 == 0; other widths take the plain loop. */
typedef float esfm_v8f __attribute__((vector_size(32), aligned(4)));
typedef int esfm_v8i __attribute__((vector_size(32)));

int esfm_ref_ratio_filter(const int32_t *idx, const float *dist, int nq, double ratio,
                          int32_t *query_idx, int32_t *train_idx, float *distance);

/* train rows -> blocks of eight rows, element-major inside a block (tt needs 8 * ceil(nt / 8) * dim floats) */
static void transpose_train8(const float *t, int nt, int dim, float *tt)
{
    const int nblk = (nt + 7) / 8;
    for (int b = 0; b < nblk; ++b)
        for (int e = 0; e < dim; ++e)
            for (int l = 0; l < 8; ++l) {
                const int row = 8 * b + l;
                tt[((size_t)b * dim + e) * 8 + l] = row < nt ? t[(size_t)row * dim + e] : INFINITY;
            }
}

/* the two nearest rows of a CHUNK of up to ESFM_QCHUNK consecutive queries against a transposed train set: block of eight train rows
 * outermost, so that the block (8 x dim floats) is read from memory once per chunk and stays in L1 for its queries -- with one
 * query per pass over the train set every thread streamed the whole set (1 MiB at 4096 x 64) per query and 256 threads were bound
 * by the shared cache: 95 pairs/s on the box where one thread did 25.  Per query the rows still arrive in ascending index. */
#define ESFM_QCHUNK 64
static inline void knn2_chunk8(const float *q, int nqc, const float *tt, int nblk, int dim, int32_t *idx, float *dist)
{
    float d0[ESFM_QCHUNK], d1[ESFM_QCHUNK];
    int32_t i0[ESFM_QCHUNK], i1[ESFM_QCHUNK];
    for (int k = 0; k < nqc; ++k) { d0[k] = d1[k] = FLT_MAX; i0[k] = i1[k] = -1; }
    for (int b = 0; b < nblk; ++b) {
        const esfm_v8f *tb = (const esfm_v8f *)(tt + (size_t)b * dim * 8);
        for (int k = 0; k < nqc; ++k) {
            const float *a = q + (size_t)k * dim;
            esfm_v8f acc[8];
            for (int c = 0; c < 8; ++c) acc[c] = (esfm_v8f){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < dim; j += 8)
                for (int c = 0; c < 8; ++c) {
                    const float ac = a[j + c];
                    const esfm_v8f d = (esfm_v8f){ac, ac, ac, ac, ac, ac, ac, ac} - tb[j + c];
                    acc[c] = acc[c] + d * d;
                }
            const esfm_v8f s0 = acc[0] + acc[4], s1 = acc[1] + acc[5], s2 = acc[2] + acc[6], s3 = acc[3] + acc[7];
            esfm_v8f dd = s0 + s1;
            dd = dd + s2;
            dd = dd + s3;
            float dv[8];
            int any = 0;
            const float lim = d1[k];
            for (int l = 0; l < 8; ++l) { dv[l] = sqrtf(dd[l]); any |= dv[l] < lim; }
            if (any)
                for (int l = 0; l < 8; ++l) {          /* ascending train index, like the plain loop (inf / NaN never pass `<`) */
                    const float d = dv[l];
                    const int32_t j2 = 8 * b + l;
                    KNN2_INSERT(d, j2, d0[k], i0[k], d1[k], i1[k]);
                }
        }
    }
    for (int k = 0; k < nqc; ++k) { idx[2 * k] = i0[k]; idx[2 * k + 1] = i1[k]; dist[2 * k] = d0[k]; dist[2 * k + 1] = d1[k]; }
}

void esfm_ref_knn2_l2_f32(const float *q, int nq, const float *t, int nt, int dim,
                          int32_t *idx, float *dist)
{
    if (dim % 8 != 0 || dim <= 0 || nt <= 0 || nq <= 0) { esfm_ref_knn2_l2_f32_scalar(q, nq, t, nt, dim, idx, dist); return; }
    const int nblk = (nt + 7) / 8;
    float *tt = (float *)aligned_alloc(64, (size_t)nblk * 8 * (size_t)dim * sizeof(float));
    if (!tt) { esfm_ref_knn2_l2_f32_scalar(q, nq, t, nt, dim, idx, dist); return; }
    transpose_train8(t, nt, dim, tt);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < nq; i += ESFM_QCHUNK) {
        const int nqc = nq - i < ESFM_QCHUNK ? nq - i : ESFM_QCHUNK;
        knn2_chunk8(q + (size_t)i * dim, nqc, tt, nblk, dim, idx + 2 * i, dist + 2 * i);
    }
    free(tt);
}

/* The pair loop of test/sfm.cpp:140-161 in one call (the CPU side of esfm_match_pairs_dev; bench.py's cpu_baseline): all sets in one
 * row-concatenated buffer, pair p = (query set, train set), its survivors in the slice [out_offset[p], out_offset[p] + n_out[p]) of
 * query_idx / train_idx / distance (out_offset = exclusive prefix sum of the query sets' row counts, filled here).  One parallel
 * region over ALL (pair, query) items -- a region per pair costs a fork / join of every thread per 4096 queries, which on a
 * 256-thread host was 95 % of round 3's baseline.  Every set is transposed once.  dim % 8 == 0.  Returns 0, or -1 on a bad argument. */
int esfm_ref_match_pairs_l2(const float *desc, const int32_t *set_row_offset, int n_sets, int dim, const int32_t *pairs, int n_pairs,
                            double ratio, int32_t *query_idx, int32_t *train_idx, float *distance, int32_t *n_out, int64_t *out_offset)
{
    if (dim % 8 != 0 || dim <= 0 || n_sets <= 0 || n_pairs < 0) return -1;
    out_offset[0] = 0;
    for (int p = 0; p < n_pairs; ++p) {
        const int qs = pairs[2 * p];
        out_offset[p + 1] = out_offset[p] + (set_row_offset[qs + 1] - set_row_offset[qs]);
    }
    const int64_t total = out_offset[n_pairs];
    size_t *toff = (size_t *)malloc(sizeof(size_t) * ((size_t)n_sets + 1));
    toff[0] = 0;
    for (int s_ = 0; s_ < n_sets; ++s_) toff[s_ + 1] = toff[s_] + (size_t)((set_row_offset[s_ + 1] - set_row_offset[s_] + 7) / 8) * 8 * (size_t)dim;
    float *tt = (float *)aligned_alloc(64, (toff[n_sets] + 16) * sizeof(float));
    int32_t *kidx = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(total > 0 ? total : 1));
    float *kdist = (float *)malloc(sizeof(float) * 2 * (size_t)(total > 0 ? total : 1));
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int s_ = 0; s_ < n_sets; ++s_)
            transpose_train8(desc + (size_t)set_row_offset[s_] * dim, set_row_offset[s_ + 1] - set_row_offset[s_], dim, tt + toff[s_]);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t g0 = 0; g0 < total; g0 += ESFM_QCHUNK) {     /* chunks of the global query numbering, cut at pair boundaries */
            int64_t g = g0;
            const int64_t gend = g0 + ESFM_QCHUNK < total ? g0 + ESFM_QCHUNK : total;
            while (g < gend) {
                int lo = 0, hi = n_pairs - 1;               /* last p with out_offset[p] <= g (pairs with an empty query set share offsets) */
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (out_offset[mid] <= g) lo = mid; else hi = mid - 1; }
                const int p = lo, qs = pairs[2 * p], ts = pairs[2 * p + 1];
                const int nt = set_row_offset[ts + 1] - set_row_offset[ts];
                const int64_t pend = out_offset[p + 1] < gend ? out_offset[p + 1] : gend;
                const int nqc = (int)(pend - g);
                const float *a = desc + ((size_t)set_row_offset[qs] + (size_t)(g - out_offset[p])) * dim;
                if (nt > 0) knn2_chunk8(a, nqc, tt + toff[ts], (nt + 7) / 8, dim, kidx + 2 * g, kdist + 2 * g);
                else for (int64_t e = g; e < pend; ++e) { kidx[2 * e] = kidx[2 * e + 1] = -1; kdist[2 * e] = kdist[2 * e + 1] = FLT_MAX; }
                g = pend;
            }
        }
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int p = 0; p < n_pairs; ++p) {
            const int64_t o = out_offset[p];
            n_out[p] = esfm_ref_ratio_filter(kidx + 2 * o, kdist + 2 * o, (int)(out_offset[p + 1] - o), ratio, query_idx + o, train_idx + o, distance + o);
        }
    }
    free(kdist); free(kidx); free(tt); free(toff);
    return 0;
}

/* knnMatch(query, train, out, 2) for "BruteForce-Hamming" (feature_matching.cpp:74,80).
 * Distances are exact ints, reported as float like cv::DMatch::distance. */
void esfm_ref_knn2_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes,
                           int32_t *idx, float *dist)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < nq; ++i) {
        int d0 = INT32_MAX, d1 = INT32_MAX;
        int32_t i0 = -1, i1 = -1;
        const uint8_t *a = q + (size_t)i * nbytes;
        for (int j = 0; j < nt; ++j) {
            int d = esfm_ref_hamming(a, t + (size_t)j * nbytes, nbytes);
            KNN2_INSERT(d, j, d0, i0, d1, i1);
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        dist[2 * i] = (i0 >= 0) ? (float)d0 : FLT_MAX;
        dist[2 * i + 1] = (i1 >= 0) ? (float)d1 : FLT_MAX;
    }
}

/* Lowe ratio filter, feature_matching.cpp:84-92 / :129-137.  Returns the number
 * of survivors, written in query order (DMatch.queryIdx, .trainIdx, .distance). */
int esfm_ref_ratio_filter(const int32_t *idx, const float *dist, int nq, double ratio,
                          int32_t *query_idx, int32_t *train_idx, float *distance)
{
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        if (idx[2 * i] < 0 || idx[2 * i + 1] < 0) continue; /* reference: UB */
        if ((double)dist[2 * i] < ratio * (double)dist[2 * i + 1]) {
            query_idx[n] = i;
            train_idx[n] = idx[2 * i];
            distance[n] = dist[2 * i];
            ++n;
        }
    }
    return n;
}

/* matchFeaturesSURF with exact brute force (feature_matching.cpp:115-142). */
int esfm_ref_match_l2_f32(const float *q, int nq, const float *t, int nt, int dim, double ratio,
                          int32_t *query_idx, int32_t *train_idx, float *distance,
                          int32_t *scratch_idx /*2*nq*/, float *scratch_dist /*2*nq*/)
{
    esfm_ref_knn2_l2_f32(q, nq, t, nt, dim, scratch_idx, scratch_dist);
    return esfm_ref_ratio_filter(scratch_idx, scratch_dist, nq, ratio, query_idx, train_idx, distance);
}

/* matchFeaturesORB (feature_matching.cpp:71-97). */
int esfm_ref_match_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes, double ratio,
                           int32_t *query_idx, int32_t *train_idx, float *distance,
                           int32_t *scratch_idx, float *scratch_dist)
{
    esfm_ref_knn2_hamming(q, nq, t, nt, nbytes, scratch_idx, scratch_dist);
    return esfm_ref_ratio_filter(scratch_idx, scratch_dist, nq, ratio, query_idx, train_idx, distance);
}

void esfm_ref_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int esfm_ref_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
